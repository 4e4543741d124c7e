#!/usr/bin/env python3
"""profiles/r03_pmc_bench.{json,md} from the three counter passes of tools/gpu_pmc_bench.sh
(gpurun_out/pmcb_{fetch,write,mfma}/summary.txt).  The epilogue-mode instantiations of a kernel are merged
(launch-weighted).  FETCH_SIZE / WRITE_SIZE are KiB at the memory side of the L2s; fetch is doubled (gfx950
counts 128-byte read requests as 64 bytes).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES over the SIMD cycles of the
launch (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs).
usage: pmc_bench_summary.py [out_dir=profiles]"""
import json
import os
import re
import sys
from collections import defaultdict

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "profiles")
tot = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))   # kernel -> counter -> [launches, sum]
for tag in ("fetch", "write", "mfma"):
    for line in open(os.path.join(root, "gpurun_out", f"pmcb_{tag}", "summary.txt")):
        k, c, n, s, _ = line.rstrip("\n").split("\t")
        k = re.sub(r"^(gemm16_(?:big|wide)_kernel<\d)(, \w+)+>", r"\1>", k)     # merge the epilogue modes
        k = re.sub(r"^(gemm16_(?:big|wide)_kernel<\d)>.*", r"\1>", k)
        a = tot[k][c]
        a[0] += int(n.split("=")[1])
        a[1] += float(s.split("=")[1])
kernels = {}
for k, d in tot.items():
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        continue
    n = d["FETCH_SIZE"][0]
    busy = 0.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE", [0, 0.0])[1] > 0:
        busy = d["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (d["GRBM_GUI_ACTIVE"][1] / 8.0 * 256 * 4)
    kernels[k] = {"launches": n, "fetch_bytes_per_launch": 2.0 * 1024 * d["FETCH_SIZE"][1] / n,
                  "write_bytes_per_launch": 1024 * d["WRITE_SIZE"][1] / d["WRITE_SIZE"][0], "mfma_busy_frac": busy}
order = sorted(kernels, key=lambda k: -kernels[k]["launches"] * (kernels[k]["fetch_bytes_per_launch"] +
                                                               kernels[k]["write_bytes_per_launch"]))
import hashlib
sha16 = hashlib.sha256(open(os.path.join(root, "slsforasvspoof-2021-df_amd", "libslsmi.so"), "rb").read()).hexdigest()[:16]
cmd = ("rocprofv3 --pmc <pass> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt "
       "(one run per pass: FETCH_SIZE | WRITE_SIZE | SQ/GRBM)")
json.dump({"command": cmd, "libslsmi_sha16": sha16,
           "units": "bytes per launch; FETCH_SIZE (KiB, memory-side requests of the L2s, Infinity-Cache hits included) "
                    "doubled as the gfx950 guide prescribes for 16-byte-per-lane streaming reads; WRITE_SIZE (KiB) as read",
           "kernels": {k: kernels[k] for k in order}}, open(os.path.join(out_dir, "r03_pmc_bench.json"), "w"), indent=1)
with open(os.path.join(out_dir, "r03_pmc_bench.md"), "w") as f:
    f.write("# PMC passes over the bench workload (B=64, f16 operands), one rocprofv3 run per pass\n\n")
    f.write(f"libslsmi.so build the counters belong to (first 16 hex digits of its SHA-256): `{sha16}`\n\n")
    f.write("`rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt`, passes: "
            "`FETCH_SIZE`; `WRITE_SIZE`; `SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 "
            "SQ_INSTS_MFMA GRBM_GUI_ACTIVE`. Table made by `tools/pmc_bench_summary.py`.\n\n")
    f.write("FETCH_SIZE and WRITE_SIZE are KiB of memory-side L2 requests (Infinity-Cache hits included); fetch is doubled "
            "(gfx950 counts 128-B read requests as 64 B). MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x "
            "256 CUs x 4 SIMDs).\n\n| kernel | launches | fetch MB/launch | write MB/launch | MFMA busy |\n|---|---:|---:|---:|---:|\n")
    for k in order[:20]:
        v = kernels[k]
        f.write(f"| {k} | {v['launches']} | {v['fetch_bytes_per_launch'] / 1e6:.1f} | {v['write_bytes_per_launch'] / 1e6:.1f} | "
                f"{100 * v['mfma_busy_frac']:.1f} % |\n")
print(open(os.path.join(out_dir, "r03_pmc_bench.md")).read())
