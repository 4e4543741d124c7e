#!/usr/bin/env python3
"""One scoring forward cut out of a rocprofv3 kernel trace (`--kernel-trace --output-format csv`): the kernels from a
conv0 launch to the head's last kernel, in launch order, then summed by kernel.  Takes the LAST complete forward of
the timed region whose launches are all back to back (no isolated-shape timings, no set-up kernels in between).

usage: one_forward.py <kernel_trace.csv> [--seq] [--skip N] > profiles/<name>.md
  --seq     also print the launch sequence
  --skip N  take the forward N before the last complete one (default 1: not the very last, which timing windows follow)
"""
import csv
import re
import sys


def short(name: str) -> str:
    name = name.replace("slsmi::(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:64]


def main():
    path = sys.argv[1]
    seq = "--seq" in sys.argv
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 1
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    starts = [i for i, n in enumerate(names) if n.startswith("conv0_mfma_kernel")]
    ends = [i for i, n in enumerate(names) if n.startswith("sls_tail_kernel") or n.startswith("sae_fc2_kernel")]
    forwards = []
    for s in starts:
        e = next((e for e in ends if e > s), None)
        if e is not None and not any(s < t <= e for t in starts):
            forwards.append((s, e))
    if len(forwards) <= skip:
        raise SystemExit("no complete forward in the trace")
    s, e = forwards[-1 - skip]
    cut = rows[s:e + 1]
    t0, t1 = int(cut[0]["Start_Timestamp"]), int(cut[-1]["End_Timestamp"])
    agg = {}
    for r in cut:
        k = short(r["Kernel_Name"])
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += d
    tot = sum(a[1] for a in agg.values())
    print(f"source: {path}; forward {len(forwards) - skip} of {len(forwards)} in the trace, {len(cut)} launches\n")
    print(f"first launch to last end: {(t1 - t0) / 1e6:.3f} ms; sum of kernel durations {tot / 1e6:.3f} ms\n")
    print("| kernel | launches | total us | avg us | % of kernel time |")
    print("|---|---:|---:|---:|---:|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {a[0]} | {a[1] / 1e3:.1f} | {a[1] / a[0] / 1e3:.1f} | {100 * a[1] / tot:.2f} |")
    if seq:
        print("\nlaunch sequence (us from the first launch, duration us, kernel, grid):\n")
        for r in cut:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            print(f"    {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d / 1e3:8.1f}  {short(r['Kernel_Name'])}  "
                  f"grid {r.get('Grid_Size_X', '?')} wg {r.get('Workgroup_Size_X', '?')}")


if __name__ == "__main__":
    try:
        main()
        sys.stdout.flush()
    except BrokenPipeError:
        pass
