#!/usr/bin/env python3
"""Print the headline numbers of a bench.py log (one JSON line) compactly."""
import json
import sys

for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        r = d["roofline"]
        print(f"{d['dtype']}: {d['value']:.0f} utt/s  {d['ms_per_step']:.3f} ms/step  path {r['achieved']:.0f} TF "
              f"({100 * r['frac']:.1f}%)  device step {r['device_step_tflops']:.0f} TF  gemm isolated "
              f"{r['gemm_isolated_tflops']:.0f} TF ({100 * r['gemm_isolated_frac']:.1f}%)  n_gpus {d['n_gpus']} rccl {d.get('rccl_ranks')}")
        print("  " + "  ".join(f"{s['M']}x{s['N']}x{s['K']}:{s['tflops']:.0f}" for s in r["shapes"]))
        if "cpu_baseline" in d:
            print(f"  cpu {d['cpu_baseline']['value']:.2f} utt/s on {d['cpu_baseline']['cores']} threads")
        if "score_check" in d:
            print(f"  score_check {d['score_check']}")
        if "alt" in d:
            print(f"  alt {d['alt']}")
        if "gemm_in_place" in r:
            g = r["gemm_in_place"]
            print(f"  GEMM launches in place: {g['launches_per_step']:.0f} per step, {g['avg_launch_us']:.1f} us each, "
                  f"{g['ms_per_step']:.2f} ms per step, {g['tflops']:.0f} TF ({100 * g['frac']:.1f}%)")
        print(f"  traffic {r['traffic']} ({r['traffic_note']})")
        if "mfma_loop_tflops" in r:
            print(f"  register-only MFMA loop on this box {r['mfma_loop_tflops']:.0f} TF; path / loop = {r['achieved'] / r['mfma_loop_tflops']:.3f}")
