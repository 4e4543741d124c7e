#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats
file into a short table (template arguments kept, argument lists dropped).

usage: summarize_prof.py <kernel_stats.csv> [forwards] > profiles/<name>.md
"""
import csv
import re
import sys


def short(name: str) -> str:
    name = name.replace("slsmi::(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    if name.startswith("at::native"):
        m = re.search(r"at::native::(\w+)", name)
        name = "torch:" + (m.group(1) if m else name[:40])
    return name[:60]


def main():
    path = sys.argv[1]
    fwd = float(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = list(csv.DictReader(open(path)))
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
        a[2] = min(a[2], float(r["MinNs"]))
        a[3] = max(a[3], float(r["MaxNs"]))
    tot = sum(a[1] for a in agg.values())
    print(f"source: {path}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | {a[3] / 1e3:.1f} | {100 * a[1] / tot:.2f} |")
    print(f"\ntotal kernel time {tot / 1e6:.2f} ms" + (f"; {fwd:g} forwards in the run" if fwd else ""))


if __name__ == "__main__":
    try:
        main()
        sys.stdout.flush()
    except BrokenPipeError:  # the reader (`| head`) has gone: not a failure of the run being summarised
        import os
        os.dup2(os.open(os.devnull, os.O_WRONLY), sys.stdout.fileno())
