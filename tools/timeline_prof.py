#!/usr/bin/env python3
"""Print one forward of a rocprofv3 kernel trace in launch order: start offset, duration and the idle gap before
each kernel, with consecutive equal kernels of the encoder layers condensed.

usage: timeline_prof.py <kernel_trace.csv> [forward index from the end, default 1]
"""
import csv
import re
import sys


def short(name):
    name = name.replace("slsmi::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", name)[:48]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "conv0_mfma" in r["Kernel_Name"]]
    a = starts[-back]
    b = starts[-back + 1] if back > 1 else len(rows)
    seg = rows[a:b]
    t0 = int(seg[0]["Start_Timestamp"])
    prev_end = t0
    busy = gaps = 0
    per = {}
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = short(r["Kernel_Name"])
        if "tail" in k and "sls" in k:
            last = e
        gap = s - prev_end
        d = per.setdefault(k, [0, 0, 0])
        d[0] += 1
        d[1] += e - s
        d[2] += max(gap, 0)
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap / 1e3:6.1f}  {k}")
        busy += e - s
        gaps += max(gap, 0)
        prev_end = max(prev_end, e)
    print(f"\nspan {(prev_end - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, idle gaps {gaps / 1e3:.1f} us")
    for k, d in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:48s} n={d[0]:3d} dur {d[1] / 1e3:8.1f} us  gaps before {d[2] / 1e3:7.1f} us")


if __name__ == "__main__":
    main()
