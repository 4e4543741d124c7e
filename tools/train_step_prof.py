#!/usr/bin/env python3
"""Per-kernel time of ONE training step from a rocprofv3 kernel trace of `bench.py --mode train`:
the launches between the last two loss evaluations (nll_loss_forward), summed per kernel name."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "nll_loss_forward" in r["Kernel_Name"]]
if len(marks) < 2:
    sys.exit("fewer than two training steps in the trace")
step = rows[marks[-2]:marks[-1]]


def short(name):
    name = re.sub(r"void |at::native::|\(anonymous namespace\)::|slsmi::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


dur = collections.Counter()
cnt = collections.Counter()
for r in step:
    k = short(r["Kernel_Name"])
    dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[k] += 1
span = int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])
total = sum(dur.values())
print(f"one step: span {span / 1e3:.1f} us, kernel time {total / 1e3:.1f} us, {len(step)} launches")
print("| kernel | launches | total us | avg us | % |")
print("|---|---:|---:|---:|---:|")
for k, v in dur.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    print(f"| {k} | {cnt[k]} | {v / 1e3:.1f} | {v / 1e3 / cnt[k]:.1f} | {100 * v / total:.1f} |")
