#!/usr/bin/env python3
"""Per-kernel totals and per-launch means of every counter in a rocprofv3 counter_collection csv."""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    k = re.sub(r"\(.*", "", name.replace("slsmi::(anonymous namespace)::", "").replace("void ", ""))[:60]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    for c, v in sorted(d.items()):
        print(f"{k}\t{c}\tlaunches={len(v)}\tsum={sum(v):.6g}\tmean={sum(v) / len(v):.6g}")
