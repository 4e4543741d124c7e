#!/usr/bin/env python3
"""Per-kernel mean of each counter from a rocprofv3 counter_collection csv (gemm kernels only)."""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "gemm16" not in name:
        continue
    k = re.sub(r"\(.*", "", name.replace("slsmi::(anonymous namespace)::", "").replace("void ", "")) + f" grid={r.get('Grid_Size','')}"
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
