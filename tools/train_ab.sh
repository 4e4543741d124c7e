#!/bin/bash
# usage: tools/train_ab.sh lib1.so lib2.so ...  -- training-step A/B of library builds on one box, two rounds
for round in 1 2; do
  for lib in "$@"; do
    echo "$lib $(SLSMI_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --mode train --steps 40 --warmup 8 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms")')"
  done
done
