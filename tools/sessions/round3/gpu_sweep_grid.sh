#!/bin/bash
# sweep: workgroups of the instance-walking table kernels beside the scans x batches in flight (bench, 30 steps, no legs)
OUT=gpurun_out/r3k; mkdir -p $OUT
for cfg in "0 3" "128 3" "128 4" "64 4" "192 3" "64 5"; do
  set -- $cfg
  MDBG_BENCH_TABLE_GRID=$1 timeout 200 python bench.py --steps 30 --in-flight $2 --legs none --cpu-sample 0 > $OUT/grid_$1_if$2.json 2> $OUT/grid_$1_if$2.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/grid_$1_if$2.json"))
    k = d["kernel_ms_per_step"]
    print("grid $1 in_flight $2: %.1f Gbp/s  %.2f ms/step  scan %.1f insert %.1f purge %.1f rescue %.1f emit %.1f" % (d["value"], d["ms_per_step"], k["scan"], k["kminmer_insert"], k["purge_palindromes"], k["kminmer_rescue"], k["kminmer_emit"]))
except Exception as e:
    print("grid $1 in_flight $2 failed", e)
PY
done
