#!/bin/bash
# round 3, first GPU call: the whole GPU suite, the default bench, a sweep of the table kernels' CU confinement
set -u
OUT=gpurun_out/r3a
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host.txt; free -g | head -2 >> $OUT/host.txt; df -h /tmp | tail -1 >> $OUT/host.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
timeout 1500 python bench.py > $OUT/bench_stdout.json 2> $OUT/bench_stderr.log
echo "bench exit $?"
tail -5 $OUT/bench_stderr.log
for c in 0 16 32 64; do
  MDBG_BENCH_TABLE_CUS=$c timeout 300 python bench.py --steps 30 --legs none --cpu-sample 0 > $OUT/sweep_cus_$c.json 2> $OUT/sweep_cus_$c.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/sweep_cus_$c.json"))
    print("table_cus", $c, "Gbp/s %.1f ms_per_step %.2f scan_ms %.2f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]), d["kernel_ms_per_step"])
except Exception as e:
    print("table_cus", $c, "failed", e)
PY
done
head -c 1500 $OUT/bench_stdout.json
