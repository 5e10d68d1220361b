#!/bin/bash
# usage (GPU box, repo root): tools/round5_first.sh  -- round 5's first look: the whole GPU suite, the driver's own bench command (is the line parseable?), counters of the index passes
set -u
OUT=gpurun_out/round5_a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout.json 2> $OUT/bench_stderr.log
echo "bench exit $? line bytes $(wc -c < $OUT/bench_stdout.json)"
cp bench_detail.json $OUT/bench_detail.json 2>/dev/null
cat $OUT/bench_stdout.json
tools/index_traffic.sh round5_a 10000000 7 2>&1 | tail -20
