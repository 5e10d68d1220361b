#!/bin/bash
# usage (GPU box, repo root): tools/round5_last.sh -- the closing check of the round on the tree as committed: smoke, the whole GPU suite, the driver's bench command
set -u
OUT=gpurun_out/round5_last
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf gpurun_out/test_failures
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; grep -B5 -A40 "^___\|FAILED" $OUT/pytest_gpu.log | head -80
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout_line.json 2> $OUT/bench_stderr.log
echo "bench exit $? line bytes $(wc -c < $OUT/bench_stdout_line.json)"; cp bench_detail.json $OUT/bench_detail.json
head -c 400 $OUT/bench_stdout_line.json; echo
python -c "
import json; d=json.load(open('$OUT/bench_stdout_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline_kminmer']['traffic'], d['roofline_index']['6'], d['checks'], d['legs'], d['parity'])"
test -d gpurun_out/test_failures && ls gpurun_out/test_failures; true
