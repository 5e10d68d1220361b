#!/bin/bash
# usage (GPU box, repo root): tools/round5_e.sh -- the whole GPU suite, the quality sums beside the scan (A/B), the default bench as the driver runs it
set -u
OUT=gpurun_out/round5_e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
grep -B5 -A60 "^___\|FAILED" $OUT/pytest_gpu.log | head -150
ls gpurun_out/test_failures 2>/dev/null && for f in gpurun_out/test_failures/*.stderr; do echo "== $f"; grep -v "SIGTERM\|^ *time\|^ *host\|error_file\|^ *\[\|^ *rank\|^ *exitcode\|traceback\|^-*$\|^=*$" "$f" | tail -60; done
timeout 600 python tools/ont_quality_ab.py 2000000 > $OUT/ont_quality_ab.json 2> $OUT/ont_quality_ab.err
echo "ont ab exit $?"; python -c "
import json; d=json.load(open('$OUT/ont_quality_ab.json'))
print(d['same_minimizers'], {k:(v['wall_ms_best'], v['kernel_ms_last']) for k,v in d.items() if isinstance(v, dict)})"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout.json 2> $OUT/bench_stderr.log
echo "bench exit $? line bytes $(wc -c < $OUT/bench_stdout.json)"; cp bench_detail.json $OUT/; cat $OUT/bench_stdout.json
