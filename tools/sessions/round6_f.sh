#!/bin/bash
# round 6 (GPU box): the multi-rank tests after the communicator changes, then the driver's bench command
set -u
O=gpurun_out/r6f; mkdir -p $O
python -m pytest tests/test_gpu_multirank.py tests/test_gpu_fullsize_multik.py -q -m gpu -x > $O/pytest_multirank.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_multirank.log | tail -5
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_stderr.log; echo "bench rc $?"
cp bench_detail.json $O/bench_detail.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6f/bench_line.json'))
print(len(json.dumps(d)), d['value'], d['ms_per_step'], d.get('legs'), d.get('checks'), d['roofline'].get('valu_floor'))
print({k:v['ms'] for k,v in d.get('roofline_index',{}).items()})
PY
