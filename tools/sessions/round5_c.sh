#!/bin/bash
# usage (GPU box, repo root): tools/round5_c.sh -- the index passes' kernels of round 5 (a slot in one trip, plain-load first look, two windows in flight) and asmStep
set -u
OUT=gpurun_out/round5_c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
grep -B5 -A40 "^___\|Error\|FAILED" $OUT/pytest_gpu.log | head -150
timeout 900 python tools/index_forms_time.py 10000000 11 > $OUT/index_forms.json 2> $OUT/index_forms.err
echo "forms exit $?"; tail -3 $OUT/index_forms.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/round5_c/index_forms.json"))
print("equal:", d["tables_equal_in_all_forms"])
for f, v in d["forms"].items():
    print(f, v["loop_ms_incl_first_pass"], {k: x["kernel_ms_total"] for k, x in v["per_k"].items()})
    print("   k=8:", v["per_k"]["8"]["kernel_ms"])
PY
timeout 900 python bench.py --steps 10 --warmup 4 --legs end_to_end,multik > $OUT/bench_stdout.json 2> $OUT/bench_stderr.log
echo "bench exit $?"; cp bench_detail.json $OUT/; cat $OUT/bench_stdout.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/round5_c/bench_detail.json"))
print(d["legs"]["end_to_end"])
print(d["legs"]["multik"]["seconds"], d["legs"]["multik"]["ms"])
PY
