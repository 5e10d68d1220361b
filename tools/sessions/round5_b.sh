#!/bin/bash
# usage (GPU box, repo root): tools/round5_b.sh -- the bucket tables of the passes above firstK: the GPU suite, both forms timed, counters of the new form
set -u
OUT=gpurun_out/round5_b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
grep -B5 -A40 "^___\|Error\|FAILED" $OUT/pytest_gpu.log | head -120
timeout 600 python tools/index_forms_time.py 10000000 11 > $OUT/index_forms.json 2> $OUT/index_forms.err
echo "forms exit $?"; tail -3 $OUT/index_forms.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/round5_b/index_forms.json"))
print("equal:", d["tables_equal_in_all_forms"])
for f, v in d["forms"].items():
    print(f, v["loop_ms_incl_first_pass"], {k: (x["kernel_ms_total"], x["wall_ms"]) for k, x in v["per_k"].items()})
PY
tools/index_traffic.sh round5_b 10000000 7 2>&1 | tail -16
