#!/bin/bash
# usage (GPU box, repo root): tools/round5_h.sh -- look-up and insert of an index pass in one kernel: parity in every form, then timed beside the others
set -u
OUT=gpurun_out/round5_h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_multik.py tests/test_gpu_tool.py -q -x -p no:cacheprovider -k "refined or index or next_k or multik or multi_k or graph_next_k or handover" > $OUT/pytest_subset.log 2>&1
grep -E "passed|failed" $OUT/pytest_subset.log | tail -2; grep -B5 -A30 "^___\|FAILED" $OUT/pytest_subset.log | head -80
timeout 900 python tools/index_forms_time.py 10000000 9 > $OUT/index_forms.json 2> $OUT/index_forms.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/round5_h/index_forms.json"))
print("equal:", d["tables_equal_in_all_forms"])
for f, v in d["forms"].items():
    print(f, v["loop_ms_incl_first_pass"], {k: (x["kernel_ms_total"], x["wall_ms"]) for k, x in v["per_k"].items()})
    print("    k=8", v["per_k"]["8"]["kernel_ms"])
PY
