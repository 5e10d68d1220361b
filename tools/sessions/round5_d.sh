#!/bin/bash
# usage (GPU box, repo root): tools/round5_d.sh -- the eight-rank peer-copy run that failed once: six more times with what every rank said kept; the refined pass's new insert timed
set -u
OUT=gpurun_out/round5_d
mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  timeout 400 python -m pytest tests/test_gpu_multirank.py -q -x -p no:cacheprovider -k "test_ranks_equal_one_by_peer_copies and 8-2" > $OUT/peer8_$i.log 2>&1
  echo "run $i: $(grep -E 'passed|failed' $OUT/peer8_$i.log | tail -1)"
done
ls gpurun_out/test_failures 2>/dev/null
for f in gpurun_out/test_failures/*.stderr; do [ -f "$f" ] && { echo "== $f"; grep -v "SIGTERM\|^ *time\|^ *host\|error_file\|^ *\[\|^ *rank\|^ *exitcode\|traceback\|^-*$\|^=*$" "$f" | tail -60; }; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_partition.py -q -x -p no:cacheprovider -k "refined or index or next_k or partition or first_pass or scan_reads" > $OUT/pytest_subset.log 2>&1
grep -E "passed|failed" $OUT/pytest_subset.log | tail -2
timeout 600 python tools/index_forms_time.py 10000000 8 > $OUT/index_forms.json 2> $OUT/index_forms.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/round5_d/index_forms.json"))
print("equal:", d["tables_equal_in_all_forms"])
for f, v in d["forms"].items():
    print(f, v["loop_ms_incl_first_pass"], {k: x["kernel_ms_total"] for k, x in v["per_k"].items()}, v["per_k"]["5"]["kernel_ms"])
PY
