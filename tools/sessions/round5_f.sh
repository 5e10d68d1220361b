#!/bin/bash
# usage (GPU box, repo root): tools/round5_f.sh -- the multi-rank tests over and over (a GPU memory access fault once in ten runs before the exchange stopped unmapping and freeing in mid-run)
set -u
OUT=gpurun_out/round5_f
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf gpurun_out/test_failures
bad=0
for i in 1 2 3 4 5 6 7 8; do
  timeout 900 python -m pytest tests/test_gpu_multirank.py -q -x -p no:cacheprovider -k "peer or library or equal_one" > $OUT/multirank_$i.log 2>&1
  line=$(grep -E 'passed|failed' $OUT/multirank_$i.log | tail -1)
  echo "run $i: $line"
  case "$line" in *failed*) bad=$((bad+1));; esac
done
echo "runs with a failure: $bad"
ls gpurun_out/test_failures 2>/dev/null && for f in gpurun_out/test_failures/*.stderr; do echo "== $f"; head -2 "$f" | cut -c1-300; grep -n "fault\|MdbgError\|mdbg\|Error" "$f" | head -20; done
timeout 600 python -m pytest tests/test_gpu_tool.py tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "library_exchange or exchange_failures" > $OUT/pytest_exchange.log 2>&1
grep -E "passed|failed" $OUT/pytest_exchange.log | tail -2
