#!/bin/bash
# usage (GPU box): tools/inflight_sweep.sh <outfile>   -- headline step for batches in flight x resident table blocks per CU
OUT=${1:-gpurun_out/inflight_sweep.txt}
: > $OUT
for inflight in 2 3 4; do
  for blocks in 1 2 4; do
    MDBG_TABLE_BLOCKS_PER_CU=$blocks timeout 200 python bench.py --steps 24 --legs none --cpu-sample 0 --in-flight $inflight 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('in_flight $inflight table_blocks_per_cu $blocks  %.1f Gbp/s  %.2f ms/step  scan %.1f insert %.1f purge %.1f' % (d['value'], d['ms_per_step'], k['scan'], k['kminmer_insert'], k['purge_palindromes']))" >> $OUT
  done
done
cat $OUT
