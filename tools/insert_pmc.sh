#!/bin/bash
# usage (GPU box, repo root): tools/insert_pmc.sh <outdir> [n_reads]  -- counters of the k-min-mer table kernels, a few per pass
# (separate --pmc passes with --kernel-trace only, as the guide prescribes; a pass whose counters this rocprofv3 does not know is skipped)
set -u
OUT=${1:-gpurun_out/pmc_insert}
N=${2:-10000000}
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
: > $OUT/pmc_insert.txt
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_INSTS_FLAT SQ_INSTS_SMEM" "TCC_REQ_sum TCC_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_ATOMIC_sum" "TCC_READ_sum TCC_WRITE_sum" "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
         "TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" "MeanOccupancyPerCU" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | tr " " "_")
  ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $c -d $ROOT/$OUT/p_$n -o p -- python $ROOT/tools/insert_once.py $N > /dev/null 2> $ROOT/$OUT/p_$n.err )
  if [ -f $OUT/p_$n/p_results.db ]; then
    python tools/rocpd_summary.py $OUT/p_$n/p_results.db 2>&1 | grep "n=" | grep -E "count_insert|rescue_count|slot_flag|emit_slots|emit_rescued" >> $OUT/pmc_insert.txt
  else
    echo "# $c: no result (counter unknown to this rocprofv3?): $(tail -1 $OUT/p_$n.err)" >> $OUT/pmc_insert.txt
  fi
  rm -rf $OUT/p_$n
done
cat $OUT/pmc_insert.txt
