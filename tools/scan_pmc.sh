#!/bin/bash
# usage (GPU box, repo root): tools/scan_pmc.sh <outdir> [n_reads]   -- SQ / HBM counters of the scan kernel, one counter pair per pass
set -u
OUT=${1:-gpurun_out/pmc}
N=${2:-1000000}
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
: > $OUT/pmc_scan.txt
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VALU SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_SCA SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | tr " " "_")
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c -d $ROOT/$OUT/p_$n -o p -- python $ROOT/tools/scan_once.py $N > /dev/null 2> $ROOT/$OUT/p_$n.err )
  python tools/rocpd_summary.py $OUT/p_$n/p_results.db 2>&1 | grep "n=" | grep "scan_" >> $OUT/pmc_scan.txt
  rm -rf $OUT/p_$n
done
cat $OUT/pmc_scan.txt
