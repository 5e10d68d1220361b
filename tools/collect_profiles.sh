#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <tag>
# Writes under gpurun_out/<tag>/ what profiles/<tag>_* is made of:
#   bench_stdout.json            python bench.py (default flags: N=1, cpu_baseline on)
#   bench_kernel_stats.txt       rocprofv3 --kernel-trace --stats of the same command (cpu sample off)
#   pmc_FETCH_SIZE.txt / pmc_WRITE_SIZE.txt   separate --pmc passes, one scan launch each (HBM traffic)
#   pmc_sq.txt                   SQ instruction counters of the scan kernel
set -u
TAG=${1:-r01c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
timeout 600 python bench.py > $OUT/bench_stdout.json 2> $OUT/bench_stderr.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/kt -o kt -- python $ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $ROOT/$OUT/kt_bench.json 2> $ROOT/$OUT/kt.err )
python tools/rocpd_summary.py $OUT/kt/kt_results.db > $OUT/bench_kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  # one batch in flight: the counters are device-wide while the kernel runs, a second batch's kernels would be counted too
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $ROOT/$OUT/pmc_$c -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --in-flight 1 > /dev/null 2> $ROOT/$OUT/pmc_$c.err )
  python tools/rocpd_summary.py $OUT/pmc_$c/p_results.db > $OUT/pmc_$c.txt 2>&1
done
: > $OUT/pmc_sq.txt
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  n=$(echo $c | tr " " "_")
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $ROOT/$OUT/sq_$n -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --in-flight 1 > /dev/null 2> $ROOT/$OUT/sq_$n.err )
  python tools/rocpd_summary.py $OUT/sq_$n/p_results.db 2>&1 | grep "n=" | grep scan_kernel >> $OUT/pmc_sq.txt
done
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/sq_*/   # keep the text, not the 64 MiB of databases
cat $OUT/bench_stdout.json | head -c 600; echo; head -12 $OUT/bench_kernel_stats.txt; grep -h "scan_kernel" $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt | grep "n="; cat $OUT/pmc_sq.txt
