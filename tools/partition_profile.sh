#!/bin/bash
# usage (GPU box, repo root): tools/partition_profile.sh <tag> [n_reads] [plan ...]
#   <tag>/partition_time.txt           tools/partition_time.py (HIP events, best of three)
#   <tag>/partition_kernel_stats.txt   rocprofv3 --kernel-trace of the same command: per-kernel time of the partitioned first pass
set -u
TAG=${1:-r4_partition}
N=${2:-10000000}
shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
timeout 300 python tools/partition_time.py $N "$@" > $OUT/partition_time.txt 2>&1
( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $ROOT/$OUT/kt -o kt -- python $ROOT/tools/partition_time.py $N "$@" > /dev/null 2> $ROOT/$OUT/kt.err )
python tools/rocpd_summary.py $OUT/kt/kt_results.db > $OUT/partition_kernel_stats.txt 2>&1
rm -rf $OUT/kt
