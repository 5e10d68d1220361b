#!/bin/bash
# usage (GPU box, repo root): tools/round5_final2.sh -- after the last change of csrc/kminmer.hip (the fused index kernel, kept as a measured option): the suite, the
# counters of the first pass and of the refined / index passes again on the final blobs, and the default line that reports them
set -u
TAG=round5_final
OUT=gpurun_out/${TAG}2
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf gpurun_out/test_failures
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; grep -B5 -A40 "^___\|FAILED" $OUT/pytest_gpu.log | head -100
tools/partition_traffic.sh ${TAG}2 10000000 > /dev/null 2>&1
tools/index_traffic.sh ${TAG}2 10000000 7 > $OUT/index_traffic_summary.txt 2>&1
cp $OUT/kminmer_traffic.json profiles/${TAG}_kminmer_traffic.json; cp $OUT/index_traffic.json profiles/${TAG}_index_traffic.json
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout_line.json 2> $OUT/bench_stderr.log
echo "bench exit $? line bytes $(wc -c < $OUT/bench_stdout_line.json)"; cp bench_detail.json $OUT/bench_detail.json
cat $OUT/bench_stdout_line.json; tail -6 $OUT/index_traffic_summary.txt | cut -c1-200
ls gpurun_out/test_failures 2>/dev/null
