#!/bin/bash
# usage (GPU box, repo root): tools/round5_g.sh -- an N = 8 job in the shape the driver launches it (torch.distributed.run, one process per rank, the library's
# exchange by peer copies), all eight ranks on the one GPU of the box (MDBG_BENCH_SHARE_GPU=1; torch.distributed over gloo: RCCL refuses ranks that share a device)
set -u
OUT=gpurun_out/round5_g
mkdir -p $OUT
export TMPDIR=/tmp
MDBG_BENCH_SHARE_GPU=1 MDBG_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --reads 1000000 --steps 12 --warmup 4 --detail $OUT/eight_processes_detail.json > $OUT/eight_processes_line.json 2> $OUT/eight.err
echo "exit $? bytes $(wc -c < $OUT/eight_processes_line.json)"; cat $OUT/eight_processes_line.json
MDBG_BENCH_SHARE_GPU=1 MDBG_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 4 --total-reads 8000000 --steps 12 --warmup 4 --detail $OUT/four_processes_strong_detail.json > $OUT/four_processes_strong_line.json 2> $OUT/four.err
echo "exit $? bytes $(wc -c < $OUT/four_processes_strong_line.json)"; cat $OUT/four_processes_strong_line.json
grep -c "" $OUT/eight.err $OUT/four.err; grep -i "fault\|error" $OUT/eight.err $OUT/four.err | head
