#!/bin/bash
# usage (GPU box, repo root): tools/round5_i.sh -- where the time inside an exchange goes (mdbg_comm_times): the per-rank workload of an 8-GPU job on one GPU, one rank
# and two processes sharing the GPU; the multi-rank tests and the smoke test once more on the last tree
set -u
OUT=gpurun_out/round5_i
mkdir -p $OUT
export TMPDIR=/tmp
{
  echo "== one rank, 5 M reads of a 40 M-read metagenome, peer copies"
  MDBG_BENCH_FORCE_EXCHANGE=1 MDBG_BENCH_SPEC_RANKS=8 MDBG_COMM_MODE=peer timeout 400 python bench.py --reads 5000000 --steps 20 --warmup 4 --legs none --cpu-sample 0 --detail $OUT/x_peer.json 2> $OUT/x_peer.err > /dev/null
  python -c "import json; d=json.load(open('$OUT/x_peer.json')); e=d['config']['exchange']; print('   value', round(d['value'],1), 'exchange_ms_per_step', round(e['exchange_ms_per_step'],2), e['exchange_ms_per_step_rank0'], 'kernel ms per step', {k: round(v,2) for k,v in d['kernel_ms_per_step'].items() if k.startswith('shard')})"
  echo "== two processes x 5 M reads sharing the GPU, peer copies"
  MDBG_BENCH_SHARE_GPU=1 MDBG_BENCH_BACKEND=gloo MDBG_COMM_MODE=peer timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 2 --steps 20 --warmup 4 --detail $OUT/x_two.json 2> $OUT/x_two.err > /dev/null
  python -c "import json; d=json.load(open('$OUT/x_two.json')); e=d['config']['exchange']; print('   value', round(d['value'],1), 'exchange_ms_per_step', round(e['exchange_ms_per_step'],2), e['exchange_ms_per_step_rank0'], 'parity', d['parity'].get('table_equal'))"
} > $OUT/exchange_time_account.txt 2>&1
cat $OUT/exchange_time_account.txt
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_fullsize_multik.py -q -x -p no:cacheprovider > $OUT/pytest_subset.log 2>&1
grep -E "passed|failed" $OUT/pytest_subset.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
