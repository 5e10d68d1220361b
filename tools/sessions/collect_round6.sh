#!/bin/bash
# usage (GPU box, repo root): tools/sessions/collect_round6.sh [tag]  -- what profiles/<tag>_* is made of (every step under its own timeout)
#   pytest_gpu.log, pytest_gpu_again.log   python -m pytest tests -m gpu, twice (the multi-rank tests once failed one session in three)
#   bench_stdout_line.json, bench_detail.json   the driver's own command: python bench.py --gpus 1 --steps 20 --warmup 5 (one compact line; the full result)
#   bench_kernel_stats_timed_workload_only.txt  rocprofv3 --kernel-trace of the timed workload alone (--legs none --cpu-sample 0)
#   scan_traffic.json, pmc_scan_10m_reads.txt   FETCH_SIZE / WRITE_SIZE and SQ counters of the scan kernel at 10 M reads (separate --pmc passes), blob-tagged
#   kminmer_traffic.json, kminmer_pmc.txt       the same for the k = 4 first pass (tools/partition_traffic.sh)
#   index_traffic.json, index_pmc.txt           the same for the refined and index passes (tools/index_traffic.sh)
#   exchange_per_rank_workload_one_gpu.txt      the per-rank workload of an 8-GPU job through the library's exchange on one GPU; two processes sharing it
set -u
TAG=${1:-round6_final4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
rm -rf gpurun_out/test_failures
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
tools/scan_pmc.sh $OUT 10000000 > /dev/null 2>&1
mv $OUT/pmc_scan.txt $OUT/pmc_scan_10m_reads.txt
python - <<PY
import hashlib, json, re
def val(name):
    for line in open("$OUT/pmc_scan_10m_reads.txt"):
        if line.split()[0] == name and "scan_fast_kernel" in line:
            m = re.search(r"avg=\s*([0-9.]+)", line)
            return float(m.group(1)) if m else None
f, w = val("FETCH_SIZE"), val("WRITE_SIZE")
alg = 0.25 * 1.0e11 + 10.0 * 373753601          # 0.25 B per base + 10 B per minimizer of this workload
data = open("metamdbg_amd/csrc/scan.hip", "rb").read()
blob = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
json.dump({"round": 6, "kernel": "scan_fast_kernel<HPC=1,QUAL=0,APPROX=1>", "workload": "10000000 x 10000 bp synthetic HiFi reads, one launch",
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python tools/scan_once.py (tools/scan_pmc.sh)",
           "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "gfx950_fetch_correction": 2.0,
           "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0 if f and w else None, "algorithmic_bytes_per_launch": alg,
           "scan_hip_blob": blob,
           "note": "FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for coalesced streaming reads on gfx950; "
                   "WRITE_SIZE as reported; one scan alone on the device; scan_hip_blob = git hash-object of csrc/scan.hip at collection",
           "reads": 10000000, "read_len": 10000}, open("$OUT/scan_traffic.json", "w"), indent=1)
PY
tools/partition_traffic.sh $TAG 10000000 > /dev/null 2>&1
tools/index_traffic.sh $TAG 10000000 7 > $OUT/index_traffic_summary.txt 2>&1
# the counters of THIS tree go where bench.py looks for them (profiles/*_traffic.json, matched by git blob hash), then the line
cp $OUT/scan_traffic.json profiles/${TAG}_scan_traffic.json; cp $OUT/kminmer_traffic.json profiles/${TAG}_kminmer_traffic.json; cp $OUT/index_traffic.json profiles/${TAG}_index_traffic.json
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout_line.json 2> $OUT/bench_stderr.log
echo "bench exit $? line bytes $(wc -c < $OUT/bench_stdout_line.json)"; cp bench_detail.json $OUT/bench_detail.json
( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $ROOT/$OUT/kt -o kt -- python $ROOT/bench.py --steps 10 --warmup 3 --legs none --cpu-sample 0 --detail $ROOT/$OUT/kt_detail.json > $ROOT/$OUT/kt_bench.json 2> $ROOT/$OUT/kt.err )
python tools/rocpd_summary.py $OUT/kt/kt_results.db > $OUT/bench_kernel_stats_timed_workload_only.txt 2>&1
# round 6: `graph` as one process per k from files; the loop to k = 70; the look-up / insert ablation
timeout 600 python tools/graph_per_k.py 10000000 4,5,6,11 3 > $OUT/graph_per_k_10m_reads.json 2> $OUT/graph_per_k.err
timeout 900 python tools/index_by_k.py 10000000 70 -1 > $OUT/index_by_k_to_70.json 2> $OUT/index_by_k.err
timeout 300 tools/ubench/lookup_ablate 10000000 6 14000000 0.22 > $OUT/lookup_ablation_10m_reads.txt 2>&1
python tools/scan_gaps.py $OUT/kt/kt_results.db 6 > $OUT/bench_scan_gaps.txt 2>&1
rm -rf $OUT/kt
# the exchange on one GPU: (a) the per-rank workload of an 8-GPU job (one rank, forced through the library's exchange), peer copies and RCCL;
# (b) two real processes of 5 M reads each sharing the GPU (peer copies)
{
  for mode in peer rccl; do
    echo "== (a) one rank, 5 M reads of a 40 M-read metagenome, MDBG_COMM_MODE=$mode"
    MDBG_BENCH_FORCE_EXCHANGE=1 MDBG_BENCH_SPEC_RANKS=8 MDBG_COMM_MODE=$mode timeout 400 python bench.py --reads 5000000 --steps 20 --warmup 4 --legs none --cpu-sample 0 --detail $OUT/x_$mode.json 2> $OUT/x_$mode.err | cut -c1-1200
    python -c "import json; d=json.load(open('$OUT/x_$mode.json')); print('   value', round(d['value'],1), 'exchange', d['config']['exchange'])"
  done
  echo "== plain step on the same reads (no exchange)"
  MDBG_BENCH_SPEC_RANKS=8 timeout 400 python bench.py --reads 5000000 --steps 20 --warmup 4 --legs none --cpu-sample 0 --detail $OUT/x_plain.json 2> /dev/null | cut -c1-400
  echo "== (b) two processes x 5 M reads sharing the GPU, peer copies"
  MDBG_BENCH_SHARE_GPU=1 MDBG_BENCH_BACKEND=gloo MDBG_COMM_MODE=peer timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 20 --warmup 4 --detail $OUT/x_two.json 2> $OUT/x_two.err | cut -c1-1500
  python -c "import json; d=json.load(open('$OUT/x_two.json')); print('   value', round(d['value'],1), 'exchange', d['config']['exchange'], 'parity', d['parity'].get('table_equal'), d['parity'].get('single_gpu_gbps'))"
} > $OUT/exchange_per_rank_workload_one_gpu.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu_again.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu_again.log
grep -E "passed|failed" $OUT/pytest_gpu_again.log | tail -2
ls gpurun_out/test_failures 2>/dev/null
head -c 600 $OUT/bench_stdout_line.json; echo; head -8 $OUT/bench_kernel_stats_timed_workload_only.txt; cat $OUT/index_traffic_summary.txt | tail -12; cat $OUT/exchange_per_rank_workload_one_gpu.txt | cut -c1-700
