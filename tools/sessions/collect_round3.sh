#!/bin/bash
# usage (GPU box, repo root): tools/collect_round3.sh <tag>  -- what profiles/<tag>_* is made of (every step under its own timeout)
#   pytest_gpu.log                 python -m pytest tests -m gpu
#   bench_stdout.json              python bench.py (default flags: parity on the whole configs[1], cpu_baseline, every leg)
#   bench_kernel_stats.txt         rocprofv3 --kernel-trace of the timed workload alone (--legs none --cpu-sample 0)
#   scan_traffic.json              FETCH_SIZE / WRITE_SIZE of the scan kernel at the bench's 10 M reads (separate --pmc passes), with the
#                                  git blob hash of csrc/scan.hip they were collected on
#   end_to_end_steady.json         tools/e2e_steady.py: 50 Gbp FASTA, 20 Gbp FASTQ, 10 Gbp gzip from /dev/shm, tool only
set -u
TAG=${1:-round3_final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_stdout.json 2> $OUT/bench_stderr.log
echo "bench exit $?"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $ROOT/$OUT/kt -o kt -- python $ROOT/bench.py --steps 10 --warmup 3 --legs none --cpu-sample 0 > $ROOT/$OUT/kt_bench.json 2> $ROOT/$OUT/kt.err )
python tools/rocpd_summary.py $OUT/kt/kt_results.db > $OUT/bench_kernel_stats.txt 2>&1
rm -rf $OUT/kt
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $ROOT/$OUT/pmc_$c -o p -- python $ROOT/tools/scan_once.py 10000000 > /dev/null 2> $ROOT/$OUT/pmc_$c.err )
  python tools/rocpd_summary.py $OUT/pmc_$c/p_results.db 2>&1 | grep "n=" | grep "scan_" > $OUT/pmc_$c.txt
  rm -rf $OUT/pmc_$c
done
python - <<PY
import hashlib, json, re
def val(path):
    m = re.search(r"avg=\s*([0-9.]+)", open(path).read())
    return float(m.group(1)) if m else None
f, w = val("$OUT/pmc_FETCH_SIZE.txt"), val("$OUT/pmc_WRITE_SIZE.txt")
b = json.load(open("$OUT/bench_stdout.json"))
alg = b["roofline"]["algorithmic_bytes_per_launch"]
data = open("metamdbg_amd/csrc/scan.hip", "rb").read()
blob = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
json.dump({"round": 3, "kernel": "scan_fast_kernel<HPC=1,QUAL=0,APPROX=1>", "workload": "10000000 x 10000 bp synthetic HiFi reads, one launch",
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python tools/scan_once.py 10000000",
           "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "gfx950_fetch_correction": 2.0,
           "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0 if f and w else None, "algorithmic_bytes_per_launch": alg,
           "scan_hip_blob": blob,
           "note": "FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for coalesced streaming reads on gfx950; "
                   "WRITE_SIZE as reported (uncalibrated); one scan alone on the device; scan_hip_blob = git hash-object of csrc/scan.hip at collection",
           "reads": 10000000, "read_len": 10000}, open("$OUT/scan_traffic.json", "w"), indent=1)
PY
timeout 700 python tools/e2e_steady.py --reads 5000000 --fastq-reads 2000000 --gz-reads 1000000 --threads 32 --out $OUT/end_to_end_steady.json > /dev/null 2> $OUT/e2e_steady.err
grep -v "^$" $OUT/e2e_steady.err | tail -5
head -c 600 $OUT/bench_stdout.json; echo; head -12 $OUT/bench_kernel_stats.txt; cat $OUT/scan_traffic.json | head -12
