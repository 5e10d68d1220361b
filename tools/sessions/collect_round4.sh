#!/bin/bash
# usage (GPU box, repo root): tools/collect_round4.sh <tag> [steps]  -- what profiles/<tag>_* is made of (every step under its own timeout)
#   pytest_gpu.log                 python -m pytest tests -m gpu
#   bench_stdout.json              python bench.py (default flags: parity on the whole configs[1], cpu_baseline, every leg, the self-checks)
#   bench_kernel_stats.txt         rocprofv3 --kernel-trace of the timed workload alone (--legs none --cpu-sample 0)
#   scan_traffic.json              FETCH_SIZE / WRITE_SIZE of the scan kernel at the bench's 10 M reads (separate --pmc passes), with the git blob
#                                  hash of csrc/scan.hip they were collected on;  pmc_scan.txt: its SQ counters on the same blob (tools/scan_pmc.sh)
#   kminmer_traffic.json, kminmer_pmc.txt   the same for the k = 4 first pass (tools/partition_traffic.sh)
set -u
TAG=${1:-round4_final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
timeout 1100 python bench.py > $OUT/bench_stdout.json 2> $OUT/bench_stderr.log
echo "bench exit $?"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $ROOT/$OUT/kt -o kt -- python $ROOT/bench.py --steps 10 --warmup 3 --legs none --cpu-sample 0 > $ROOT/$OUT/kt_bench.json 2> $ROOT/$OUT/kt.err )
python tools/rocpd_summary.py $OUT/kt/kt_results.db > $OUT/bench_kernel_stats.txt 2>&1
python tools/scan_gaps.py $OUT/kt/kt_results.db 6 > $OUT/bench_scan_gaps.txt 2>&1
rm -rf $OUT/kt
tools/scan_pmc.sh $OUT 10000000 > /dev/null 2>&1
python - <<PY
import hashlib, json, re
def val(name):
    for line in open("$OUT/pmc_scan.txt"):
        if line.split()[0] == name and "scan_fast_kernel" in line:
            m = re.search(r"avg=\s*([0-9.]+)", line)
            return float(m.group(1)) if m else None
f, w = val("FETCH_SIZE"), val("WRITE_SIZE")
b = json.loads(open("$OUT/bench_stdout.json").read().strip().splitlines()[-1])
alg = b["roofline"]["algorithmic_bytes_per_launch"]
data = open("metamdbg_amd/csrc/scan.hip", "rb").read()
blob = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
json.dump({"round": 4, "kernel": "scan_fast_kernel<HPC=1,QUAL=0,APPROX=1>", "workload": "10000000 x 10000 bp synthetic HiFi reads, one launch",
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python tools/scan_once.py (tools/scan_pmc.sh)",
           "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "gfx950_fetch_correction": 2.0,
           "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0 if f and w else None, "algorithmic_bytes_per_launch": alg,
           "scan_hip_blob": blob,
           "note": "FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for coalesced streaming reads on gfx950; "
                   "WRITE_SIZE as reported (uncalibrated); one scan alone on the device; scan_hip_blob = git hash-object of csrc/scan.hip at collection; "
                   "the SQ counters of the same blob are in pmc_scan.txt beside this file",
           "reads": 10000000, "read_len": 10000}, open("$OUT/scan_traffic.json", "w"), indent=1)
PY
tools/partition_traffic.sh $TAG 10000000 > /dev/null 2>&1
head -c 400 $OUT/bench_stdout.json; echo; head -8 $OUT/bench_kernel_stats.txt; head -12 $OUT/scan_traffic.json; tail -3 $OUT/bench_scan_gaps.txt
