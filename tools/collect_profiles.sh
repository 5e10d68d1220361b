#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <tag>
# Writes under gpurun_out/<tag>/ what profiles/<tag>_* is made of:
#   bench_stdout.json            python bench.py (default flags: N=1, 10 M reads, parity + cpu_baseline + legs)
#   bench_in_flight_1.json       one context, no legs
#   bench_kernel_stats.txt       rocprofv3 --kernel-trace of the timed workload alone (--legs none): the scan launches are the bench's
#   bench_legs_kernel_stats.txt  the same with the legs (multik, pcie, ont, parity sample): every kernel they run
#   scan_traffic.json            FETCH_SIZE / WRITE_SIZE of the scan kernel at the bench's 10 M reads (separate --pmc passes)
#   pmc_scan.txt                 SQ counters of the scan kernel (1 M reads)
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
timeout 600 python bench.py > $OUT/bench_stdout.json 2> $OUT/bench_stderr.log
timeout 300 python bench.py --in-flight 1 --steps 20 --legs none --cpu-sample 0 > $OUT/bench_in_flight_1.json 2>> $OUT/bench_stderr.log
# the timed workload alone (warm-up, overlap probe and timed steps: every scan launch is the 10 M-read batch), then the legs
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $ROOT/$OUT/kt -o kt -- python $ROOT/bench.py --steps 10 --warmup 3 --legs none --cpu-sample 0 > $ROOT/$OUT/kt_bench.json 2> $ROOT/$OUT/kt.err )
python tools/rocpd_summary.py $OUT/kt/kt_results.db > $OUT/bench_kernel_stats.txt 2>&1
rm -rf $OUT/kt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $ROOT/$OUT/ktl -o kt -- python $ROOT/bench.py --steps 3 --warmup 3 --cpu-sample 20000 > $ROOT/$OUT/kt_legs_bench.json 2> $ROOT/$OUT/kt_legs.err )
python tools/rocpd_summary.py $OUT/ktl/kt_results.db > $OUT/bench_legs_kernel_stats.txt 2>&1
rm -rf $OUT/ktl
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $ROOT/$OUT/pmc_$c -o p -- python $ROOT/tools/scan_once.py 10000000 > /dev/null 2> $ROOT/$OUT/pmc_$c.err )
  python tools/rocpd_summary.py $OUT/pmc_$c/p_results.db 2>&1 | grep "n=" | grep "scan_" > $OUT/pmc_$c.txt
  rm -rf $OUT/pmc_$c
done
python - <<PY
import json, re
def val(path):
    m = re.search(r"avg=\s*([0-9.]+)", open(path).read())
    return float(m.group(1)) if m else None
f, w = val("$OUT/pmc_FETCH_SIZE.txt"), val("$OUT/pmc_WRITE_SIZE.txt")
b = json.load(open("$OUT/bench_stdout.json"))
alg = b["roofline"]["algorithmic_bytes_per_launch"]
json.dump({"round": 2, "kernel": "scan_fast_kernel<HPC,noQual>", "workload": "10000000 x 10000 bp synthetic HiFi reads, one launch",
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python tools/scan_once.py 10000000",
           "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "gfx950_fetch_correction": 2.0,
           "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0 if f and w else None, "algorithmic_bytes_per_launch": alg,
           "note": "FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for coalesced streaming reads on gfx950; "
                   "WRITE_SIZE as reported (uncalibrated); one scan alone on the device",
           "reads": 10000000, "read_len": 10000}, open("$OUT/scan_traffic.json", "w"), indent=1)
PY
tools/scan_pmc.sh $OUT/pmc > /dev/null 2>&1
cp $OUT/pmc/pmc_scan.txt $OUT/pmc_scan.txt 2>/dev/null
head -c 700 $OUT/bench_stdout.json; echo; head -30 $OUT/bench_kernel_stats.txt; cat $OUT/scan_traffic.json
