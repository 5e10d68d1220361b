#!/bin/bash
# round 3, fourth GPU call: whole GPU suite on the current tree; reads with N routed one by one (timing); steady-state end to end
set -u
OUT=gpurun_out/r3d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log
timeout 600 python tools/scan_with_n_time.py 400000 1000 > $OUT/scan_with_n.txt 2> $OUT/scan_with_n.err
MDBG_SCAN_NO_FAST=1 timeout 600 python tools/scan_with_n_time.py 400000 1000 2>> $OUT/scan_with_n.err | sed 's/^/MDBG_SCAN_NO_FAST=1  /' >> $OUT/scan_with_n.txt
cat $OUT/scan_with_n.txt; tail -3 $OUT/scan_with_n.err
timeout 1200 python tools/e2e_steady.py --reads 5000000 --fastq-reads 2000000 --gz-reads 0 --threads 32,64 --out $OUT/e2e_steady.json > /dev/null 2> $OUT/e2e_steady.err
grep -v "^$" $OUT/e2e_steady.err | tail -6
python - <<PY
import json
d = json.load(open("$OUT/e2e_steady.json"))
for sec in ("fasta", "fastq"):
    for t, v in d[sec].items():
        print(sec, t, v["all_total_s"])
        for ln in v["trace_read_selection"] + v["trace_graph"]:
            print("   ", ln)
PY
