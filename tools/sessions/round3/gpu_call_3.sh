#!/bin/bash
# round 3, third GPU call: the reworked tool (tests + steady-state end to end, one and two consumers)
set -u
OUT=gpurun_out/r3c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tool.py tests/test_gpu_multirank.py -q -p no:cacheprovider > $OUT/pytest_tool.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_tool.log
tail -5 $OUT/pytest_tool.log
timeout 1200 python tools/e2e_steady.py --reads 5000000 --fastq-reads 2000000 --gz-reads 0 --threads 32 --out $OUT/e2e_steady.json > /dev/null 2> $OUT/e2e_steady.err
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
