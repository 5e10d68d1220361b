#!/bin/bash
# round 3, second GPU call: tests touched since call 1; count_insert ablations + counters; pcie / ont legs; steady-state end to end
set -u
OUT=gpurun_out/r3b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_tool.py -q -p no:cacheprovider > $OUT/pytest_multirank_tool.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_multirank_tool.log
timeout 300 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "exchange or concat or export or checksum" >> $OUT/pytest_multirank_tool.log 2>&1
tail -6 $OUT/pytest_multirank_tool.log
# ---- count_insert_kernel: what its parts cost (INSERT_ABLATE builds), 10 M and 1 M reads
: > $OUT/insert_ablation.txt
for v in "" abl1 abl2 abl3; do
  lib=metamdbg_amd/libmdbg_hip${v:+_$v}.so
  for n in 10000000 1000000; do
    MDBG_LIB=$PWD/$lib timeout 200 python tools/insert_time.py $n "${v:-full}" >> $OUT/insert_ablation.txt 2>> $OUT/insert_ablation.err
  done
done
cat $OUT/insert_ablation.txt
timeout 1500 tools/insert_pmc.sh $OUT/pmc_insert 10000000 > /dev/null 2>&1
cp $OUT/pmc_insert/pmc_insert.txt $OUT/pmc_insert.txt; cat $OUT/pmc_insert.txt
# ---- legs
timeout 900 python bench.py --steps 10 --cpu-sample 100000 --legs pcie,ont > $OUT/bench_legs.json 2> $OUT/bench_legs.err
python - <<PY
import json
d = json.load(open("$OUT/bench_legs.json"))
print("pcie", {k: v for k, v in d["legs"]["pcie"].items() if k != "workload"})
o = d["legs"]["ont"]
print("ont", {k: v for k, v in o.items() if k not in ("workload", "parity", "cpu_reference")})
print("ont parity", o.get("parity"))
PY
# ---- steady-state end to end
timeout 1500 python tools/e2e_steady.py --reads 5000000 --fastq-reads 2000000 --gz-reads 1000000 --threads 32,64 --out $OUT/e2e_steady.json > /dev/null 2> $OUT/e2e_steady.err
grep -v "^$" $OUT/e2e_steady.err | tail -12
