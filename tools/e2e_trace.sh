#!/bin/bash
# usage: tools/e2e_trace.sh <n_reads>   -- phase timings of mdbg_tool readSelection on a FASTA in /dev/shm
set -e
N=${1:-200000}
W=$(mktemp -d /dev/shm/mdbg_tr_XXXX)
python - <<PY
import sys, os
sys.path.insert(0, "$PWD")
from metamdbg_amd import capi, synth, formats
ctx = capi.Context(0)
spec = synth.hifi_spec($N, seed=42, read_len=10000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
with open("$W/reads.fasta", "wb") as f:
    for r0 in range(0, $N, 20000):
        n = min(20000, $N - r0)
        b, o = reads.export_ascii(r0, n)
        for r in range(n):
            f.write(b">r%d\n" % (r0 + r)); f.write(b[int(o[r]):int(o[r+1])].tobytes()); f.write(b"\n")
os.makedirs("$W/tmp/filter", exist_ok=True)
formats.Parameters().save("$W/tmp/parameters.gz")
open("$W/tmp/input.txt", "w").write("$W/reads.fasta\n")
PY
for t in 4 16 32; do
  echo "== threads $t"
  MDBG_TRACE=1 ./metamdbg_amd/bin/mdbg_tool readSelection $W/tmp $W/tmp/read_data_init.txt $W/tmp/input.txt --threads $t --min-read-quality 0 2>&1 | grep "mdbg_tool\]"
done
MDBG_TRACE=1 ./metamdbg_amd/bin/mdbg_tool readSelection $W/tmp $W/tmp/read_data_init.txt $W/tmp/input.txt --threads 16 --min-read-quality 0 --batch-bases 268435456 2>&1 | grep "mdbg_tool\]"
rm -rf $W
