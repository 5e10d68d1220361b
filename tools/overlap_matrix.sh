#!/bin/bash
# usage (GPU box, repo root): tools/overlap_matrix.sh <tag>  -- short default-workload bench runs under the knobs that decide whether the
# other batches' table kernels can be RESIDENT beside a scan (LDS, registers, wave slots): one line per setting in <tag>/overlap_matrix.txt
set -u
TAG=${1:-r4_overlap}
OUT=gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/overlap_matrix.txt
run() {   # label, in-flight, env...
  label=$1; nf=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 3 --legs none --cpu-sample 0 --in-flight $nf > $OUT/line.json 2> $OUT/err.log
  python - "$label" "$nf" "$*" <<'PY' >> $OUT/overlap_matrix.txt
import json, sys
try:
    d = json.loads(open("gpurun_out/%s/line.json" % __import__("os").environ.get("TAGDIR", "")).read().strip().splitlines()[-1])
    k = d["kernel_ms_per_step"]
    print(f"{sys.argv[1]:34s} in-flight {sys.argv[2]}  {d['value']:7.1f} Gbp/s  step {d['ms_per_step']:7.2f} ms  scan {k['scan']:7.2f}  purge {k['purge_palindromes']:6.2f}  "
          f"split {k.get('kminmer_split', 0):6.2f}  insert {k['kminmer_insert']:6.2f}  rescue {k['kminmer_rescue']:5.2f}  emit {k['kminmer_emit']:5.2f}  scans {k['prefix_scan']:5.2f}   [{sys.argv[3]}]")
except Exception as exc:
    print(f"{sys.argv[1]:34s} in-flight {sys.argv[2]}  FAILED: {exc}")
PY
}
export TAGDIR=$TAG
if [ "${2:-}" = "variance" ]; then
  for rep in 1 2 3; do
    run "partitioned, as is"               3 MDBG_X=0
    run "partitioned, as is"               2 MDBG_X=0
    run "tile 2048 + no list, no pad"      3 MDBG_PARTITION_TILE=2048 MDBG_PARTITION_SLOT_LIST=0
    run "tile 2048 + no list, no pad"      2 MDBG_PARTITION_TILE=2048 MDBG_PARTITION_SLOT_LIST=0
    run "pad 3072 + tile 2048 + no list"   3 MDBG_SCAN_LDS_PAD=3072 MDBG_PARTITION_TILE=2048 MDBG_PARTITION_SLOT_LIST=0
    run "pad 3072 + tile 2048 + no list"   2 MDBG_SCAN_LDS_PAD=3072 MDBG_PARTITION_TILE=2048 MDBG_PARTITION_SLOT_LIST=0
  done
  sort $OUT/overlap_matrix.txt
  exit 0
fi
run "r3 one-table path"                3 MDBG_FIRST_PASS_MODE=1
run "partitioned, as is"               3 MDBG_X=0
run "partitioned, as is"               1 MDBG_X=0
run "partitioned, as is"               2 MDBG_X=0
run "scan 4 blocks/CU (pad 3072)"      3 MDBG_SCAN_LDS_PAD=3072
run "pad 3072 + tile 2048 + no list"   3 MDBG_SCAN_LDS_PAD=3072 MDBG_PARTITION_TILE=2048 MDBG_PARTITION_SLOT_LIST=0
run "pad 3072 + tile 2048 + no list"   2 MDBG_SCAN_LDS_PAD=3072 MDBG_PARTITION_TILE=2048 MDBG_PARTITION_SLOT_LIST=0
run "pad 3072 + tile 2048 + no list"   1 MDBG_SCAN_LDS_PAD=3072 MDBG_PARTITION_TILE=2048 MDBG_PARTITION_SLOT_LIST=0
run "tile 2048 + no list, no pad"      3 MDBG_PARTITION_TILE=2048 MDBG_PARTITION_SLOT_LIST=0
run "pad 3072, one-table path"         3 MDBG_SCAN_LDS_PAD=3072 MDBG_FIRST_PASS_MODE=1
run "scan 3 blocks/CU (pad 10240+1)"   3 MDBG_SCAN_LDS_PAD=10304 MDBG_PARTITION_TILE=2048 MDBG_PARTITION_SLOT_LIST=0
cat $OUT/overlap_matrix.txt
