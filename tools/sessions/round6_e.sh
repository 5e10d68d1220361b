#!/bin/bash
set -u
O=gpurun_out/r6e; mkdir -p $O
python tools/graph_per_k.py 10000000 5,6 3 > $O/graph_per_k_w8_s8.json 2> $O/err.txt
MDBG_TOOL_UPLOAD_WORKERS=16 python tools/graph_per_k.py 10000000 6 3 > $O/graph_per_k_w16_s8.json 2>> $O/err.txt
MDBG_TOOL_UPLOAD_WORKERS=4 python tools/graph_per_k.py 10000000 6 3 > $O/graph_per_k_w4_s8.json 2>> $O/err.txt
MDBG_TOOL_UPLOAD_WORKERS=8 MDBG_TOOL_SLAB_MB=32 python tools/graph_per_k.py 10000000 6 3 > $O/graph_per_k_w8_s32.json 2>> $O/err.txt
MDBG_TOOL_UPLOAD_WORKERS=16 MDBG_TOOL_SLAB_MB=2 python tools/graph_per_k.py 10000000 6 3 > $O/graph_per_k_w16_s2.json 2>> $O/err.txt
tail -3 $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6e/graph_per_k_*.json')):
    d=json.load(open(f))
    print(f.split('/')[-1], d["all_tables_equal"])
    for k,v in d["per_k"].items(): print(' ',k, v["wall_s"], v["table_equals_in_process_pass"], v["phases_s"])
PY
