#!/bin/bash
# round 6 (GPU box): record files as bytes -- the library test, every tool test, then `graph` per k at 10 M reads with and without the launcher
set -u
O=gpurun_out/r6d; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "record_files or bad_arguments" > $O/pytest_bytes.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_bytes.log | tail -3
python -m pytest tests/test_gpu_tool.py tests/test_gpu_reference_binding.py -q -m gpu -x > $O/pytest_tool.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_tool.log | tail -5
python tools/graph_per_k.py 10000000 4,5,6,11 2 > $O/graph_per_k.json 2> $O/graph_per_k.err; tail -3 $O/graph_per_k.err
MDBG_TOOL_NO_DETACH=1 python tools/graph_per_k.py 10000000 5,11 2 > $O/graph_per_k_one_process.json 2> $O/graph_per_k_one_process.err
MDBG_TOOL_PARSE_ON_HOST=1 MDBG_TOOL_NO_DETACH=1 python tools/graph_per_k.py 10000000 5 2 > $O/graph_per_k_round5_way.json 2> $O/graph_per_k_round5_way.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6d/graph_per_k*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f,'unreadable',e); continue
    print(f.split('/')[-1], d["all_tables_equal"])
    for k,v in d["per_k"].items(): print(' ',k, v["wall_s"], v["records"], v["table_equals_in_process_pass"], v["phases_s"])
PY
