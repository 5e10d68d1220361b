#!/bin/bash
# round 6 (GPU box): the block-wise generic window hash -- parity (every table test, the deep-k fixtures), then the loop to k = 70.
set -u
O=gpurun_out/r6c; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_partition.py -q -m gpu -x -k "not full_size and not 1m and not extreme" > $O/pytest_tables.log 2>&1; grep -n "passed\|failed" $O/pytest_tables.log
timeout 900 python tools/index_by_k.py 10000000 70 -1 > $O/index_by_k_deep.json 2>$O/index_by_k_deep.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6c/index_by_k_*.json')):
    d=json.load(open(f))
    for t,v in d['tunings'].items():
        ks=v['per_k']
        print(f.split('/')[-1], 'tuning',t,'loop',v['loop_ms_incl_first_pass'], ' '.join(f"k{k}:{ks[k]['wall_ms']:.1f}/{ks[k]['kernel_ms'].get('kminmer_prev_lookup',0):.1f}+{ks[k]['kernel_ms'].get('kminmer_insert',0):.1f}" for k in ks if int(k)<50))
PY
