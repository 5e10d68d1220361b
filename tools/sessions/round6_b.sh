#!/bin/bash
# round 6, second table run (GPU box): parity, the ablation (fixed), the loop by tuning at the default load, deep k.
set -u
O=gpurun_out/r6b; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_partition.py -q -m gpu -x -k "not full_size and not 1m and not extreme" > $O/pytest_tables.log 2>&1; grep -n "passed\|failed" $O/pytest_tables.log
for l in 0.12 0.16 0.22 0.30 0.40 0.50; do timeout 120 tools/ubench/lookup_ablate 10000000 6 14000000 $l brief; done > $O/lookup_ablate_load_sweep.txt 2>&1
timeout 200 tools/ubench/lookup_ablate 10000000 6 14000000 0.22 > $O/lookup_ablate_full.txt 2>&1
timeout 600 python tools/index_by_k.py 10000000 11 19,3,83 > $O/index_by_k_tunings.json 2>$O/index_by_k_tunings.err
timeout 900 python tools/index_by_k.py 10000000 70 -1 > $O/index_by_k_deep.json 2>$O/index_by_k_deep.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6b/index_by_k_*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable', e); continue
    for t,v in d['tunings'].items():
        ks=v['per_k']
        print(f.split('/')[-1], 'tuning',t,'loop',v['loop_ms_incl_first_pass'], 'equal',v['tables_equal_to_first_tuning'], ' '.join(f"k{k}:{ks[k]['wall_ms']:.1f}/{ks[k]['kernel_ms'].get('kminmer_prev_lookup',0):.1f}+{ks[k]['kernel_ms'].get('kminmer_insert',0):.1f}" for k in ks))
PY
