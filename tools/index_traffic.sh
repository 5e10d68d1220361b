#!/bin/bash
# usage (GPU box, repo root): tools/index_traffic.sh <tag> [n_reads] [last_k]
# HBM traffic and L2 behaviour of the refined (k = 5) and index (k >= 6) passes over n x 10 kb HiFi reads, one context alone: separate
# rocprofv3 --pmc passes with --kernel-trace only (as /opt/skills/guides/MI355X_MICROARCH.md prescribes); tools/index_once.py runs the loop twice.
# Writes gpurun_out/<tag>/index_pmc.txt (per kernel and counter) and gpurun_out/<tag>/index_traffic.json (bytes per launch of every kernel, with the
# git blob hashes of the sources they were collected on -- what bench.py's roofline_index.per_k.*.traffic reports).
set -u
TAG=${1:-r5_index}
N=${2:-10000000}
LASTK=${3:-7}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
: > $OUT/index_pmc.txt
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum" "SQ_INSTS_VALU SQ_WAVES"; do
  n=$(echo $c | tr " " "_")
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $ROOT/$OUT/p_$n -o p -- python $ROOT/tools/index_once.py $N $LASTK > $ROOT/$OUT/index_once.log 2> $ROOT/$OUT/p_$n.err )
  if [ -f $OUT/p_$n/p_results.db ]; then
    python tools/rocpd_summary.py $OUT/p_$n/p_results.db 2>&1 | grep "n=" | grep -E "prev_|index_|refined|distinct|lookup|image|keep_|slot_flag|emit_slots|table_clear|fillBuffer|scan_reduce|scan_apply|scan_small|mark_starts" >> $OUT/index_pmc.txt
  else
    echo "# $c: no result (counter unknown to this rocprofv3?): $(tail -1 $OUT/p_$n.err)" >> $OUT/index_pmc.txt
  fi
  rm -rf $OUT/p_$n
done
python - $OUT $N $LASTK <<'PY'
import hashlib, json, re, sys
out, n, lastk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
per_kernel = {}
for line in open(f"{out}/index_pmc.txt"):
    m = re.match(r"\s*(\S+) n=\s*(\d+) sum=\s*([0-9.]+) avg=\s*([0-9.]+)\s+(\S+.*)", line)
    if not m:
        continue
    c, cnt, s, avg, kern = m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4)), m.group(5).strip()
    per_kernel.setdefault(kern[:70], {})[c] = {"launches": cnt, "avg_per_launch": avg}
def blob(path):
    d = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(d) + d).hexdigest()
summary = {}
for kern, cs in per_kernel.items():
    f = cs.get("FETCH_SIZE", {}).get("avg_per_launch")
    w = cs.get("WRITE_SIZE", {}).get("avg_per_launch")
    hit, miss = cs.get("TCC_HIT_sum", {}).get("avg_per_launch"), cs.get("TCC_MISS_sum", {}).get("avg_per_launch")
    if f is None or w is None:
        continue
    # random 32-byte slots: FETCH_SIZE counts what was fetched (no doubling: that correction is for wide coalesced streams) -> both forms are given
    summary[kern] = {"launches": cs["FETCH_SIZE"]["launches"], "fetch_bytes": f * 1024, "write_bytes": w * 1024, "traffic_bytes_uncorrected": (f + w) * 1024,
                     "traffic_bytes_fetch_doubled": (2 * f + w) * 1024, "l2_hit_rate": hit / (hit + miss) if hit is not None and miss and hit + miss > 0 else None}
json.dump({"round": 5, "reads": n, "read_len": 10000, "last_k": lastk, "workload": f"{n} x 10000 bp synthetic HiFi reads, loop k = 4 .. {lastk}, benchmark mode, one context alone, two loops (averages per launch over all k of a kernel)",
           "command": "rocprofv3 --kernel-trace --pmc <counter> (separate passes) -- python tools/index_once.py", "per_kernel": summary,
           "counters": per_kernel, "blobs": {f: blob(f"metamdbg_amd/csrc/{f}") for f in ("kminmer.hip", "table.hpp", "kminmer_dev.hpp")}},
          open(f"{out}/index_traffic.json", "w"), indent=1)
for k, v in summary.items():
    print(f"{k[:60]:60s} launches {v['launches']:3d}  fetch {v['fetch_bytes']/1e9:7.3f} GB  write {v['write_bytes']/1e9:7.3f} GB  L2 hit {v['l2_hit_rate']}")
PY
