#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace rocpd database of a bench.py run: what happens between one scan launch and the next.
    python tools/scan_gaps.py <results.db> [first_launch_to_print]
Per scan launch: duration, the gap until the next scan kernel starts, and the kernels that ran inside that gap (name, queue,
start relative to the end of the scan, duration), to see what a step waits for beyond its scan."""
import sqlite3
import sys


def main(path: str, skip: int = 0) -> None:
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    ks = {r[0]: r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = cur.execute(f"select kernel_id, start, end{', ' + qcol if qcol else ''} from rocpd_kernel_dispatch order by start").fetchall()
    scans = [r for r in rows if "scan_fast_kernel" in ks.get(r[0], "")]
    print(f"# {len(rows)} dispatches, {len(scans)} scan launches; columns of rocpd_kernel_dispatch: {cols}")
    gaps = []
    for i in range(len(scans) - 1):
        a, b = scans[i], scans[i + 1]
        gap = (b[1] - a[2]) / 1e6
        gaps.append(gap)
        if i < skip or i >= skip + 6:
            continue
        print(f"\nscan {i}: {(a[2] - a[1]) / 1e6:.2f} ms, gap to the next scan {gap:.2f} ms, start-to-start {(b[1] - a[1]) / 1e6:.2f} ms")
        inside = [r for r in rows if r[2] > a[2] and r[1] < b[1] and r is not a and r is not b]
        for r in inside[:60]:
            print(f"    {(r[1] - a[2]) / 1e6:8.3f} ms  +{(r[2] - r[1]) / 1e6:8.3f} ms  q={r[3] if qcol else '-'}  {ks.get(r[0], '?')[:70]}")
    if gaps:
        g = sorted(gaps)
        print(f"\n# gaps: mean {sum(g) / len(g):.2f} ms, median {g[len(g) // 2]:.2f}, max {g[-1]:.2f}; scan mean {sum((s[2] - s[1]) for s in scans) / len(scans) / 1e6:.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
