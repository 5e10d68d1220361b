#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel trace and/or PMC) as text for profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path: str) -> None:
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    ks = {r[0]: r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
    rows = cur.execute("select kernel_id, start, end from rocpd_kernel_dispatch").fetchall()
    stats = {}
    for kid, s, e in rows:
        d = stats.setdefault(ks.get(kid, str(kid)), [])
        d.append(e - s)
    total = sum(sum(v) for v in stats.values()) or 1
    print(f"# kernel-trace summary of {path} ({len(rows)} dispatches)")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print(f"{len(v):7d} {sum(v) / 1e6:10.3f} {sum(v) / len(v) / 1e3:10.2f} {min(v) / 1e3:10.2f} {max(v) / 1e3:10.2f} "
              f"{100.0 * sum(v) / total:6.2f}  {name[:110]}")
    # PMC counters, if any
    try:
        pmc = cur.execute(
            "select k.kernel_name, p.name, count(*), sum(e.value), avg(e.value) from rocpd_pmc_event e "
            "join rocpd_info_pmc p on p.id = e.pmc_id join rocpd_kernel_dispatch d on d.event_id = e.event_id "
            "join rocpd_info_kernel_symbol k on k.id = d.kernel_id group by k.kernel_name, p.name").fetchall()
    except sqlite3.Error as exc:
        pmc = []
        print("# (no PMC data:", exc, ")")
    if pmc:
        print("\n# PMC counters per kernel (sum over dispatches / average per dispatch)")
        for kn, pn, n, s, a in sorted(pmc, key=lambda r: (r[0], r[1])):
            print(f"{pn:>28} n={n:5d} sum={s:18.1f} avg={a:16.1f}  {kn[:90]}")


if __name__ == "__main__":
    main(sys.argv[1])
