#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (sqlite) outputs as text: per-kernel time statistics from a
--kernel-trace run and per-kernel counter sums from a --pmc run.

    python tools/rocpd_summary.py stats  gpurun_out/<tag>/prof_c2/c2_results.db
    python tools/rocpd_summary.py pmc    gpurun_out/<tag>/pmc_c2_FETCH_SIZE/c2_results.db
"""
import sqlite3
import sys


def short(name):
    name = name.split("(")[0]
    return name if len(name) <= 70 else name[:67] + "..."


def stats(db):
    con = sqlite3.connect(db)
    rows = con.execute('select name, count(*), sum("end" - start), avg("end" - start), '
                       'min("end" - start), max("end" - start) from kernels group by name '
                       'order by 3 desc').fetchall()
    total = sum(r[2] for r in rows)
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for n, c, tot, avg, mn, mx in rows:
        print(f'"{short(n)}",{c},{tot},{avg:.1f},{100.0 * tot / total:.4f},{mn},{mx}')


def pmc(db):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = con.execute(f'select {name_col}, counter_name, count(distinct dispatch_id), sum(value) '
                       f'from counters_collection group by 1, 2 order by 4 desc').fetchall()
    print('"Kernel","Counter","Dispatches","Sum","PerDispatch"')
    for n, c, d, v in rows:
        print(f'"{short(n)}","{c}",{d},{v:.0f},{v / max(d, 1):.1f}')


def timeline(db, first="gain_prepare_stream", nth=-1):
    """Kernels from the nth launch of `first` to the next one, with the idle time before each:
    where a step's wall time goes that the per-kernel sums do not show."""
    con = sqlite3.connect(db)
    rows = con.execute('select name, start, "end" from kernels order by start').fetchall()
    marks = [i for i, r in enumerate(rows) if first in r[0]]
    j = int(nth) % len(marks)
    a = marks[j]
    b = marks[j + 1] if j + 1 < len(marks) else len(rows)
    t0, busy, idle, prev_end = rows[a][1], 0, 0, rows[a][1]
    agg = {}
    for n, st, en in rows[a:b]:
        gap = max(0, st - prev_end)
        k = short(n)
        e = agg.setdefault(k, [0, 0, 0])
        e[0] += 1; e[1] += en - st; e[2] += gap
        busy += en - st; idle += gap; prev_end = max(prev_end, en)
    print(f"step of {b - a} kernels: wall {1e-6 * (prev_end - t0):.3f} ms, busy {1e-6 * busy:.3f} ms, "
          f"idle between kernels {1e-6 * idle:.3f} ms")
    print('"Name","Calls","BusyUs","IdleBeforeUs"')
    for k, (c, bu, gp) in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
        print(f'"{k}",{c},{bu / 1e3:.1f},{gp / 1e3:.1f}')


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "timeline": timeline}[sys.argv[1]](*sys.argv[2:])
