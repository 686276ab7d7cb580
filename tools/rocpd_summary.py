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


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
