#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 run that wrote its default output (a rocpd SQLite database) instead of CSV:

    rocprofv3 --kernel-trace --stats -d out -o name -- python3 tool.py ...      ->  out/name_results.db
    python tools/rocprof_db_stats.py out/name_results.db [more.db ...]

(`--output-format csv` gives the same table as *_kernel_stats.csv directly; profiles/ holds those.)"""
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by 4 desc").fetchall()
    print(path, "total kernel ms", sum(r[3] for r in rows) / 1e6)
    for name, calls, avg, total in rows[:22]:
        print(f"{calls:5d} {avg / 1e3:10.1f} us  {total / 1e6:9.2f} ms  {name[:100]}")
