#!/usr/bin/env python3
"""Per-kernel totals of a `rocprofv3 --kernel-trace --stats -d DIR` run (rocpd .db): calls, total / average / min / max duration in us.
usage: kernel_table.py DIR"""
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
print("%-64s %6s %11s %9s %9s %9s" % ("kernel", "calls", "total us", "avg us", "min us", "max us"))
for r in cur.execute("select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 from kernels group by name order by 3 desc limit 16"):
    print("%-64s %6d %11.1f %9.2f %9.2f %9.2f" % (r[0].split("(")[0].replace("void ", "")[:64], r[1], r[2], r[3], r[4], r[5]))
