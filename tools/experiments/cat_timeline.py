#!/usr/bin/env python3
"""Kernel timeline of the last bootstrap step of tools/categorical_bench.py from a rocprofv3 --kernel-trace database: start / duration / gap to the previous kernel."""
import glob, sqlite3, sys
db = (glob.glob(sys.argv[1] + "/*.db") + glob.glob(sys.argv[1] + "/*/*.db"))[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -40:]
prev = None
for name, s, e in rows:
    print("%-60s start %10.1f us  dur %8.1f us  gap %8.1f us" % (name.split("(")[0].replace("void ", "")[:60], (s - rows[0][1]) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    prev = e
