cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hp; timeout 300 rocprofv3 --kernel-trace -d /tmp/hp -o hp -- python $GRAFT_REPO_ROOT/tools/experiments/hoc_timeline.py 2>/dev/null | tail -1
python - <<'P'
import glob, sqlite3
db = (glob.glob("/tmp/hp/*.db") + glob.glob("/tmp/hp/*/*.db"))[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
# the last call = after the largest gap (the 50 ms sleep)
gaps = [(rows[i][1] - rows[i-1][2], i) for i in range(1, len(rows))]
g, i0 = max(gaps)
rows = rows[i0:]
t0 = rows[0][1]
print("kernels in the last call:", len(rows), "span ms", (rows[-1][2] - t0) / 1e6)
prev = None
agg = {}
for name, s, e in rows:
    n = name.split("(")[0].replace("void ", "")[:50]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    if (e - s) > 300e3 or (prev is not None and s - prev > 200e3):
        print("%-50s start %9.1f us dur %9.1f us gap %8.1f us" % (n, (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    prev = e
print("--- totals")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-50s calls %5d total %10.1f us" % (n, c, t))
P
python - <<'P'
import glob, sqlite3
db = (glob.glob("/tmp/hp/*.db") + glob.glob("/tmp/hp/*/*.db"))[0]
rows = list(sqlite3.connect(db).cursor().execute("select name, start, end from kernels order by start"))[-21:]
t0 = rows[0][1]; prev = None
print("--- the last three trips of the second stage")
for name, s, e in rows:
    print("%-44s start %8.1f us dur %7.1f us gap %6.1f us" % (name.split("(")[0].replace("void ", "")[:44], (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    prev = e
P
