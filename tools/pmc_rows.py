"""Print per-kernel averages of the counters in rocprofv3 --pmc output databases.  usage: pmc_rows.py <dir> [kernel substring ...]"""
import glob, sqlite3, sys
keys = sys.argv[2:] or ["solver", "gram_i8", "resample"]
for db in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection c where grid_size = "
         "(select max(grid_size) from counters_collection c2 where c2.kernel_name = c.kernel_name) group by kernel_name, counter_name")
    for r in cur.execute(q):
        if any(k in r[0] for k in keys):
            print("%-28s %-28s n=%-3d avg=%-16.1f dur_us=%.1f" % (r[0].split("(")[0].replace("void ", "")[:28], r[1], r[2], r[3], r[4] / 1e3))
