"""Host-side HIP API statistics from a rocprofv3 --hip-runtime-trace database: calls, mean and total duration per API name."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
for view in ("regions", "regions_and_samples"):
    if view in tabs:
        rows = db.execute(f"select name, count(*), avg(end-start), sum(end-start) from {view} group by name order by 4 desc limit 25").fetchall()
        for r in rows:
            print("%-40s n=%7d  avg %7.2f us  total %8.1f ms" % (str(r[0])[:40], r[1], r[2] / 1e3, r[3] / 1e6))
        break
else:
    print("no region view; tables:", tabs[:40])
