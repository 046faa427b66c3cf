"""Per-kernel statistics (calls, average, share) from a rocprofv3 --kernel-trace results database; optional CSV output.
Usage: python tools/kstats.py gpurun_out/prof_x/x_results.db [out.csv] [last_steps]
last_steps: only the kernels of the LAST that many training steps of the trace (a step starts at its forward gather launch) — e.g.
the ms_per_step_by_T loop that ends a bench.py run."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
where = ""
if len(sys.argv) > 3 and int(sys.argv[3]) > 0:
    starts = [r[0] for r in c.execute("select start from kernels where name like '%gather_clip_fwd%' order by start").fetchall()]
    k = min(int(sys.argv[3]), len(starts))
    if k > 0:
        where = " where start >= %d" % starts[-k]
rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels" + where +
                 " group by name order by 6 desc").fetchall()
tot = sum(r[5] for r in rows)
lines = ["Name,Calls,AverageNs,MinNs,MaxNs,TotalDurationNs,Percentage"]
for r in rows:
    lines.append('"%s",%d,%.0f,%d,%d,%d,%.2f' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[5] / tot))
if len(sys.argv) > 2 and sys.argv[2] not in ("", "-"):
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
if where:
    print("(kernels of the last %s steps of the trace)" % sys.argv[3])
for r in rows[:40]:
    n = r[0].replace("(anonymous namespace)::", "").replace("void ", "").replace("_ZN12_GLOBAL__N_1", "")[:70]
    print("%-70s %5d %8.1f us %5.1f%%" % (n, r[1], r[2] / 1e3, 100.0 * r[5] / tot))
