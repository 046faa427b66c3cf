"""Mean idle time of the main stream in FRONT of selected kernels over every step of a rocprofv3 --kernel-trace database: how long
after the previous kernel of the same stream ended did this one start?  (Barrier packets, event waits and flag polls show up here.)
Usage: python tools/gap_stats.py <results.db>"""
import sqlite3, sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
main = None
for r in rows:
    if "gather_clip_fwd" in r[0]:
        main = r[3]
        break
last_end, last_name = {}, {}
gaps = {}
for name, s, e, st in rows:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if st == main and st in last_end:
        gaps.setdefault(short[:46], []).append(((s - last_end[st]) / 1e3, last_name[st][:30]))
    last_end[st], last_name[st] = e, short
print("%-46s %6s %8s %8s %8s %8s   %s" % ("kernel on the main stream", "n", "mean us", "p50 us", "p90 us", "max us", "usual predecessor"))
for k, v in sorted(gaps.items(), key=lambda kv: -np.mean([x[0] for x in kv[1]]) * len(kv[1])):
    g = np.array([x[0] for x in v])
    if len(g) < 50:
        continue
    names = [x[1] for x in v]
    pred = max(set(names), key=names.count)
    print("%-46s %6d %8.2f %8.2f %8.2f %8.1f   %s" % (k, len(g), g.mean(), np.median(g), np.percentile(g, 90), g.max(), pred))
