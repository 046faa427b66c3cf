"""Average duration of one kernel, split by the grid size of the step's gather_clip_fwd launch (a proxy for the session
length T of the step).  Usage: python tools/kfilter.py results.db kernel_substring"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'gather_clip_fwd' in r[0]]
acc = collections.defaultdict(list)
for k in range(10, len(idx) - 1):
    g = rows[idx[k]][3] // rows[idx[k]][4]
    for r in rows[idx[k]:idx[k + 1]]:
        if sys.argv[2] in r[0]:
            acc[g].append((r[2] - r[1]) / 1e3)
for g in sorted(acc):
    print("fwd-grid %5d: n=%3d avg %.1f us" % (g, len(acc[g]), sum(acc[g]) / len(acc[g])))
