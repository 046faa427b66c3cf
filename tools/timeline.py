"""Print one training step's kernel timeline from a rocprofv3 --kernel-trace results database.
Usage: python tools/timeline.py gpurun_out/prof_x/x_results.db [step_index [delimiter kernel]]
(step_index may be negative: from the end; the delimiter — default gather_clip_fwd — is the kernel a step starts with)"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = c.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels order by start").fetchall()
pat = sys.argv[3] if len(sys.argv) > 3 else 'gather_clip_fwd'
idx = [i for i, r in enumerate(rows) if pat in r[0]]
i0, i1 = idx[k], idx[k + 1]
t0 = rows[i0][1]
for r in rows[i0:i1]:
    n = r[0].replace("(anonymous namespace)::", "").replace("void ", "").replace("_ZN12_GLOBAL__N_1", "")[:58]
    print("%8.1f %8.1f  s%-3d %-58s g=%d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], n, r[4] // max(r[5], 1)))
print("step span %.1f us" % ((rows[i1][1] - t0) / 1e3))
