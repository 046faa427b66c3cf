"""Per-step period (gather start to gather start) and the duration of a few named kernels for EVERY step of a rocprofv3 --kernel-trace
database: where does a short run (the driver's 20 steps / 5 warm-up) lose time against the 200-step figure?
Usage: python tools/step_periods.py <results.db> [first [last]]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'gather_clip_fwd' in r[0]]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
last = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) - 1
watch = ("gemm_bf16_kernel<0, 0, 3", "gemm_bf16_kernel<0, 1, 1", "gemm_bf16_kernel<1, 1, 1", "clip_adam_rest", "ce_fold_rescale", "clip_adam_early")
print("step  period_us  " + "  ".join(w[-14:] for w in watch))
for k in range(first, min(last, len(idx) - 1)):
    i0, i1 = idx[k], idx[k + 1]
    d = {w: 0.0 for w in watch}
    for r in rows[i0:i1]:
        for w in watch:
            if w in r[0]:
                d[w] += (r[2] - r[1]) / 1e3
    print("%4d %9.1f  " % (k, (rows[i1][1] - rows[i0][1]) / 1e3) + "  ".join("%14.1f" % d[w] for w in watch))
