"""Gap between the end of a variant kernel's predecessors and its start, per variant (median over the repetitions), from a
rocprofv3 --kernel-trace database of tools/micro/event_cost."""
import sqlite3, sys, statistics
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
PRED = {"b_wait_live": ("spin_a", "c_other"), "b_ahead_live": ("spin_a", "c_other"), "b_flag_join": ("spin_a", "c_signalling"),
        "b_flag_fork": ("c_signalling",), "c_flag_consumer": ("c_signalling",), "b_ahead_done": ("spin_a",)}
last_end, gaps = {}, {}
for name, st, en in rows:
    nm = name.split("(")[0]
    if nm.startswith("b_") or nm == "c_flag_consumer":
        pred = PRED.get(nm, ("spin_a",))
        gaps.setdefault(nm, []).append((st - max(last_end[p] for p in pred)) / 1e3)
    last_end[nm] = en
for k, v in gaps.items():
    print("%-18s starts after its predecessors %s: median %6.2f us  min %6.2f  max %6.2f  (n=%d)" % (k, "+".join(PRED.get(k, ("spin_a",))), statistics.median(v), min(v), max(v), len(v)))
