"""Copy what tools/final_profiles.sh left under gpurun_out/ (final_* and the pmc_*.json packs) to profiles/<round>_*.
Usage: python tools/collect_profiles.py r05"""
import glob
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
n = 0
for src in sorted(glob.glob(os.path.join(go, "final_*"))):
    base = os.path.basename(src)[len("final_"):]
    if base.endswith((".err", "_trace.log", "_pmc_gather.log")) or base in ("timeline.txt", "kernel_stats.txt", "kernel_stats.csv", "py.log", "ab.txt", "trace.log", "sh.log"):
        continue
    if base.startswith("pmc_") and base.endswith(".log"):
        continue
    dst = os.path.join(pr, "%s_%s" % (rnd, base.replace("pytest.log", "pytest_gpu.log")))
    if base == "pytest.log":                     # keep the tail only
        lines = open(src, errors="replace").read().splitlines()[-12:]
        open(dst, "w").write("\n".join(lines) + "\n")
    else:
        shutil.copyfile(src, dst)
    n += 1
for tag, name in (("score_fwd_ce_n3", "pmc_score_fwd_ce_n3"), ("score_dx_onehot_n1", "pmc_score_dx_onehot_n1"), ("score_dE_qz_n1", "pmc_score_dE_qz_n1"),
                  ("gather_fwd", "pmc_gather_fwd")):
    src = os.path.join(go, "pmc_%s.json" % tag)
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(pr, "%s_%s.json" % (rnd, name)))
        n += 1
print("copied %d files to profiles/%s_*" % (n, rnd))
