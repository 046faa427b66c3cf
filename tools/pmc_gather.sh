#!/bin/bash
# PMC passes over the embedding gather's throughput form (gather_clip_fwd_big_kernel) at 655,360 rows: FETCH_SIZE and WRITE_SIZE in
# separate runs, kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3); packed into gpurun_out/pmc_gather_fwd.json
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  OUT=$ROOT/gpurun_out/pmc_gather_$SET
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT -- python3 $ROOT/tools/gather_bench.py > $OUT.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT gather_clip_fwd_big_kernel 2 > $ROOT/gpurun_out/pmc_gather_$SET.json 2>> $OUT.log
  find $OUT -name "*.csv" ! -name "*counter_collection.csv" -delete
done
python3 - <<PY
import json, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
f = json.load(open(root + "/gpurun_out/pmc_gather_FETCH_SIZE.json")); w = json.load(open(root + "/gpurun_out/pmc_gather_WRITE_SIZE.json"))
B, T = 16384, 40
alg_r = alg_w = B * (3536.0 * T + 512)
# bytes that must come from / go to HBM: item + content rows (2 x 1 KB incl. pads per click) are gathered from a 2 M-row table;
# the six small tables (position, five time tables, dwell: 1,536 B per click algorithmically) stay in LDS / L2
hbm_r = 2 * f["FETCH_SIZE"] * 1024
hbm_w = w["WRITE_SIZE"] * 1024
dur = f.get("avg_duration_us") or w.get("avg_duration_us")
out = {"kernel": "gather_clip_fwd_big_kernel (embed.hip), tools/gather_bench.py: 2,000,000 items, B = 16384, T = 40 -> 655,360 rows",
       "command": "tools/pmc_gather.sh  (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes)",
       "avg_duration_us": dur, "FETCH_SIZE_KB": f["FETCH_SIZE"], "WRITE_SIZE_KB": w["WRITE_SIZE"],
       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) reads -> x2; WRITE_SIZE exact",
       "hbm_read_bytes": hbm_r, "hbm_write_bytes": hbm_w,
       "algorithmic_read_bytes": alg_r, "algorithmic_write_bytes": alg_w,
       "hbm_TBps": round((hbm_r + hbm_w) / (dur * 1e-6) / 1e12, 3), "frac_hbm_real_bytes": round((hbm_r + hbm_w) / (dur * 1e-6) / 8e12, 4),
       "algorithmic_TBps": round((alg_r + alg_w) / (dur * 1e-6) / 1e12, 3), "frac_hbm_algorithmic": round((alg_r + alg_w) / (dur * 1e-6) / 8e12, 4)}
json.dump(out, open(root + "/gpurun_out/pmc_gather_fwd.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
