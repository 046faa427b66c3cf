"""Pack the four PMC passes of tools/pmc_gemm.sh <mode> <nsplit> into one summary:
gpurun_out/pmc_<tag>_n<nsplit>.json (copied to profiles/r02_pmc_<tag>_n<nsplit>.json, which bench.py's roofline.traffic reads).
HBM bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE (KB counters; gfx950 reports half of the bytes of wide coalesced reads —
MI355X_MICROARCH.md, HBM section; WRITE_SIZE is exact)."""
import json, os, sys
mode, ns = sys.argv[1], int(sys.argv[2])
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
N, B, EK = int(os.environ.get("GB_N", 46033)), 512, 832
Npad = (N + 127) // 128 * 128
tag = {"fwd": "score_fwd", "fwdce": "score_fwd_ce_classic", "fwdce2": "score_fwd_ce", "dx": "score_dx", "de": "score_dE",
       "dx2": "score_dx_onehot", "de2": "score_dE_qz"}[mode]
get = lambda s: json.load(open(os.path.join(root, "gpurun_out", "pmc_%s_n%d_%s.json" % (mode, ns, s))))
f, w, q, t = get("FETCH_SIZE"), get("WRITE_SIZE"), get("SQ_VALU_MFMA_BUSY_CYCLES"), get("TCC_HIT_sum")
planes = 2 if ns == 3 else 1
alg = {"fwd": planes * 2 * (Npad * EK + B * EK) + 4 * B * Npad,                       # E + attout planes in, fp32 logits out
       "fwdce": planes * 2 * (Npad * EK + B * EK) + 2 * B * Npad + 8 * B * (Npad // 96),   # ... bf16 exp plane + group statistics out
       # one-hot form: item | content planes (512 columns, hi + lo) of E and attout, the one-hot plane (160 columns, ONE plane) and
       # the time-score planes (160 columns, hi + lo) in; bf16 exp plane + group statistics out
       "fwdce2": 4 * (Npad * 512 + B * 512) + 2 * Npad * 160 + 4 * B * 160 + 2 * B * Npad + 8 * B * (Npad // 96),
       "dx": planes * 2 * (B * Npad + Npad * EK) + 4 * int(os.environ.get("GB_SPLITK", 18)) * B * EK,   # dlogits + E planes in, slabs out
       "de": planes * 2 * (B * Npad + B * 576) + 4 * N * 576,                        # dlogits + packed attout planes in, dE out
       # one-hot form (hi planes): dlogits + item | content planes of E + the one-hot plane in, slabs of 672 columns out
       "dx2": 2 * (B * Npad + Npad * 512 + Npad * 160) + 4 * int(os.environ.get("GB_SPLITK", 18)) * B * 672,
       # dlogits + packed attout planes in; item block of dE + 5 (q, z) pairs per candidate out
       "de2": 2 * (B * Npad + B * 576) + 4 * N * 256 + 8 * 5 * N}[mode]
flops = {"fwd": 2.0 * B * N * 820, "fwdce": 2.0 * B * N * 820, "fwdce2": 2.0 * B * N * 820, "dx": 2.0 * B * N * 820, "de": 2.0 * B * N * 570,
         "dx2": 2.0 * B * N * 820, "de2": 2.0 * B * N * 570}[mode]
hbm = int(2 * f["FETCH_SIZE"] * 1024 + w["WRITE_SIZE"] * 1024)
dur = q.get("avg_duration_us") or f.get("avg_duration_us")
out = {"tag": tag, "nsplit": ns, "shape_N_B": [N, B],
       "command": "tools/pmc_gemm.sh %s %d  (rocprofv3 --kernel-trace --pmc <set> over tools/gemm_bench.py %s %d 5; FETCH_SIZE, "
                  "WRITE_SIZE, the SQ set and the TCC set in four separate passes; the GEMM runs ALONE)" % (mode, ns, mode, ns),
       "FETCH_SIZE_KB": f["FETCH_SIZE"], "WRITE_SIZE_KB": w["WRITE_SIZE"],
       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> x2; WRITE_SIZE exact",
       "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round(hbm / alg, 3),
       "avg_duration_us_alone": dur,
       "algorithmic_tflops_alone": round(flops / (dur * 1e-6) / 1e12, 1) if dur else None,
       "hbm_GBps_alone": round(hbm / (dur * 1e-6) / 1e9, 1) if dur else None,
       "sq": {k: v for k, v in q.items() if k.startswith("SQ_") or k.startswith("GRBM")},
       "l2": {"TCC_HIT_sum": t.get("TCC_HIT_sum"), "TCC_MISS_sum": t.get("TCC_MISS_sum"),
              "hit_rate": round(t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]), 3) if t.get("TCC_HIT_sum") else None}}
sq = out["sq"]
if sq.get("SQ_VALU_MFMA_BUSY_CYCLES") and sq.get("SQ_BUSY_CYCLES"):
    # busy cycles are summed over SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs: SIMD cycles available = GUI_ACTIVE/8 * 256 CUs * 4 SIMDs
    if sq.get("GRBM_GUI_ACTIVE"):
        out["mfma_busy_fraction_of_simd_cycles"] = round(sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq["GRBM_GUI_ACTIVE"] / 8 * 256 * 4), 3)
    if sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_WAIT_ANY"):
        out["wave_cycles_waiting_fraction"] = round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 3)
p = os.path.join(root, "gpurun_out", "pmc_%s_n%d.json" % (tag, ns))
json.dump(out, open(p, "w"), indent=1)
print(json.dumps(out))
