#!/usr/bin/env python3
"""bench.py — sessions/sec of TCAR training on the Globo-like configuration (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path (forward, loss, backward, per-variable clip, Adam) over one mini-batch of
B = 512 sessions whose int32 feed arrays are already resident in HBM.  Prints ONE JSON line (rank 0) with the
whole-job sessions/sec plus
  "roofline"      dominant kernel (full-catalog scoring GEMM, gemm_f32_kernel<0,0>): algorithmic FLOPs per launch
                  / mean HIP-event duration of that launch inside the timed region, against the gfx950 fp32
                  matrix peak of /opt/skills/guides/MI355X_MICROARCH.md;
  "cpu_baseline"  the CPU oracle (PyTorch-CPU fp32 restatement of the reference graph) on the host cores, on a
                  bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: Peak FP32 (matrix), v_mfma_f32_32x32x2_f32
PEAK_BF16_DENSE_TFLOPS = 2500.0     # MI355X_MICROARCH.md: Peak BF16 MFMA, dense
PEAK_HBM_GBS = 8000.0               # HBM3E spec peak


def build_batches(fold, n_batches, B, K, rng):
    """Full batches of exactly B sessions in the fold's own length mix (sampler.py:40-49 bucketing)."""
    st = fold.train
    out = []
    by_len = {}
    for T in np.unique(st.in_len):
        idx = np.where(st.in_len == T)[0]
        rng.shuffle(idx)
        for i in range(0, len(idx) - B + 1, B):
            by_len.setdefault(int(T), []).append(idx[i:i + B])
    flat = [(T, ids) for T, lst in by_len.items() for ids in lst]
    order = rng.permutation(len(flat))
    for j in order[:n_batches]:
        T, ids = flat[j]
        b = st.batch_arrays(ids, "click_delta")
        b["neg"] = rng.randint(0, fold.n_items, size=(B, K)).astype(np.int32)
        out.append(b)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch_size", type=int, default=512)
    ap.add_argument("--n_items", type=int, default=46033)
    ap.add_argument("--hidden_size", type=int, default=250)
    ap.add_argument("--time_hidden_size", type=int, default=64)
    ap.add_argument("--neg_num", type=int, default=20)
    ap.add_argument("--n_batches", type=int, default=48)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_steps", type=int, default=20)
    ap.add_argument("--cpu_threads", type=int, default=16)
    ap.add_argument("--no_kernel_timing", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU dry runs)")
    ap.add_argument("--same_device", action="store_true", help="dry run: put every rank on cuda:0")
    ap.add_argument("--scoring", default="bf16x3", choices=["f32", "bf16x3", "bf16x3-mixed", "bf16"],
                    help="precision of the full-catalog scoring GEMMs (bf16x3 = split-bf16 planes, fp32-class accuracy)")
    args = ap.parse_args()

    import torch
    import tcar_amd  # noqa: F401
    from tcar_amd.host.synth import SynthFold

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    B, K = args.batch_size, args.neg_num
    # every rank builds the same catalog (seed 2020) and its own shard of sessions (weak scaling: B per GPU)
    fold = SynthFold(n_items=args.n_items, dim=args.hidden_size, n_train=max(60000, 4 * B * args.n_batches),
                     n_test=1000, seed=2020)
    rng = np.random.RandomState(2020 + rank)
    batches = build_batches(fold, args.n_batches, B, K, rng)
    if world > 1:
        # ranks step in lock-step over buckets of the same length T: share rank 0's T schedule
        sched = torch.tensor([b["seq"].shape[1] for b in batches], device=dev)
        dist.broadcast(sched, 0)
        want = sched.cpu().tolist()
        pool = {}
        for b in build_batches(fold, 10 ** 9, B, K, rng):
            pool.setdefault(b["seq"].shape[1], []).append(b)
        batches = [pool[T][i % len(pool[T])] for i, T in enumerate(want)]

    from tcar_amd.host.model import initial_variables     # the product's own initialiser (modules.py:32-34,50-51)
    np.random.seed(2020)
    params = initial_variables(args.n_items, args.hidden_size, args.time_hidden_size, 0.002, 0.05, weight_seed=2020)
    if world > 1 or os.environ.get("TCAR_FORCE_DP"):       # TCAR_FORCE_DP=1: time the data-parallel code path on one rank
        from tcar_amd.dp import DPEngine
        eng = DPEngine(params, fold.content, fold.mwdhm, device=dev, group=(dist.group.WORLD if dist is not None else None), scoring=args.scoring)
    else:
        from tcar_amd.engine import TcarEngine
        eng = TcarEngine(params, fold.content, fold.mwdhm, device=dev, scoring=args.scoring)
    resident = [eng.make_resident(b) for b in batches]
    mean_T = float(np.mean([b["seq"].shape[1] for b in batches]))

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # TCAR_DEFER=1 (experiment): defer each step's Adam into the next step's forward pass (engine.train_step docstring);
    # the last update is flushed INSIDE the timed region, so K timed steps contain exactly K updates.  Measured neutral.
    defer = {"defer_update": True} if (os.environ.get("TCAR_DEFER") and world == 1 and not os.environ.get("TCAR_FORCE_DP")) else {}
    for i in range(args.warmup):
        eng.train_step(None, bt=resident[i % len(resident)], **defer)
    tags = ["score_fwd", "score_dx", "score_dE", "weight_grads", "gather_fwd", "softmax_ce", "adam_item"]
    if not args.no_kernel_timing:
        eng.enable_native_timing(args.steps)     # HIP events around the logits GEMM inside tcar_train_step
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.train_step(None, bt=resident[(args.warmup + i) % len(resident)], **defer)
    eng.flush()
    t_enq = time.perf_counter() - t0          # host time to enqueue every step (the loop never synchronises)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    last_loss = float(eng.loss[:B].mean())
    value = B * world * args.steps / dt

    kern = {}
    if not args.no_kernel_timing:
        # dominant kernel: timed live inside the timed region (events recorded by the C++ step driver);
        # the other kernels: a short extra pass through the Python-sequenced path with per-launch events
        ms = eng.native_timing_ms()
        eng._ev = None
        if args.scoring == "f32" and world == 1:
            eng.enable_timing(tags)
            for i in range(min(20, args.steps)):
                eng.train_step(None, bt=resident[i % len(resident)])
            sync()
            kern = eng.timing_summary()
            eng.timing = None
        kern["score_fwd"] = (len(ms), float(np.mean(ms)))
    N, H = args.n_items, args.hidden_size
    k_alg = 2 * H + 5 * args.time_hidden_size                      # 820 contraction length (model_combine.py:132-138)
    flops = {"score_fwd": 2.0 * B * N * k_alg, "score_dx": 2.0 * B * N * k_alg,
             "score_dE": 2.0 * B * N * (H + 5 * args.time_hidden_size)}
    roof = None
    kernels = {}
    for t, (n, ms) in kern.items():
        ent = {"launches": n, "avg_ms": round(ms, 5)}
        if t in flops:
            ent["tflops"] = round(flops[t] / (ms * 1e-3) / 1e12, 2)
        elif t == "gather_fwd":
            by = B * (3536.0 * mean_T + 512.0) * 2                 # read + write per session
            ent["GBps"] = round(by / (ms * 1e-3) / 1e9, 1)
        elif t == "adam_item":
            ent["GBps"] = round(28.0 * N * H / (ms * 1e-3) / 1e9, 1)
        elif t == "softmax_ce":
            ent["GBps"] = round(12.0 * B * N / (ms * 1e-3) / 1e9, 1)  # read, read, write of the fp32 row
        kernels[t] = ent
    if "score_fwd" in kern:
        ach = flops["score_fwd"] / (kern["score_fwd"][1] * 1e-3) / 1e12
        if args.scoring == "f32":
            kname, peak, mult = "gemm_f32_kernel<0,0,128,128>", PEAK_F32_MATRIX_TFLOPS, 1
        else:
            # bf16 matrix cores; bf16x3 issues 3 MFMAs per algorithmic product (hi*hi + hi*lo + lo*hi); the timed span
            # also contains the [B,832] operand split kernel (~3 us)
            x3 = args.scoring.startswith("bf16x3")
            kname, peak, mult = "gemm_bf16_kernel<0,0,%s>" % ("3" if x3 else "1"), PEAK_BF16_DENSE_TFLOPS, (3 if x3 else 1)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_logits_gemm.json")
        if args.scoring.startswith("bf16x3") and os.path.exists(pmc) and (N, B) == (46033, 512):
            # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same kernel and
            # shape (tools/gemm_bench.py fwd), gfx950 FETCH_SIZE x2 correction applied; see the file for the raw counters
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        roof = {"kernel": kname + " (full-catalog logits, model_combine.py:138)", "bound": "mfma",
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                "flops_per_launch": flops["score_fwd"], "avg_ms": round(kern["score_fwd"][1], 5),
                "mfma_executed_tflops": round(ach * mult, 2), "frac_executed": round(ach * mult / peak, 4)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.tcar_oracle import TcarOracle
        # thread sweep on the GPU box host (256 hw threads): 8/16/32/64/128 threads -> 945/1066/991/592/208
        # sessions/s; more threads oversubscribe the many small ops, so the baseline runs at its best setting
        cores = min(os.cpu_count() or 1, args.cpu_threads)
        torch.set_num_threads(cores)
        ora = TcarOracle(params, fold.content, fold.mwdhm, dtype=torch.float32)
        ora.train_step(batches[0])                                  # warm-up
        c0 = time.perf_counter()
        for i in range(args.cpu_steps):
            ora.train_step(batches[(i + 1) % len(batches)])
        cdt = time.perf_counter() - c0
        cpu = {"value": round(B * args.cpu_steps / cdt, 1), "unit": "sessions/s", "cores": cores, "kind": "port",
               "sample": "%d training steps of B=%d (same synthetic Globo-like batches), PyTorch-CPU fp32 oracle, "
                         "%d threads" % (args.cpu_steps, B, cores)}

    if rank == 0:
        out = {"metric": "sessions/sec TCAR train on Globo (synthetic Globo-like fold)", "value": round(value, 1),
               "unit": "sessions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.scoring, "data": "synthetic",
               "config": {"workload": "TCAR Globo-like fold 0: N=%d items, %d-d content, B=%d/GPU, K=%d negatives, "
                                      "mean input length %.2f, full-catalog scoring, clip %d + Adam" %
                                      (N, H, B, K, mean_T, 150),
                          "global_batch": B * world, "parallelism": "dp%d" % world},
               "roofline": roof, "cpu_baseline": cpu, "kernels": kernels, "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 4), "last_loss": round(last_loss, 4)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
