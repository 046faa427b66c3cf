#!/usr/bin/env python3
"""bench.py — sessions/sec of TCAR training (BASELINE.json metric), one JSON line per run.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config globo|adressa|mind|stress10m]

N > 1: when the process is not already a rank of a torch.distributed.run job (WORLD_SIZE unset) it STARTS one as a child
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`)
before anything touches the GPU, relays the ranks' JSON line and exits with the child's return code; under
torch.distributed.run (the driver's own launch form) it simply is one of the ranks.

One "step" = one pass of the training loop's body (model_combine.py:204-231: the sampler forms the next batch — session rows,
time features, K negatives — then forward, loss, backward, per-variable clip, Adam) over one mini-batch of B = 512 sessions per
GPU.  Everything the step reads is resident in HBM when the timed region starts: model state, the session store, the negative
sources and the example indices of every batch of the schedule; the device sampler (tcar_form_batch) forms batch i + 1 on a side
stream while step i runs.  `--resident_feed` times the loop over pre-formed feeds instead (no sampler inside: reported as
"device_step_sessions_per_s" in the default run).  Rank 0 prints ONE JSON line: whole-job sessions/sec plus
  "roofline"      the kernel with the largest total time among the three full-catalog GEMMs (logits = attout E^T,
                  dX = dlogits E, dE = dlogits^T attout; the logits GEMM unless another exceeds it by > 10 %): algorithmic FLOPs per launch / mean HIP-event duration of that
                  launch, measured live in a second pass of the same K steps, on the stream the kernel is launched on (the C++
                  step driver records the events; the headline pass runs without them), against the dense bf16 MFMA peak (fp32 matrix peak for --scoring f32) of
                  /opt/skills/guides/MI355X_MICROARCH.md; "others" lists the other two the same way; "traffic" = HBM bytes
                  per launch from the committed rocprofv3 --pmc passes under profiles/ (null when the shape differs);
  "gather_roofline"  the embedding-gather kernel at 655,360 rows against the 8 TB/s HBM peak (the north star's >= 70 % target);
  "cpu_baseline"  the CPU oracle (PyTorch-CPU fp32 restatement of the reference graph) on the host cores, on a bounded
                  sample of the same workload (rank 0, N = 1 only);
  "end_to_end_sessions_per_s"  one epoch through the trainer loop of main.py (host sampler + H2D + device step), N = 1.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: Peak FP32 (matrix), v_mfma_f32_32x32x2_f32
PEAK_BF16_DENSE_TFLOPS = 2500.0     # MI355X_MICROARCH.md: Peak BF16 MFMA, dense
PEAK_HBM_GBS = 8000.0               # HBM3E spec peak
DTYPE_NOTE = {"f32": "f32", "bf16": "bf16",
              "bf16x3": "bf16x3 (split-bf16 planes, 3 MFMAs per product: fp32-class; fp32 accumulate, fp32 master state)",
              "bf16x3-mixed": "bf16 (logits GEMM on split-bf16 planes = 3 MFMAs per product, gradient GEMMs plain bf16; fp32 "
                              "accumulate, fp32 master state)"}

# BASELINE.json configs[1..4] as synthetic folds of the same shape (no dataset files exist in this environment)
CONFIGS = {
    # configs[1]: Globo, ~46k items, 250-d content, uniform negatives (sampler.py:98-99), click-delta dwell (sampler.py:91-94)
    "globo": dict(n_items=46033, hidden=250, neg_mode="uniform", gap_mode="click_delta", fold={}),
    # configs[2]: Adressa, active_t dwell seconds (adre_preprocess.py:14-20), negative-impression sampling on (sampler.py:96,118-131)
    "adressa": dict(n_items=15000, hidden=250, neg_mode="impression", gap_mode="active_t", fold=dict(active_t=True)),
    # configs[3]: MIND, one click time per session, active_t = 1 (mind_preprocess.py:22), generate_neighbor.py negatives
    # (sampler.py:97,133-140)
    "mind": dict(n_items=30000, hidden=250, neg_mode="neighbor", gap_mode="active_t",
                 fold=dict(active_t=True, same_click_time=True)),
    # configs[4]: synthetic 10M-item catalog, d = 256 (needs ~150 GB of HBM per GPU)
    "stress10m": dict(n_items=10_000_000, hidden=256, neg_mode="uniform", gap_mode="click_delta",
                      fold=dict(zipf_s=1.05, lean=True)),
}


def launch_ranks(n: int) -> int:
    """Parent of a multi-GPU run: never imports torch / touches the GPU; the ranks are child processes."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    last = None
    for line in p.stdout:
        line = line.rstrip("\n")
        if line.startswith("{") and '"metric"' in line:
            last = line
        elif line:
            print(line, file=sys.stderr)
    rc = p.wait()
    if last is not None:
        print(last)
        sys.stdout.flush()
    return rc if rc else (0 if last is not None else 1)


def build_batches(fold, n_batches, B, K, rng, cfg, with_ids=False):
    """Full batches of exactly B sessions in the fold's own length mix (sampler.py:40-49 bucketing); negatives in the
    configuration's mode, drawn with the vectorised host rules of host/sampler.py (outside the timed region).
    with_ids: also return the example indices of every batch (the device sampler forms the same batches from them)."""
    from tcar_amd.host.sampler import Sampler
    st = fold.train
    by_len = {}
    for T in np.unique(st.in_len):
        idx = np.where(st.in_len == T)[0]
        rng.shuffle(idx)
        for i in range(0, len(idx) - B + 1, B):
            by_len.setdefault(int(T), []).append(idx[i:i + B])
    flat = [(T, ids) for T, lst in by_len.items() for ids in lst]
    order = rng.permutation(len(flat))
    neg_sampler = None
    if cfg["neg_mode"] != "uniform":
        src = fold.neighbor_dict() if cfg["neg_mode"] == "neighbor" else fold.impression_dict(st)
        neg_sampler = Sampler({}, None, None, src, fold.item_dict, K, batch_size=B, gap_mode=cfg["gap_mode"],
                              neg_mode=cfg["neg_mode"], store=st, verbose=False, neg_fast=True)
    out, out_ids = [], []
    state = np.random.get_state()
    np.random.seed(int(rng.randint(1 << 30)))
    for j in order[:n_batches]:
        T, ids = flat[j]
        b = st.batch_arrays(ids, cfg["gap_mode"])
        if neg_sampler is None:
            b["neg"] = rng.randint(0, fold.n_items, size=(B, K)).astype(np.int32)
        else:
            b["neg"] = neg_sampler._negatives(ids.tolist(), b["label"], ids)
        out.append(b)
        out_ids.append(np.asarray(ids, dtype=np.int32))
    np.random.set_state(state)
    return (out, out_ids) if with_ids else out


def pmc_traffic(tag, nsplit, N, B):
    """HBM bytes per launch of one scoring GEMM from the committed rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE,
    MI355X_MICROARCH.md §HBM; tools/pmc_gemm.sh); only valid for the shape and plane count it was collected on."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):      # the newest committed pass for this kernel form
        p = os.path.join(ROOT, "profiles", "%s_pmc_%s_n%d.json" % (rnd, tag, nsplit))
        if os.path.exists(p):
            d = json.load(open(p))
            if tuple(d.get("shape_N_B", ())) == (N, B):
                return d.get("hbm_bytes_per_launch"), os.path.relpath(p, ROOT)
    return None, None


def gather_roofline(dev):
    """The embedding-gather kernel against the HBM roofline (north star: >= 70 % on this kernel), measured here because the
    training step itself gathers ~1,100 rows per launch (latency bound): 655,360 rows = 16,384 sessions x 40 clicks from a
    2 M-item table (model_combine.py:54-107 forward: item + content + position + five time rows + dwell row per click, clipped
    and concatenated; algorithmic bytes per session of SURVEY.md §8(d): read 3536 T + 512, written the same)."""
    import ctypes as C
    import torch
    from tcar_amd import _lib
    from tcar_amd._lib import Batch, Dims, Tables
    lib = _lib.load()
    N, B, T, H, Ht, ldh, ldt = 2_000_000, 16384, 40, 250, 64, 256, 64
    ic, pt, ct, ek = 512, 320, 128, 832
    g = torch.Generator(device="cpu").manual_seed(0)
    ri = lambda lo, hi, *s: torch.randint(lo, hi, s, generator=g, dtype=torch.int32).to(dev)
    E = torch.randn(N, ek, device=dev) * 0.05
    small = [torch.randn(v, ldt, device=dev) * 0.3 for v in (13, 32, 8, 25, 61, 11)]
    pos = torch.randn(40, ldh, device=dev) * 0.02
    seq, pub = ri(1, N + 1, B, T), [ri(1, v, B, T) for v in (13, 32, 8, 25, 61)]
    gap, cw, ch = ri(0, 11, B, T), ri(0, 7, B), ri(0, 24, B)
    d = Dims(N, H, Ht, ldh, ldt)
    tab = Tables()
    tab.E, tab.pos, tab.dur = E.data_ptr(), pos.data_ptr(), small[5].data_ptr()
    bt = Batch()
    bt.B, bt.T, bt.K = B, T, 0
    bt.seq, bt.cw, bt.ch, bt.gap = seq.data_ptr(), cw.data_ptr(), ch.data_ptr(), gap.data_ptr()
    for k in range(5):
        tab.time[k], bt.pub[k] = small[k].data_ptr(), pub[k].data_ptr()
    outs = [torch.empty(B * T, ic, device=dev), torch.empty(B * T, pt, device=dev), torch.empty(B * T, ldt, device=dev),
            torch.empty(B, ct, device=dev)]
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    run = lambda: lib.tcar_gather_clip_fwd(C.byref(d), C.byref(tab), C.byref(bt), *[p(t) for t in outs], st)
    for _ in range(3):
        assert run() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 10
    e0.record(torch.cuda.current_stream(dev))
    for _ in range(iters):
        run()
    e1.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = 2.0 * B * (3536.0 * T + 512)
    gbs = nbytes / ms / 1e6
    # HEADLINE = counter-verified REAL bytes: FETCH_SIZE x 2 + WRITE_SIZE of the committed rocprofv3 --pmc passes over exactly this
    # launch (the six small tables — 1,536 of the 3,536 algorithmic read bytes per click — are LDS-resident and never reach HBM, so
    # the algorithmic figure of SURVEY.md 8(d) overstates what the memory system moves); the algorithmic figure sits beside it
    real, src = None, None
    for rnd in ("r06", "r05", "r04", "r03"):
        pm = os.path.join(ROOT, "profiles", rnd + "_pmc_gather_fwd.json")
        if os.path.exists(pm):
            d_ = json.load(open(pm))
            real, src = d_["hbm_read_bytes"] + d_["hbm_write_bytes"], os.path.relpath(pm, ROOT)
            break
    out = {"kernel": "gather_clip_fwd (throughput form), model_combine.py:54-107", "bound": "hbm", "rows": B * T, "avg_ms": round(ms, 4),
           "peak": PEAK_HBM_GBS, "unit": "GB/s",
           "algorithmic": {"bytes_per_launch": nbytes, "achieved": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4),
                           "note": "read + written bytes of SURVEY.md 8(d): 2 x (3536 T + 512) per session"},
           "note": "HIP-event time of 10 launches; separate from the step, whose own gather is latency bound (in_step below)"}
    if real is not None:
        rg = real / (ms * 1e-3) / 1e9
        out.update({"achieved": round(rg, 1), "frac": round(rg / PEAK_HBM_GBS, 4), "bytes_per_launch": real, "traffic": real,
                    "traffic_source": src, "basis": "counter-verified HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE) / this run's time"})
    else:
        out.update({"achieved": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4), "bytes_per_launch": nbytes, "traffic": None,
                    "basis": "algorithmic bytes (no committed counter pass for this launch)"})
    return out


def bucket_batch(fold, T, B, K, rng):
    """One full batch of the length bucket T (sampler.py:40-49 buckets by exact input length): items from the fold's Zipf popularity
    law, their publish fields from the catalog, click fields / gap buckets / labels / K uniform negatives at random."""
    n = fold.n_items
    ranks = np.searchsorted(fold._cdf, rng.random_sample(B * T)).clip(0, n - 1)
    items0 = fold._perm[ranks].reshape(B, T)
    pub = fold.mwdhm[items0]                                   # [B, T, 5]: month, day, isoweekday, hour + 1, minute + 1
    b = {"seq": items0 + 1, "label": fold._perm[np.searchsorted(fold._cdf, rng.random_sample(B)).clip(0, n - 1)],
         "pm": pub[..., 0], "pd": pub[..., 1], "pw": pub[..., 2], "ph": pub[..., 3], "pmi": pub[..., 4],
         "cw": rng.randint(0, 7, B), "ch": rng.randint(0, 24, B), "gap": rng.randint(0, 11, (B, T)), "neg": rng.randint(0, n, (B, K))}
    return {k: np.ascontiguousarray(v).astype(np.int32) for k, v in b.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="globo", choices=sorted(CONFIGS),
                    help="BASELINE.json configuration (synthetic fold of that shape)")
    ap.add_argument("--batch_size", type=int, default=512)
    ap.add_argument("--n_items", type=int, default=0, help="override the configuration's catalog size")
    ap.add_argument("--hidden_size", type=int, default=0, help="override the configuration's hidden size")
    ap.add_argument("--time_hidden_size", type=int, default=64)
    ap.add_argument("--neg_num", type=int, default=20)
    ap.add_argument("--n_batches", type=int, default=48)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_e2e", action="store_true", help="skip the end-to-end epoch through the trainer loop")
    ap.add_argument("--cpu_steps", type=int, default=20)
    ap.add_argument("--cpu_threads", type=int, default=0,
                    help="host threads of the CPU baseline; 0 (default) = the best of a short sweep over 8 / 16 / 32 / 64")
    ap.add_argument("--no_kernel_timing", action="store_true")
    ap.add_argument("--stall_ms", type=float, default=0.0,
                    help="diagnostic (profiling): hold the main stream for this long at the start of the timed loop, so that the host "
                         "enqueues the timed steps AHEAD of the device — under rocprofv3's kernel trace the host is otherwise the "
                         "bottleneck and the timeline shows its launch latency, not the device's schedule.  The reported time "
                         "then includes the stall: use only for timelines")
    ap.add_argument("--dp_mode", default="auto", choices=["auto", "replica", "sharded"],
                    help="multi-GPU exchange: replica = all-reduce of the dense item gradient; sharded = catalog-sharded scoring")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU dry runs)")
    ap.add_argument("--same_device", action="store_true", help="dry run: put every rank on cuda:0")
    ap.add_argument("--no_gather_roofline", action="store_true", help="skip the embedding-gather HBM roofline measurement")
    ap.add_argument("--resident_feed", action="store_true",
                    help="headline loop over pre-formed feeds (no sampler inside the timed steps); default: the device sampler "
                         "forms every batch inside the loop (BASELINE.md §3: sampler + fwd + bwd + update)")
    ap.add_argument("--scoring", default="bf16x3-mixed", choices=["f32", "bf16x3", "bf16x3-mixed", "bf16"],
                    help="precision of the full-catalog scoring GEMMs.  bf16x3-mixed (default; BASELINE.json configs[1] is "
                         "quoted in bf16): logits from split-bf16 planes (three MFMAs per product, fp32-class: the 1e-3 "
                         "logits / HR@20 / MRR@20 gate of the north star), the two gradient GEMMs in plain bf16, fp32 "
                         "accumulation and fp32 master state throughout; bf16x3: all three GEMMs fp32-class")
    ap.add_argument("--no_by_T", action="store_true", help="skip ms_per_step_by_T (step time per input-length bucket T = 1, 2, 5, 10, 40; "
                                                             "it is also skipped by --no_kernel_timing, by TCAR_FORCE_DP and for N > 1)")
    ap.add_argument("--by_T", type=str, default="1,2,5,10,40", help="the bucket lengths of ms_per_step_by_T")
    ap.add_argument("--launch_check", action="store_true",
                    help="only the rank launch + the three collectives of the exchanges on tiny tensors (dp.preflight): with "
                         "--backend gloo on CPU (tests/), with nccl on the GPUs; prints n_gpus and the backend / RCCL versions")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))           # before any import of torch: this process never initialises a GPU
    if args.launch_check:
        # the rank launch and the three collectives of the exchanges, nothing else: `--launch_check --backend gloo` runs on CPU
        # tensors (tests/test_bench_launch.py), `--launch_check` (nccl = RCCL) puts rank r on cuda:r — a failing rendezvous or
        # collective names itself here (backend, world, device, library versions) and exits non-zero
        import torch
        import torch.distributed as dist
        import tcar_amd  # noqa: F401
        from tcar_amd.dp import preflight
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1:
            on_gpu = args.backend == "nccl"
            dev = None
            if on_gpu:
                lr = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
                torch.cuda.set_device(lr)
                dev = "cuda:%d" % lr
                dist.init_process_group("nccl", device_id=torch.device(dev))
            else:
                dist.init_process_group(args.backend)
            try:
                info = preflight(dist.group.WORLD, dev)
            except Exception as e:
                print("[tcar] launch_check FAILED on rank %s: %s" % (os.environ.get("RANK"), e), file=sys.stderr)
                sys.exit(3)
            if dist.get_rank() == 0:
                print(json.dumps({"metric": "launch_check", "n_gpus": world, "collectives": info}))
            dist.destroy_process_group()
        else:
            print(json.dumps({"metric": "launch_check", "n_gpus": 1}))
        return

    import torch
    import tcar_amd  # noqa: F401
    from tcar_amd.host.synth import SynthFold

    cfg = dict(CONFIGS[args.config])
    N = args.n_items or cfg["n_items"]
    H = args.hidden_size or cfg["hidden"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    # TCAR_FORCE_COLLECTIVES=1 on ONE rank: a process group of one over the chosen backend, and the data-parallel engines issue every
    # collective of their exchange (identity results) — RCCL under the N > 1 code path on the one GPU a box has
    force_coll = world == 1 and bool(int(os.environ.get("TCAR_FORCE_COLLECTIVES", "0") or 0))
    if force_coll:
        os.environ.setdefault("TCAR_FORCE_DP", "1")
        os.environ.setdefault("MASTER_PORT", "29531")
    if world > 1 or force_coll:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        # first contact: the three collectives of the exchanges on tiny tensors, before anything large is built; a job that
        # cannot communicate says so here (backend, world, device, RCCL version) and exits non-zero
        from tcar_amd.dp import preflight
        try:
            caps = preflight(dist.group.WORLD, dev)
        except Exception as e:
            print("[tcar] rank %d: %s" % (rank, e), file=sys.stderr)
            sys.exit(3)

    B, K = args.batch_size, args.neg_num
    lean = bool(cfg["fold"].get("lean"))
    # every rank builds the same catalog (seed 2020) and its own shard of sessions (weak scaling: B per GPU)
    fold = SynthFold(n_items=N, dim=H, n_train=max(60000, 4 * B * args.n_batches), n_test=1000, seed=2020, **cfg["fold"])
    rng = np.random.RandomState(2020 + rank)
    batches, batch_ids = build_batches(fold, args.n_batches, B, K, rng, cfg, with_ids=True)
    if world > 1:
        # ranks step in lock-step over buckets of the same length T: share rank 0's T schedule
        sched = torch.tensor([b["seq"].shape[1] for b in batches], device=dev)
        dist.broadcast(sched, 0)
        want = sched.cpu().tolist()
        pool = {}
        for b, ids in zip(*build_batches(fold, 10 ** 9, B, K, rng, cfg, with_ids=True)):
            pool.setdefault(b["seq"].shape[1], []).append((b, ids))
        picked = [pool[T][i % len(pool[T])] for i, T in enumerate(want)]
        batches, batch_ids = [x[0] for x in picked], [x[1] for x in picked]

    from tcar_amd.host.model import initial_variables     # the product's own initialiser (modules.py:32-34,50-51)
    np.random.seed(2020)
    params = initial_variables(N, H, args.time_hidden_size, 0.002, 0.05, weight_seed=2020, lean=lean)
    exchange = None
    if world > 1 or os.environ.get("TCAR_FORCE_DP"):       # TCAR_FORCE_DP=1: time the data-parallel code path on one rank
        from tcar_amd.dp import make_dp_engine
        eng = make_dp_engine(params, fold.content, fold.mwdhm, device=dev, group=(dist.group.WORLD if dist is not None else None),
                             scoring=args.scoring, mode=args.dp_mode)
    else:
        from tcar_amd.engine import TcarEngine
        eng = TcarEngine(params, fold.content, fold.mwdhm, device=dev, scoring=args.scoring)
    resident = [eng.make_resident(b) for b in batches]
    # the form the step driver itself takes for these batches (tcar_step_form), reported as `step_form`
    step_form0 = eng.step_form(resident[0]) if (world == 1 and hasattr(eng, "step_form") and not os.environ.get("TCAR_FORCE_DP")) else {}
    mean_T = float(np.mean([b["seq"].shape[1] for b in batches]))
    if not os.environ.get("TCAR_NO_RESERVE"):
        # setup, not steps: size the activation workspace for the longest bucket once (a trainer does the same from its bucket
        # table), so that no step of a short run re-allocates when a longer bucket first appears
        eng._ensure_work(B, max(b["seq"].shape[1] for b in batches))

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Single rank: each step's Adam is applied at the start of the NEXT train_step (engine.train_step(defer_update=True): the
    # rows the next batch gathers first on the main stream, every other item row on the aux stream beside the next forward
    # pass — bitwise the same arithmetic, tests/test_gpu_parity.py).  The last update is flushed INSIDE the timed region, so K
    # timed steps contain exactly K forward passes, K backward passes and K updates.  TCAR_NO_DEFER=1: update inside its step.
    # (the catalog-sharded engine on ONE rank takes the same split update — ShardedEngine.can_defer; with more ranks its update is
    #  followed by the exchange of the owned rows and stays inside the step)
    defer = {"defer_update": True} if (world == 1 and not os.environ.get("TCAR_NO_DEFER") and
                                       (not os.environ.get("TCAR_FORCE_DP") or getattr(eng, "can_defer", False))) else {}
    # The sampler of the step: everything it reads is RESIDENT in HBM before the timed region starts (session store, negative
    # sources, and the example indices of every batch of the schedule: DeviceSampler.plan); inside the loop tcar_form_batch
    # forms batch i + 1 (session rows, time features, K negatives) on a side stream while step i runs.
    from tcar_amd.device_sampler import DeviceSampler
    src = None if cfg["neg_mode"] == "uniform" else (fold.neighbor_dict() if cfg["neg_mode"] == "neighbor" else fold.impression_dict(fold.train))
    ds = DeviceSampler(eng, fold.train, cfg["neg_mode"], src, fold.item_dict, seed=2020 + rank)
    sched = lambda first, n: [batch_ids[(first + i) % len(batch_ids)] for i in range(n)]

    def run(n, first, sampler_in_loop):
        if sampler_in_loop:
            ds.plan(sched(first, n))
            for bt in ds.planned(K, cfg["gap_mode"]):
                eng.train_step(None, bt=bt, **defer)
        else:
            for i in range(n):
                eng.train_step(None, bt=resident[(first + i) % len(resident)], **defer)
        eng.flush()                              # K steps contain exactly K updates

    def timed(sampler_in_loop):
        import gc
        gc.collect()
        gc.disable()                             # no collector pause inside a loop of a few milliseconds
        try:
            return timed_body(sampler_in_loop)
        finally:
            gc.enable()

    def timed_body(sampler_in_loop):
        run(args.warmup, 0, sampler_in_loop)     # the last warm-up step's update belongs to the warm-up
        if sampler_in_loop:
            ds.plan(sched(args.warmup, args.steps))        # the timed schedule's indices: resident before the clock starts
        sync()
        t0 = time.perf_counter()
        if args.stall_ms > 0:
            torch.cuda._sleep(int(args.stall_ms * 1e-3 * 2.0e9))        # (~2 GHz shader clock: the length is approximate)
        if sampler_in_loop:
            for bt in ds.planned(K, cfg["gap_mode"]):
                eng.train_step(None, bt=bt, **defer)
            eng.flush()
        else:
            run(args.steps, args.warmup, False)
        t_enq = time.perf_counter() - t0         # host time to enqueue every step (the loop never synchronises)
        sync()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, t_enq

    headline_sampler = not args.resident_feed
    other = None
    if not (args.no_cpu_baseline and args.no_e2e):          # (the A/B tools time the headline loop only)
        # (the FIRST timed loop of a process ran up to 1.5x slow on some boxes — clocks / first-touch of the workspaces —
        #  so the secondary figure gets a longer warm-up of its own; the headline loop always runs second)
        run(5 * args.warmup, 0, not headline_sampler)
        dt_o, _ = timed(not headline_sampler)
        other = B * world * args.steps / dt_o
    dt, t_enq = timed(headline_sampler)
    eng.check_forks()                          # (raises if a flag fork's poll ever timed out: the measurement would be void)
    last_loss = float(eng.loss[:B].mean())
    dt_ev = None
    if not args.no_kernel_timing:
        # kernel timing: a SECOND pass of the same K steps with HIP events around the three scoring GEMMs and the projection
        # launch, recorded by the C++ step driver on the stream each kernel is launched on.  The events are kept out of the
        # headline pass: every event pair in the middle of a stream's chain costs ~5-10 us of bubble (8 pairs per step = 2-3 %).
        eng.enable_native_timing(args.steps)
        dt_ev, _ = timed(headline_sampler)
    if hasattr(eng, "exchange_info"):
        exchange = eng.exchange_info()        # after the timed steps: carries the bytes each collective moved per step
        if dist is not None:
            # what the BACKEND saw: world size and rank as the process group reports them, backend, RCCL /
            # HIP versions, reduce-scatter capability — from the three preflight collectives of this very job
            exchange["process_group"] = {k: caps.get(k) for k in ("world", "backend", "rccl", "hip", "torch", "device", "reduce_scatter")}
            exchange["process_group"]["ranks_seen_by_all_gather"] = dist.get_world_size()
    value = B * world * args.steps / dt
    eng_splitk = getattr(eng, "splitk", 18)

    # ---- roofline of the three full-catalog GEMMs, timed live inside the timed steps -------------------------------
    Ht = args.time_hidden_size
    k_alg = 2 * H + 5 * Ht                                          # 820 contraction length (model_combine.py:132-138)
    n_local = getattr(eng, "n_local_items", N)                      # catalog-sharded scoring: this rank's share of N
    b_glob = getattr(eng, "score_batch", B)                         # ... scored against the all-gathered session batch
    flops = {"score_fwd": 2.0 * b_glob * n_local * k_alg, "score_dx": 2.0 * b_glob * n_local * k_alg,
             "score_dE": 2.0 * b_glob * n_local * (H + 5 * Ht)}     # content columns are frozen: no dE for them
    ref = {"score_fwd": "full-catalog logits = attout E^T, model_combine.py:138",
           "score_dx": "dX = dlogits E (gradient of :138 w.r.t. attout)",
           "score_dE": "dE = dlogits^T attout (gradient of :138 w.r.t. the item table and candidate time vectors)"}
    roof, kernels = None, {}
    by_T = None
    if not args.no_kernel_timing and hasattr(eng, "native_timing_ms"):
        import ctypes as C
        g = eng.geo
        x3 = args.scoring.startswith("bf16x3")
        nsb = 1 if args.scoring in ("bf16", "bf16x3-mixed") else 3
        peak, mult_f = (PEAK_F32_MATRIX_TFLOPS, 1) if args.scoring == "f32" else (PEAK_BF16_DENSE_TFLOPS, 3 if x3 else 1)
        shapes = {"score_fwd": (1, b_glob, n_local, g.ek, mult_f, 1),
                  "score_dx": (0, b_glob, g.ek, g.Npad if n_local == N else ((n_local + 127) // 128) * 128, nsb, eng.splitk),
                  "score_dE": (2, n_local, g.ldh + g.pt, (b_glob + 31) & ~31, nsb, 1)}
        # catalog-sharded step: two composite spans (split + logits + statistics | combine + dlogits + dE + dX + slab reduce)
        flops["shard_score"] = flops["score_fwd"]
        flops["shard_backward"] = flops["score_dx"] + flops["score_dE"]
        ref["shard_score"] = "attout planes + logits of the shard + softmax statistics (sharded.py)"
        ref["shard_backward"] = "lse combine + dlogits planes + dE of the shard + dX partial + slab reduce (sharded.py)"
        # the form the step driver itself chose for these batches (tcar_step_form: fused CE / one-hot segment / one-hot backward) —
        # asked of the engine, not re-derived from the environment
        form_ = eng.step_form(resident[0]) if (world == 1 and hasattr(eng, "step_form") and not os.environ.get("TCAR_FORCE_DP")) else \
            {"fused_ce": False, "onehot_fwd": False, "onehot_bwd": False, "sorted_rows": False}
        ents = []
        for kind, tag in enumerate(eng.TIMED_KERNELS):
            ms = [m for m in eng.native_timing_ms(kind) if m > 0]
            if not ms:
                continue
            avg = float(np.mean(ms))
            if tag == "gather_fwd":
                # the step's OWN embedding gather (model_combine.py:54-107): ~B * mean_T rows per launch, latency bound — reported
                # beside the throughput form of gather_roofline
                kernels[tag] = {"launches": len(ms), "avg_ms": round(avg, 5), "rows_per_launch": round(B * mean_T, 1),
                                "algorithmic_GBps": round(2.0 * B * (3536.0 * mean_T + 512) / (avg * 1e-3) / 1e9, 1)}
                continue
            if tag == "session_proj":
                # the grouped projection launch of both attention layers (modules.py:94-96,126-131: X W_in + C W_c + I W_int,
                # X_t W'_in + C W'_c, and the first click-query layer) — the largest of the session-side small GEMMs
                # (gemm_x3_kernel, fp32 operands split to bf16 hi/lo while staging: 3 MFMAs per product).  Algorithmic flops
                # per SURVEY.md 8(d) at the schedule's mean input length; a latency-bound launch, priced against the MFMA peak.
                fl = B * (2.0 * mean_T * ((2 * H + H + Ht) * H + (5 * Ht + H) * H) + 2.0 * (2 * Ht) * H)
                ach = fl / (avg * 1e-3) / 1e12
                ents.append({"kernel": "gemm_x3_kernel, grouped projections of both attention layers (modules.py:94-96,126-131)",
                             "tag": tag, "bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                             "frac": round(ach / peak, 5), "frac_mfma_executed": round(3 * ach / peak, 5), "traffic": None,
                             "flops_per_launch": fl, "launches": len(ms), "avg_ms": round(avg, 5), "total_ms": round(avg * len(ms), 3),
                             "note": "latency bound: one 128-deep K chunk per workgroup (416 workgroups at T = 1), 7 such launches per step",
                             "timing": "HIP events on the launch stream, second pass of the same steps"})
                kernels[tag] = {"launches": len(ms), "avg_ms": round(avg, 5), "tflops": round(ach, 3)}
                continue
            if tag.startswith("shard_"):
                ach = flops[tag] / (avg * 1e-3) / 1e12
                ents.append({"kernel": "tcar_%s: %s" % (tag, ref[tag]), "tag": tag, "bound": "mfma", "achieved": round(ach, 2),
                             "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                             "flops_per_launch": flops[tag], "launches": len(ms), "avg_ms": round(avg, 5),
                             "total_ms": round(avg * len(ms), 3),
                             "timing": "HIP events around the C-ABI call (several kernels), second pass of the same steps"})
                kernels[tag] = {"launches": len(ms), "avg_ms": round(avg, 5), "tflops": round(ach, 2)}
                continue
            ach = flops[tag] / (avg * 1e-3) / 1e12
            lay, M_, N_, K_, mult, sk = shapes[tag]
            name = "gemm_f32_kernel (fp32 MFMA)"
            if args.scoring != "f32":
                buf = C.create_string_buffer(160)
                eng.lib.tcar_gemm_bf16_variant(lay, M_, N_, K_, 3 if mult == 3 else 1, sk, buf, 160)
                name = buf.value.decode()
            ce_form = tag == "score_fwd" and n_local == N and form_["fused_ce"]
            traffic, src = pmc_traffic(tag + ("_ce" if ce_form else ""), 3 if mult == 3 else 1, N, B) if (world == 1 and args.scoring != "f32") else (None, None)
            # algorithmic HBM bytes of the launch: every operand once (bf16 planes: 2 B per plane and element; fp32: 4 B), the
            # result once
            opb = 4 if args.scoring == "f32" else 2 * (2 if mult == 3 else 1)
            n_rows = g.Npad if n_local == N else ((n_local + 127) // 128) * 128
            # (training steps of the mixed precision: the logits GEMM's softmax epilogue writes a bf16 plane + group statistics,
            # not fp32 logits)
            ce_epi = ce_form or (tag != "score_fwd" and n_local == N and form_["fused_ce"])
            out_fwd = (2 * b_glob * n_rows + 8 * b_glob * (n_rows // 96)) if ce_epi else 4 * b_glob * n_rows
            # one-hot form of the candidate time columns (training steps of the mixed precision, single rank): the GEMM reads the
            # item | content columns of both operands, ONE 160-column one-hot plane and the two 160-column time-score planes, and
            # runs 3 MFMAs per product over 2 ldh columns + 2 over 160 (executed / algorithmic flops = (3 * 512 + 2 * 160) / 820)
            onehot = ce_form and form_["onehot_fwd"]
            in_fwd = (opb * (n_rows * g.ic + b_glob * g.ic) + 2 * n_rows * 160 + 4 * b_glob * 160) if onehot else opb * (n_rows * g.ek + b_glob * g.ek)
            form = None
            if onehot and tag == "score_fwd":
                mult = (3.0 * g.ic + 2.0 * 160) / k_alg
                form = "one-hot time segment: K = %d columns at 3 MFMAs per product + 160 at 2 (one B plane)" % g.ic
            # one-hot form of the two gradient GEMMs (default of the mixed precision on one rank): dX reads the item | content planes
            # of E + the 160-column one-hot plane and writes slabs of 2 ldh + 160 columns; dE writes its item block + 5 (q, z) pairs
            # per candidate instead of the [N, 5 ldt] time block
            oh_bwd = tag != "score_fwd" and n_local == N and form_["onehot_bwd"]
            if oh_bwd:
                form = {"score_dx": "one-hot form: dlogits [E_item | E_content | OH], %d + 160 columns" % g.ic,
                        "score_dE": "one-hot form: item block + per-candidate (||gy||^2, x.gy) pairs, no [N, 5 ldt] block"}[tag]
                name = {"score_dx": name, "score_dE": "gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0> tile 192x192x32, (q, z) epilogue"}[tag]
                traffic, src = pmc_traffic({"score_dx": "score_dx_onehot", "score_dE": "score_dE_qz"}[tag], 1, N, B) if world == 1 else (None, None)
            alg_bytes = {"score_fwd": in_fwd + out_fwd,
                         "score_dx": (opb * (b_glob * n_rows + n_rows * g.ic) + 2 * n_rows * 160 + 4 * sk * b_glob * (g.ic + 160)) if oh_bwd
                         else opb * (b_glob * n_rows + n_rows * g.ek) + 4 * sk * b_glob * g.ek,
                         "score_dE": (opb * (b_glob * n_rows + b_glob * (g.ldh + g.pt)) + 4 * n_local * g.ldh + 40 * n_local) if oh_bwd
                         else opb * (b_glob * n_rows + b_glob * (g.ldh + g.pt)) + 4 * n_local * (g.ldh + g.pt)}[tag]
            gbs = alg_bytes / (avg * 1e-3) / 1e9
            f_mfma, f_hbm = ach * mult / peak, gbs / PEAK_HBM_GBS
            ent = {"kernel": "%s (%s)" % (name, ref[tag]), "tag": tag}
            if form:
                ent["form"] = form
            if f_hbm > f_mfma:       # the hi-only gradient GEMMs: a third of the MFMA work, the same fp32 result to write
                ent.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(f_hbm, 4)})
            else:
                ent.update({"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4)})
            ent.update({"traffic": traffic, "traffic_source": src, "flops_per_launch": flops[tag], "bytes_per_launch": alg_bytes,
                        "launches": len(ms), "avg_ms": round(avg, 5), "total_ms": round(avg * len(ms), 3),
                        "tflops": round(ach, 2), "mfma_executed_tflops": round(ach * mult, 2), "frac_mfma_executed": round(f_mfma, 4),
                        "hbm_GBps_algorithmic": round(gbs, 1), "frac_hbm": round(f_hbm, 4),
                        "timing": "HIP events on the launch stream in a second pass of the same steps (other streams kernels co-run)"})
            ents.append(ent)
            kernels[tag] = {"launches": len(ms), "avg_ms": round(avg, 5), "tflops": round(ach, 2)}
        if ents:
            ents.sort(key=lambda e: -e["total_ms"])
            # (the event-timed dE runs beside dX and a short run can put it a hair above the logits GEMM, which the kernel trace shows
            #  as the longer launch: the forward GEMM stays the line's kernel unless another one exceeds it by more than 10 %)
            fwd = [e for e in ents if e["tag"] == "score_fwd"]
            if fwd and fwd[0]["total_ms"] >= 0.9 * ents[0]["total_ms"]:
                ents.remove(fwd[0])
                ents.insert(0, fwd[0])
            roof = dict(ents[0])
            roof["others"] = ents[1:]
        eng._ev = None
        eng._tm = None
        eng._ctx_key = None

    # ---- step time per length bucket: the sampler buckets by EXACT input length (sampler.py:40-49), the headline is
    # quoted at the fold's mean length while the session-side head and tail of the step scale with B * T.  Same engine, same loop body
    # (deferred update, resident feeds of ONE length each, items drawn from the fold's popularity law), 40 timed steps after 15, best of two.
    # (Runs AFTER the roofline pass has been read out: its steps would otherwise overwrite that pass's event slots.)
    if world == 1 and not os.environ.get("TCAR_FORCE_DP") and not args.no_by_T and not args.no_kernel_timing and N <= 200000:
        by_T = {}
        rng_t = np.random.RandomState(77)
        # (its 4 x 95 steps per length train on synthetic buckets: the engine's variables and Adam state are put back afterwards, so
        #  nothing that runs later sees a model perturbed by them — ADVICE r05)
        state_before_by_T = eng.export_state()
        for T_ in [int(x) for x in args.by_T.split(",") if x]:
            feeds = [eng.make_resident(bucket_batch(fold, T_, B, K, rng_t)) for _ in range(4)]
            eng._ensure_work(B, T_)
            for i in range(15):
                eng.train_step(None, bt=feeds[i % 4], **defer)
            eng.flush()
            best = None
            for rep in range(2):                     # (best of two: a run of 40 steps is short enough for one host hiccup to show)
                torch.cuda.synchronize()
                w0 = time.perf_counter()
                for i in range(40):
                    eng.train_step(None, bt=feeds[i % 4], **defer)
                eng.flush()
                torch.cuda.synchronize()
                t_ = (time.perf_counter() - w0) / 40 * 1e3
                best = t_ if best is None else min(best, t_)
            by_T[str(T_)] = round(best, 4)
            del feeds
        eng.check_forks()
        eng.load_state(state_before_by_T)
        del state_before_by_T
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and N <= 200000:
        from oracle.tcar_oracle import TcarOracle
        # thread sweep on the GPU box host (256 hw threads): 8/16/32/64/128 threads -> 945/1066/991/592/208
        # sessions/s; more threads oversubscribe the many small ops, so the baseline runs at its best setting
        hw = os.cpu_count() or 1
        cores = min(hw, args.cpu_threads if args.cpu_threads > 0 else 16)
        torch.set_num_threads(cores)
        import random
        from oracle.sampler_oracle import OracleSampler, batch_to_arrays
        ora = TcarOracle(params, fold.content, fold.mwdhm, dtype=torch.float32)
        ora.train_step(batches[0])                                  # warm-up
        # the reference's loop (model_combine.py:196-231): its Python per-click sampler (sampler.py:52-113, restated in
        # oracle/sampler_oracle.py) forms every batch inside the timed region, then one training step
        take = [batch_ids[i % len(batch_ids)] for i in range(1, args.cpu_steps + 1)]
        with_active = cfg["gap_mode"] == "active_t"
        ld, sd, td = fold.to_dicts(fold.train, with_active=with_active, examples=np.unique(np.concatenate(take)))
        neg_src = {0: [0]} if cfg["neg_mode"] == "uniform" else src
        random.seed(2020)
        np.random.seed(2020)
        # thread sweep, 3 steps per setting (the many small ops of the graph oversubscribe quickly): the baseline runs at the best
        sweep = {}
        if args.cpu_threads <= 0:
            for th in [t for t in (8, 16, 32, 64) if t <= hw]:
                torch.set_num_threads(th)
                ora.train_step(batches[0])
                w0 = time.perf_counter()
                for i in range(3):
                    ora.train_step(batches[1 + i % (len(batches) - 1)])
                sweep[str(th)] = round(3 * B / (time.perf_counter() - w0), 1)
            cores = int(max(sweep, key=lambda t: sweep[t]))
            torch.set_num_threads(cores)
        # same scope as the GPU headline: the sampler object (bucketing, shuffle) is built outside the clock, as DeviceSampler and
        # its plan are on the GPU side; batch formation + negatives + the training step are inside
        smp = OracleSampler(ld, sd, td, neg_src, fold.item_dict, K, batch_size=B, gap_mode=cfg["gap_mode"], neg_mode=cfg["neg_mode"])
        c0 = time.perf_counter()
        n_cpu = 0
        while smp.has_next() and n_cpu < B * args.cpu_steps:
            feed = batch_to_arrays(smp.next_batch())
            ora.train_step(feed)
            n_cpu += feed["seq"].shape[0]
        cdt = time.perf_counter() - c0
        cpu_model = "unknown"
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        cpu = {"value": round(n_cpu / cdt, 1), "unit": "sessions/s", "cores": cores, "host_threads": hw, "cpu_model": cpu_model,
               "thread_sweep_sessions_per_s": sweep or None,
               "kind": "port",
               "sample": "%d sessions: the reference's loop on the CPU — per-click Python sampler (batch formation, K negatives; "
                         "its construction = bucketing + shuffle is outside the clock, like the GPU side's) + one training step per "
                         "batch of <= %d — PyTorch-CPU fp32 oracle on %d of the host's %d hardware threads (the fastest setting of "
                         "thread_sweep_sessions_per_s)" % (n_cpu, B, cores, hw)}

    e2e = None
    if rank == 0 and world == 1 and not args.no_e2e and N <= 200000 and not os.environ.get("TCAR_FORCE_DP"):
        # one epoch through the trainer loop of main.py (Seq2SeqAttNN.train): host sampler on a prefetch thread, batches
        # packed into pinned memory, H2D, device step; the second epoch is the figure (the first builds caches)
        import contextlib
        import io
        from tcar_amd.host.model import Seq2SeqAttNN
        a = fold.model_args(batch_size=B, neg_num=K, epoch=2, hidden_size=H, time_hidden_size=Ht, scoring=args.scoring,
                            gap_mode=cfg["gap_mode"], neg_mode=cfg["neg_mode"], neg_fast=1, initial_variables=params)
        st = fold.train
        len_dict = {int(T): np.where(st.in_len == T)[0].tolist() for T in np.unique(st.in_len)}
        src = {0: [0]} if cfg["neg_mode"] == "uniform" else (fold.neighbor_dict() if cfg["neg_mode"] == "neighbor"
                                                              else fold.impression_dict(st))
        del eng, resident
        with contextlib.redirect_stdout(io.StringIO()):
            model = Seq2SeqAttNN(a)
            model.train(None, fold.item_dict, (len_dict, None, None, st), src, a, None, None)
            host_rate = model.train_sessions / model.train_seconds
            model.device_sampler = True          # same loop, batches formed + negatives drawn on the GPU (tcar_form_batch)
            model.train(None, fold.item_dict, (len_dict, None, None, st), src, a, None, None)
            dev_rate = model.train_sessions / model.train_seconds
        e2e = {"value": round(max(host_rate, dev_rate), 1), "unit": "sessions/s", "sessions": model.train_sessions,
               "host_sampler": round(host_rate, 1), "device_sampler": round(dev_rate, 1),
               "what": "2nd epoch of Seq2SeqAttNN.train on the same fold, whole loop (bucketed shuffle, batch formation, negative "
                       "sampling, H2D, device step): host_sampler = vectorised numpy sampler on a prefetch thread + pinned H2D "
                       "of the feed; device_sampler = example indices only over PCIe, feed formed by tcar_form_batch"}

    gather = None
    quick = args.no_cpu_baseline and args.no_e2e                      # A/B runs of the tools: the step only
    if rank == 0 and world == 1 and not args.no_gather_roofline and not quick and N <= 200000:
        torch.cuda.empty_cache()
        gather = gather_roofline(dev)      # ~9 GB of its own tables and outputs, after everything else is measured

    # ---- the whole step against both rooflines (tracked per round): algorithmic flops and bytes of ONE step / ms_per_step ----------
    step_roof = None
    if world == 1 and not os.environ.get("TCAR_FORCE_DP"):
        Npad_ = ((N + 127) // 128) * 128
        ldh_, ldt_ = ((H + 63) // 64) * 64, 64
        ic_, pt_ = 2 * ldh_, 5 * ldt_
        n_par = Npad_ * ldh_
        opb_ = 4 if args.scoring == "f32" else (4 if args.scoring == "bf16x3" else 2)
        fl_step = B * (2.0 * N * (k_alg + k_alg + (H + 5 * Ht)) + 3.0e6 * (0.70 * mean_T + 1.02))
        by = {"logits (E planes + attout planes in, exp plane + statistics out)": 4 * (Npad_ * ic_ + B * ic_) + 2 * Npad_ * 160 + 4 * B * 160 + 2 * B * Npad_ + 8 * B * (Npad_ // 96),
              # (anchored softmax form, round 6: no pass over the plane — a fold launch reads the group sums and the packed attout
              #  planes and writes the per-row scaled plane of dE)
              ("ce fold (group sums in; row scales + scaled attout plane out)" if step_form0.get("ce_anchored") else "ce rescale (plane in place)"):
                  (B * (8 * (Npad_ // 96) + 6 * (ldh_ + pt_)) if step_form0.get("ce_anchored") else 4 * B * Npad_),
              "dX (dlogits + E planes in, split-K slabs out and back in)": opb_ * (B * Npad_ + Npad_ * ic_) + 2 * Npad_ * 160 + 8 * eng_splitk * B * (ic_ + 160),
              "dE (dlogits + attout planes in, item block + (q, z) out)": opb_ * (B * Npad_ + B * (ldh_ + pt_)) + 4 * N * ldh_ + 40 * N,
              "clip + Adam over the item table (g, m, v, w in; m, v, w + bf16 planes out)": 32 * n_par,
              "gather fwd + bwd, negatives (SURVEY.md 8(d))": int(B * (2 * (3536.0 * mean_T + 512) + 3 * (2536.0 * mean_T + 512) + 40000 + 60000)),
              "dense arena (857k weights: w, g, m, v)": 28 * 880000}
        tot = float(sum(by.values()))
        t_s = dt / args.steps
        step_roof = {"flops_per_step": fl_step, "bytes_per_step": tot, "bytes_by_part": {k: int(v) for k, v in by.items()},
                     "achieved_TFLOPs": round(fl_step / t_s / 1e12, 1), "frac_mfma": round(fl_step / t_s / 1e12 / PEAK_BF16_DENSE_TFLOPS, 4),
                     "achieved_GBps": round(tot / t_s / 1e9, 1), "frac_hbm": round(tot / t_s / 1e9 / PEAK_HBM_GBS, 4),
                     "note": "algorithmic flops (SURVEY.md 8(d): scoring 2N(820 + 820 + 570) + session side 3 (0.70 T + 1.02) MFLOP per "
                             "session) and algorithmic HBM bytes of one step in the default precision's data layout (every operand and "
                             "result of the big kernels once), divided by the headline ms_per_step"}
    if gather is not None and "gather_fwd" in kernels:
        gather["in_step"] = dict(kernels["gather_fwd"], note="the step's own gather launch (HIP events on its stream, second pass): latency bound")
    if rank == 0:
        labels = {"globo": "TCAR Globo-like fold 0", "adressa": "TCAR Adressa-like fold (active_t dwell, impression negatives)",
                  "mind": "TCAR MIND-like fold (one click time per session, neighbour negatives)",
                  "stress10m": "synthetic 10M-item catalog"}
        out = {"metric": "sessions/sec TCAR train on Globo (synthetic %s fold)" % args.config, "value": round(value, 1),
               "unit": "sessions/s",
               # what a user of main.py gets: the trainer loop over the whole fold, tail batches of every length bucket included
               # (details under end_to_end_sessions_per_s); `value` is the loop body over full batches of B sessions
               "trainer_loop_sessions_per_s": (e2e["value"] if e2e else None),
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "step_form": step_form0,
               "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": DTYPE_NOTE[args.scoring], "scoring": args.scoring, "data": "synthetic",
               "config": {"workload": "%s: N=%d items, %d-d content, B=%d/GPU, K=%d %s negatives, mean input length %.2f, "
                                      "full-catalog scoring, clip %d + Adam" %
                                      (labels[args.config], N, H, B, K, cfg["neg_mode"], mean_T, 150),
                          "name": args.config, "global_batch": B * world, "parallelism": "dp%d" % world},
               "timed_loop": ("device sampler (tcar_form_batch: batch formation + K negatives from HBM-resident stores) + "
                              "forward + backward + clip + Adam per step" if headline_sampler else
                              "forward + backward + clip + Adam per step over pre-formed resident feeds"),
               ("device_step_sessions_per_s" if headline_sampler else "sampler_in_loop_sessions_per_s"):
                   (round(other, 1) if other else None),
               "batches": "full batches of exactly B sessions only (the per-bucket tail batches of sampler.py:46-48 are in "
                          "end_to_end_sessions_per_s, which runs the trainer loop over the whole fold)",
               "roofline_pass": "per-kernel times (roofline, kernels) come from a SECOND, event-instrumented pass of the same steps "
                                "(ms_per_step_with_kernel_events); value / ms_per_step come from the un-instrumented headline pass",
               "ms_per_step_by_T": by_T,
               "step_roofline": step_roof,
               "roofline": roof, "gather_roofline": gather, "cpu_baseline": cpu, "end_to_end_sessions_per_s": e2e, "exchange": exchange,
               "kernels": kernels, "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 4),
               "ms_per_step_with_kernel_events": (round(dt_ev / args.steps * 1e3, 4) if dt_ev else None),
               "last_loss": round(last_loss, 4)}
        if dist is not None:
            # RCCL writes its version banner through C stdio, which is block-buffered when stdout is a file and would land BEHIND
            # the JSON line at exit: drain it first, so that the JSON line is the last thing this process prints
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
