"""Host time of the catalog-sharded step with LIVE collectives on a one-rank RCCL group (TCAR_FORCE_COLLECTIVES semantics), by
function: cProfile over N steps, the top entries by cumulative time.  Usage: python tools/shard_host_profile.py [direct|pg] [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else "direct"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
import tcar_amd  # noqa
from tcar_amd.host.model import initial_variables
from tcar_amd.host.synth import SynthFold
from tcar_amd.sharded import ShardedEngine

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
N, H, B, K = 46033, 250, 512, 20
fold = SynthFold(n_items=N, dim=H, n_train=60000, n_test=1000, seed=2020)
np.random.seed(2020)
params = initial_variables(N, H, 64, 0.002, 0.05, weight_seed=2020)
eng = ShardedEngine(params, fold.content, fold.mwdhm, device="cuda:0", scoring="bf16x3-mixed", group=dist.group.WORLD,
                    force_collectives=(mode != "none"), direct_rccl=(mode == "direct") if mode != "none" else None,
                    **({"world": 1, "rank": 0} if mode == "none" else {}))
import bench
rng = np.random.RandomState(2020)
batches, _ = bench.build_batches(fold, 16, B, K, rng, dict(bench.CONFIGS["globo"]), with_ids=True)
res = [eng.make_resident(b) for b in batches]
for i in range(30):
    eng.train_step(None, bt=res[i % len(res)])
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(steps):
    eng.train_step(None, bt=res[i % len(res)])
pr.disable()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("mode %s: host enqueue %.3f ms/step (under cProfile), with device drain %.3f ms/step" % (mode, t_enq / steps * 1e3, t_all / steps * 1e3))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(32)
print("\n".join(l[:170] for l in s.getvalue().splitlines()[:60]))
dist.destroy_process_group()
