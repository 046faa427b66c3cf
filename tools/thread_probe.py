"""Diagnostic: two engines stepped from two host threads at once vs each alone (tests/test_gpu_parity.py::
test_two_engines_in_flight_at_once_match_their_solo_runs).  Round 4: the Globo-size engine differed from its solo run in most steps
(observation 2); round 5: identical — the cause was hipcc's SLP-packed v_pk_fma_f32 op_sel:[0,1,0] beside another
engine's MFMA waves (profiles/r05_obs1_erratum.txt), and the library is built without the SLP vectorizer now.  Prints, per engine, the first step whose losses differ and
by how much.  Usage: python tools/thread_probe.py [flag_fork_mask|-1 for the default] [steps]"""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["TCAR_NO_PRIO"] = "1"
import numpy as np
import torch
import tcar_amd  # noqa
from tcar_amd.engine import TcarEngine
from test_gpu_parity import _case

mask = int(sys.argv[1]) if len(sys.argv) > 1 else -1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
H, Ht, K = 250, 64, 20
cases = [_case(46033, H, Ht, 512, 2, K, seed=61), _case(9000, H, Ht, 256, 3, K, seed=62)]


def run(case, stream, barrier=None, out=None, slot=0):
    params, content, mw, batch = case
    with torch.cuda.stream(stream):
        eng = TcarEngine(params, content, mw, scoring="bf16x3-mixed")
        if mask >= 0:
            eng.set_tuning(TCAR_FLAG_FORK=mask)
        bt = eng.make_resident(batch)
        losses = []
        for _ in range(steps):
            if barrier is not None:
                barrier.wait()
            losses.append(eng.train_step(None, bt=bt, defer_update=True).clone())
        eng.flush()
        eng.check_forks()
        res = torch.stack(losses).cpu().numpy()
    if out is not None:
        out[slot] = res
    return res


solo = [run(c, torch.cuda.Stream(priority=-1)) for c in cases]
again = [run(c, torch.cuda.Stream(priority=-1)) for c in cases]
for i in range(2):
    print("engine %d solo repeatable: %s" % (i, bool((solo[i] == again[i]).all())))
for trial in range(3):
    both = [None, None]
    barrier = threading.Barrier(2)
    th = [threading.Thread(target=run, args=(cases[i], torch.cuda.Stream(priority=-1), barrier, both, i)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        d = np.abs(solo[i].astype(np.float64) - both[i])
        bad = np.where(d.max(1) > 0)[0]
        if len(bad):
            s0 = int(bad[0])
            print("trial %d engine %d: first differing step %d (of %d differing), max |d| there %.3e (loss scale %.3e), rows differing %d / %d; "
                  "max |d| overall %.3e" % (trial, i, s0, len(bad), d[s0].max(), np.abs(solo[i][s0]).max(), int((d[s0] > 0).sum()), d.shape[1], d.max()))
        else:
            print("trial %d engine %d: identical" % (trial, i))
# the same two engines on ONE thread, alternating steps (interleaved on the host, one thread)
outs = [[], []]
engs = []
for c in cases:
    e = TcarEngine(c[0], c[1], c[2], scoring="bf16x3-mixed")
    if mask >= 0:
        e.set_tuning(TCAR_FLAG_FORK=mask)
    engs.append((e, e.make_resident(c[3])))
for _ in range(steps):
    for i, (e, bt) in enumerate(engs):
        outs[i].append(e.train_step(None, bt=bt, defer_update=True).clone())
for i, (e, bt) in enumerate(engs):
    e.flush()
    e.check_forks()
    r = torch.stack(outs[i]).cpu().numpy()
    print("one thread, alternating: engine %d identical to solo: %s" % (i, bool((r == solo[i]).all())))
