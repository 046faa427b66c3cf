"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, parameter packing
round-trips, the tensorised store equals the dict path, synthetic folds follow the data contract."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import tcar_amd  # noqa: F401
from tcar_amd import _lib
from tcar_amd.engine import ARENA, Geometry, VAR_ORDER
from tcar_amd.host.data import SessionStore
from tcar_amd.host.synth import SynthFold

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    _lib.build()
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "tcar_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|float\*|const char\*) (tcar_\w+)\(", header, flags=re.M))
    assert declared == set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s)
    assert lib.tcar_abi_version() == _lib.ABI_VERSION == 29
    # the binary carries the digest of the sources it was built from; the loader refuses a stale one
    assert lib.tcar_build_id().decode() == _lib.source_build_id() == _lib.binary_build_id()
    assert lib.tcar_gemm_splitk_effective(46080, 16) == 16
    assert lib.tcar_gemm_splitk_effective(64, 16) == 2


def test_tuning_switch_defaults():
    """Every TCAR_* switch of tcar_tuning_t (include/tcar_hip.h) is reachable by name and holds its documented default (one
    table in step.hip: name, field, default).  The library keeps NO mutable switch: tcar_tuning_set writes the caller's copy
    only, the process-wide values (tcar_tuning_defaults) never change."""
    lib = _lib.load()
    env = {k: v for k, v in os.environ.items() if k.startswith("TCAR_")}
    want = {"TCAR_BF16_TILE": 0, "TCAR_BF16_KS": 2, "TCAR_WGRAD_KS": 1536, "TCAR_GATHER_BIG_ROWS": 16384, "TCAR_MHA_MFMA": 1,
            "TCAR_SORT_SCATTER": 1, "TCAR_DET_SMALL": 1, "TCAR_FUSED_CE": 2, "TCAR_ONEHOT_TIME": 2, "TCAR_FLAG_FORK": 4095,
            "TCAR_CE_FOLD": 1024, "TCAR_PROJ_SPLIT_ROWS": 1024}
    assert len(want) == 12                               # at most twelve live switches
    header = open(os.path.join(ROOT, "include", "tcar_hip.h")).read()
    block = header[header.index("typedef struct {\n  int32_t bf16_tile"):header.index("} tcar_tuning_t;")]
    documented = set(re.findall(r"/\* (TCAR_[A-Z0-9_]+)\b", block))
    assert documented == set(want), documented ^ set(want)
    assert len(re.findall(r"int32_t [a-z0-9_]+;", block)) == len(_lib.TUNING_FIELDS) == len(want)
    t = _lib.tuning()
    for name, default in want.items():
        if name in env:
            continue                                     # the process was started with an override
        mine = _lib.tuning()
        old = lib.tcar_tuning_set(C.byref(mine), name.encode(), 12345)
        assert old == default, (name, old, default)
        assert lib.tcar_tuning_set(C.byref(mine), name.encode(), old) == 12345
    assert lib.tcar_tuning_set(C.byref(t), b"TCAR_NO_SUCH_SWITCH", 1) == -2147483648
    # a caller's copy does not leak into the process-wide values
    lib.tcar_tuning_set(C.byref(t), b"TCAR_BF16_TILE", 777)
    assert _lib.tuning().bf16_tile == (int(env["TCAR_BF16_TILE"]) if "TCAR_BF16_TILE" in env else 0)
    assert _lib.tuning(bf16_tile=5).bf16_tile == 5 and _lib.tuning(TCAR_WGRAD_KS=3).wgrad_ks == 3


def test_library_keeps_no_per_thread_or_mutable_global_state():
    """re-entrant, no global mutable state, per-device handles passed in.  The fork slots live in
    tcar_ctx_t.fork_host, launch options travel as arguments (TcarOpt): no `thread_local` and no writable namespace-scope
    variable may come back into csrc/ (the only statics are the once-per-device attribute masks and const tables)."""
    csrc = os.path.join(ROOT, "session-based-news-recommendation_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".h")):
            continue
        src = open(os.path.join(csrc, fn)).read()
        code = re.sub(r"//[^\n]*", "", src)
        assert "thread_local" not in code, fn
        for m in re.finditer(r"^\s*static\s+(?!const\b|constexpr\b|inline\b|int\s+\w+\(|__device__|__global__)([^;\n(]*)[;=]", code, re.M):
            assert "TcarOnce" in m.group(0) or "const" in m.group(0), (fn, m.group(0))
    assert _lib.load().tcar_fork_state_bytes() >= 16 * 24


def test_c_abi_rejects_bad_arguments_without_touching_a_gpu():
    lib = _lib.load()
    # misaligned leading dimension / null pointers are refused before any launch
    assert lib.tcar_gemm_f32(0, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, 1, None) == -1
    assert lib.tcar_gemm_f32(7, 4, 4, 4, 16, 4, 16, 4, 16, 4, None, 0, 0, 1, None) == -1
    assert lib.tcar_gemm_f32(1, 4, 4, 6, 16, 8, 16, 8, 16, 8, None, 0, 0, 1, None) == -1     # K % 4 for k-contiguous
    assert lib.tcar_softmax_ce(2, 10, 16, 10, 16, 16, None) == -1                              # ld % 4
    assert lib.tcar_gemm_f32(0, 0, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, 1, None) == 0  # empty problem is a no-op
    # the entry points added with the fused step: same contract (error code, never a launch or an exception)
    import ctypes as C
    from tcar_amd._lib import Dims, Segments
    d = Dims(100, 250, 64, 256, 64)
    assert lib.tcar_eval_rows(2, 10, None, 12, None, 20, None, None, None, None) == -1
    assert lib.tcar_eval_rows(0, 10, None, 12, None, 20, None, None, None, None) == 0          # B = 0: no-op
    assert lib.tcar_gemm_bf16(1, 4, 4, 30, 16, 16, 32, 4, 16, 16, 32, 4, 16, 4, None, 0, 0, 3, 1, None) == -1   # K % 32
    assert lib.tcar_gemm_bf16_perm(2, 4, 4, 32, 16, 16, 32, 32, 16, 16, 32, 32, 16, 4, None, 0, 0, 16, 64, 3, 1, None) == -1
    assert lib.tcar_neg_fwd(C.byref(d), 2, 3, None, None, None, 0.01, None, None, None, None) == -1
    assert lib.tcar_neg_scatter(C.byref(d), 2, 3, None, None, None, None, None, None, 0.01, None, None) == -1
    assert lib.tcar_neg_fwd(C.byref(d), 2, 0, None, None, None, 0.01, None, None, None, None) == 0           # K = 0: no-op
    assert lib.tcar_splitk_reduce_dact(None, 2, 4, 6, 8, None, 0, 0, None, 0, 0, None, None, 0, None, None) == -1  # N % 4
    segs = Segments()
    assert lib.tcar_clip_adam_early(16, 16, 16, 16, C.byref(segs), 16, 832, 16, 16, 16, 10, 256, 0, 16, 16, 16, 1.0, 1e-3, 0.9,
                                    0.999, 1e-8, None, None, 0, None, 0, None, None) == -1              # no bitmap
    assert lib.tcar_clip_adam_rest(16, 832, 16, 16, 16, 10, 256, 0, 16, 16, 16, 1.0, 1e-3, 0.9, 0.999, 1e-8, None, None, 0,
                                   None, None) == -1
    assert lib.tcar_cand_time_bwd_indexed(C.byref(d), None, None, None, None, 0, None, None, None) == -1


def test_geometry_and_arena_cover_all_variables():
    g = Geometry(46033, 250, 64)
    assert (g.ldh, g.ldt, g.ic, g.pt, g.ct, g.ek, g.Npad) == (256, 64, 512, 320, 128, 832, 46080)
    refs = [a[1] for a in ARENA]
    assert sorted(refs + ["item_emb"]) == sorted(VAR_ORDER) and len(VAR_ORDER) == 23
    # time tables + dwell table are contiguous and in the order tcar_grads_t requires
    assert [a[0] for a in ARENA[1:7]] == ["month", "day", "week", "hour", "minute", "dur"]
    idx = g.idx("2H")
    assert idx[0] == 0 and idx[249] == 249 and idx[250] == 256 and idx[-1] == 505


def test_store_from_dicts_equals_synth_store():
    fold = SynthFold(n_items=80, dim=6, n_train=300, n_test=30, seed=4, active_t=True)
    len_d, sess, times = fold.to_dicts(fold.train, with_active=True)
    st = SessionStore.from_dicts(sess, times)
    assert st.n == fold.train.n
    np.testing.assert_array_equal(st.items, fold.train.items)
    np.testing.assert_array_equal(st.pub, fold.train.pub)
    np.testing.assert_array_equal(st.clk, fold.train.clk)
    np.testing.assert_array_equal(st.gap_active, fold.train.gap_active)
    # gap_delta is defined on input positions only
    for e in range(st.n):
        o, o2 = st.off[e], st.off[e + 1]
        np.testing.assert_array_equal(st.gap_delta[o:o2 - 1], fold.train.gap_delta[o:o2 - 1])
    assert sum(len(v) for v in len_d.values()) == st.n


def test_synth_fold_contract():
    fold = SynthFold(n_items=500, dim=10, n_train=2000, n_test=100, seed=1)
    assert fold.content.shape == (501, 10) and not fold.content[0].any()
    m = fold.mwdhm
    assert m[:, 0].min() >= 1 and m[:, 0].max() <= 12 and m[:, 2].min() >= 1 and m[:, 2].max() <= 7
    assert m[:, 3].min() >= 1 and m[:, 3].max() <= 24 and m[:, 4].min() >= 1 and m[:, 4].max() <= 60
    st = fold.train
    assert st.items.min() >= 1 and st.items.max() <= 500 and st.in_len.min() >= 1 and st.in_len.max() <= 40
    b = st.batch_arrays(np.where(st.in_len == 2)[0][:5], "click_delta")
    assert b["seq"].shape == (5, 2) and b["gap"].max() <= 11 and b["cw"].max() <= 6 and b["ch"].max() <= 23
    with pytest.raises(ValueError):
        st.batch_arrays(np.array([np.where(st.in_len == 1)[0][0], np.where(st.in_len == 2)[0][0]]))
    nb = fold.neighbor_dict(k=3)      # generate_neighbor.py:13-16 with a window of 3: 3 earlier + the item itself + 2 later
    assert all(i in v and 3 <= len(v) <= 6 for i, v in nb.items())


def test_load_fold_reads_the_reference_pickle_layout(tmp_path):
    """host/data.load_fold is data_partition (util.py:20-56) minus the hard-coded path: a fold written in that layout
    comes back as the same dicts, and its tensorised store yields the same batches as the synthetic store it was
    written from."""
    import random
    from helpers import write_reference_fold
    from tcar_amd.host.data import SessionStore, load_fold
    from tcar_amd.host.sampler import Sampler
    from tcar_amd.host.synth import SynthFold
    fold = SynthFold(n_items=120, dim=16, n_train=300, n_test=60, seed=3, active_t=True)
    base = str(tmp_path) + "/"
    item_dict = write_reference_fold(base, fold, foldnum=2)
    train, test, items, neighbor, content, publish_time, _ = load_fold(base, 2)
    assert items == item_dict and len(publish_time) == 2 and np.array_equal(publish_time[1], fold.mwdhm)
    ref_nb = fold.neighbor_dict()
    assert np.array_equal(content, fold.content) and set(neighbor) == set(ref_nb)
    assert all(np.array_equal(neighbor[k], ref_nb[k]) for k in ref_nb)
    want = fold.to_dicts(fold.train, with_active=True)
    assert train[0] == want[0] and train[1] == want[1] and train[2] == want[2]
    assert set(test[0]) == set(fold.to_dicts(fold.test, with_active=True)[0])

    def batches(data, store):
        random.seed(5)
        np.random.seed(5)
        len_d = {k: list(v) for k, v in data[0].items()}
        s = Sampler(len_d, data[1], data[2], neighbor, items, 4, batch_size=32, store=store)
        out = []
        while s.has_next():
            out.append(s.next_batch_arrays())
        return out

    a = batches(train, None)                                   # store built from the loaded dicts
    b = batches(want, None)                                    # store built from the in-memory dicts
    assert len(a) == len(b) > 0
    for x, y in zip(a, b):
        for k in x:
            assert (x[k] is None and y[k] is None) or np.array_equal(x[k], y[k]), k


def test_graft_entry_build_runs():
    """the driver's per-round build check: compiles (or reuses) libtcar_hip.so, loads it, checks the ABI, imports the oracle"""
    import __graft_entry__ as g
    g.build()


def test_impression_mode_on_a_store_with_integer_example_ids():
    """Large synthetic folds key examples by integer id; impression mode then takes the session id from the store's
    impression_key column (the value sampler.py:96 parses out of "sid_len" keys).  Exact and vectorised paths obey the
    same rules: picks are catalog items of the session's impression list, padded with uniform draws."""
    import random
    from tcar_amd.host.sampler import Sampler
    from tcar_amd.host.synth import SynthFold
    fold = SynthFold(n_items=300, dim=16, n_train=2000, n_test=100, seed=3)
    st = fold.train
    imp = fold.impression_dict(st, unknown=0.0)             # every impression is a catalog item: no padding can occur
    len_d = {int(T): np.where(st.in_len == T)[0].tolist() for T in np.unique(st.in_len)}
    for fast in (False, True):
        random.seed(1)
        np.random.seed(1)
        s = Sampler({k: list(v) for k, v in len_d.items()}, None, None, imp, fold.item_dict, 10, batch_size=64,
                    neg_mode="impression", store=st, neg_fast=fast, verbose=False)
        n = 0
        while s.has_next() and n < 5:
            keys = s.session_id_batches[s.batch_i]
            f = s.next_batch_arrays()
            for b, e in enumerate(keys):
                allowed = {fold.item_dict[x] - 1 for x in imp[int(st.impression_key[e])]}
                assert set(f["neg"][b].tolist()) <= allowed, (fast, b)
            n += 1


def test_category_table_accepts_any_label_type_and_missing_items():
    """The reference only compares categories with != (model_combine.py:180,190): string labels (MIND) must work, and an
    item without a category differs from everything (the reference raises KeyError lazily)."""
    from tcar_amd.host import metrics as M
    rev = {0: "a", 1: "b", 2: "c", 3: "d"}
    cat = M.category_table(rev, {"a": "news", "b": "sport", "c": "news"}, 4)
    assert cat[0] == cat[2] != cat[1]
    assert cat[3] not in (cat[0], cat[1])
    ints = M.category_table({0: 10, 1: 11}, {10: 7, 11: 7}, 2)
    assert ints[0] == ints[1]
    ild = M.ild_batch(np.array([[0, 1, 2]]), cat)
    assert abs(float(ild[0]) - 4.0 / 6.0) < 1e-12          # pairs (0,1),(1,0),(1,2),(2,1) differ


def test_torch_ops_are_registered_without_a_gpu():
    """torch.ops.tcar.* (SURVEY.md 8(b) op-level boundary) register at import, with schemas; only a CUDA kernel exists."""
    import torch
    from tcar_amd import torch_ops
    for name in torch_ops.OPS:
        op = getattr(torch.ops.tcar, name)
        assert "Tensor" in str(op.default._schema)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.tcar.rank_topk(torch.zeros(2, 8), torch.zeros(2, dtype=torch.int32), 8, 3)     # CPU tensors: no fallback


def test_build_neighbor_equals_the_reference_get_neighbor():
    """host/data.py:build_neighbor against outputs of the REAL generate_neighbor.get_neighbor (generate_neighbor.py:7-21;
    fixture written by tests/golden/make_reference_fixtures.py): key order, every list element for element — window of 100
    earlier items + the item itself + 99 later ones, ties as numpy's default argsort leaves them, object-array (datetime)
    input, and the pad branch with the reference's draws from `random`."""
    import datetime
    import json
    import os
    import random
    from helpers import GOLD
    from tcar_amd.host.data import build_neighbor
    with open(os.path.join(GOLD, "reference_neighbor.json")) as fh:
        cases = json.load(fh)
    assert {c["name"] for c in cases} == {"ints300", "datetimes130", "ints60_pad"}
    for c in cases:
        pt = [datetime.datetime.fromisoformat(x) for x in c["publish_time"]] if c["kind"] == "datetime" else c["publish_time"]
        random.seed(c["seed"])
        got = build_neighbor(pt)
        assert [int(k) for k in got.keys()] == c["keys"], c["name"]
        for k, want in zip(c["keys"], c["lists"]):
            assert [int(x) for x in got[k]] == want, (c["name"], k)
        n = len(pt)
        if n >= 200:       # interior items: 100 earlier + self + 99 later
            mid = c["keys"][n // 2]
            assert len(got[mid]) == 200 and int(got[mid][100]) == mid
    # the synthetic folds and `--neg_mode neighbor` without a neighbour file go through the same function
    from tcar_amd.host.synth import SynthFold
    fold = SynthFold(n_items=300, dim=8, n_train=50, n_test=10, seed=3)
    nb = fold.neighbor_dict()
    ref = build_neighbor(fold.publish_ts)
    assert all((nb[k] == ref[k]).all() for k in ref) and all(int(k) in set(int(x) for x in v) for k, v in nb.items())


def test_built_library_has_no_packed_f32_op_with_a_low_from_high_operand_select(tmp_path):
    """DESIGN.md §7, observation 1 (root-caused in round 5): on gfx950 `v_pk_fma_f32 ... op_sel:[0,1,0]` — hipcc's SLP vectorizer
    emits it for `acc = fma(row, s[r], acc)` — loses its low-half product in lanes 48-63 while another wave of the SIMD issues
    MFMAs (tools/micro/pkfma_lds.hip, tools/obs1_probe.py).  The library is therefore built with -fno-slp-vectorize (_lib.SAFE_FLAGS);
    this test disassembles EVERY gfx950 code object of the built binary and refuses any packed-f32 instruction that carries an
    `op_sel:[...]` modifier (a low result fed from a high dword); plain packed ops and op_sel_hi forms, which the same loops ran
    billions of times without a fault, are allowed."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM binutils not found")
    assert "-fno-slp-vectorize" in _lib.SAFE_FLAGS
    lib = _lib.build()
    fat = str(tmp_path / "fat.bin")
    subprocess.run([tools[0], "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    assert len(_lib.SOURCES) - 1 <= len(starts) <= len(_lib.SOURCES)        # one bundle per translation unit with device code (buildid.hip has none)
    n_pk, bad = 0, []
    for i, st in enumerate(starts):
        part = str(tmp_path / ("b%d.bin" % i))
        with open(part, "wb") as fh:
            fh.write(data[st:starts[i + 1] if i + 1 < len(starts) else len(data)])
        co = str(tmp_path / ("b%d.co" % i))
        subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part,
                        "--output=" + co], check=True)
        if not os.path.exists(co) or os.path.getsize(co) == 0:
            continue                                                       # a translation unit without device code
        dis = subprocess.run([tools[2], "-d", co], check=True, stdout=subprocess.PIPE).stdout.decode(errors="replace")
        for line in dis.splitlines():
            if re.search(r"\bv_pk_(fma|mul|add)_f32\b", line):
                n_pk += 1
                if "op_sel:[" in line:
                    bad.append(line.strip())
    assert not bad, bad[:5]
    assert n_pk < 400        # (what is left comes from explicit float2 arithmetic: ~120 plain v_pk_add / v_pk_mul in segsum.hip and mha.hip)


def test_device_sampler_chunks_grow_to_the_full_size_and_tile_the_plan():
    """DeviceSampler.planned forms the planned batches in chunks of 2, 2, 4, 8, ... up to CHUNK batches (a loop's first step waits for
    two batches only, and the host launches a 16-batch chunk only once it is that far ahead): the chunks tile [0, n) in order, none
    is empty or larger than CHUNK, sizes never shrink before the last chunk."""
    from tcar_amd.device_sampler import DeviceSampler
    for n, ch, want in ((20, 16, [2, 2, 4, 8, 4]), (200, 16, [2, 2, 4, 8] + [16] * 11 + [8]), (1, 16, [1]), (2, 16, [2]), (3, 16, [2, 1]),
                        (5, 1, [1] * 5), (40, 4, [2, 2] + [4] * 9)):
        lo, hi = DeviceSampler.chunk_bounds(n, ch)
        assert [h - l for l, h in zip(lo, hi)] == want
        assert lo[0] == 0 and hi[-1] == n and lo[1:] == hi[:-1]
        assert all(0 < h - l <= max(ch, 1) for l, h in zip(lo, hi))
