"""Loader (and in-tree builder) of ``libtcar_hip.so`` — the C-ABI of include/tcar_hip.h.

The product path has NO CPU fallback: if the shared library is missing or a symbol is absent, importing the
engine raises.  `build()` cross-compiles the HIP sources for gfx950 with hipcc (works without a GPU).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# (TCAR_LIB: load THIS binary instead — a diagnostic build of the same sources, e.g. tools/obs1_probe.py's -DTCAR_OBS1_DIAG library;
#  it must carry the same source digest and ABI, and it is never built or overwritten by build())
LIB_PATH = os.environ.get("TCAR_LIB") or os.path.join(PKG_DIR, "libtcar_hip.so")
SOURCES = ["gemm_f32.hip", "gemm_bf16.hip", "embed.hip", "pool.hip", "score.hip", "optim.hip", "step.hip", "mha.hip",
           "sampler.hip", "norm.hip", "shard.hip", "segsum.hip", "query.hip", "buildid.hip"]
BUILD_ID_TU = "buildid.hip"        # the one translation unit that carries the digest of all sources
NVAR = 22
NSLOT = 32

# every symbol include/tcar_hip.h declares
ABI_VERSION = 29          # == TCAR_ABI_VERSION of include/tcar_hip.h (struct mirrors below)

SYMBOLS = ["tcar_gather_clip_fwd", "tcar_gather_clip_bwd", "tcar_scatter_add_rows", "tcar_cand_time_fwd", "tcar_cand_time_bwd_indexed", "tcar_cand_time_ws_floats", "tcar_cand_time_bwd",
           "tcar_gemm_f32", "tcar_gemm_f32_grouped", "tcar_gemm_x3_grouped", "tcar_gemm_bf16", "tcar_gemm_bf16_perm", "tcar_gemm_bf16_variant", "tcar_gemm_bf16_ce", "tcar_ce_finish", "tcar_ce_anchor_fold", "tcar_gemm_bf16_ce_anchor", "tcar_reduce_dact_onehot_scaled", "tcar_ce_shard_stats", "tcar_ce_rescale", "tcar_time_onehot", "tcar_time_scores", "tcar_time_scores_clip", "tcar_attout_finish_scores", "tcar_gemm_bf16_dx_onehot", "tcar_gemm_bf16_dx_onehot_tuned", "tcar_reduce_dact_onehot", "tcar_gemm_bf16_de_qz", "tcar_cand_time_bwd_onehot", "tcar_query_mlp", "tcar_query_mlp_bwd", "tcar_flag_fork_selftest", "tcar_split_bf16", "tcar_splitk_reduce", "tcar_gemm_splitk_effective", "tcar_attn_pool_fwd",
           "tcar_attn_pool_bwd", "tcar_attn_pool_bwd_q", "tcar_softmax_ce", "tcar_neg_term", "tcar_neg_fwd", "tcar_neg_scatter", "tcar_splitk_reduce_dact",
           "tcar_dact_colsum", "tcar_rank_topk", "tcar_eval_rows", "tcar_eval_diversity",
           "tcar_sqnorm", "tcar_clip_adam", "tcar_clip_adam_2d", "tcar_clip_adam_2d_bf16", "tcar_cand_time_fwd_bf16", "tcar_softmax_ce_bf16",
           "tcar_mha_core_fwd", "tcar_mha_core_bwd", "tcar_layernorm_fwd", "tcar_layernorm_bwd", "tcar_clip_adam_all", "tcar_clip_adam_early", "tcar_clip_adam_rest", "tcar_clip_adam_rest_keep", "tcar_abi_version", "tcar_build_id", "tcar_tuning_defaults", "tcar_tuning_set", "tcar_fork_state_bytes", "tcar_ctx_bytes", "tcar_flag_poll_expire", "tcar_gather_clip_fwd_tuned", "tcar_gemm_bf16_tuned", "tcar_mha_core_fwd_tuned", "tcar_mha_core_bwd_tuned", "tcar_form_batch", "tcar_segsum_ws_bytes", "tcar_segsum_index", "tcar_segsum_rows_buffer", "tcar_segsum_norms_buffer",
           "tcar_segsum_apply", "tcar_sqnorm_det", "tcar_softmax_stats", "tcar_softmax_combine", "tcar_softmax_combine_rowstat", "tcar_softmax_grad", "tcar_neg_scatter_range",
           "tcar_step_session_forward", "tcar_shard_score", "tcar_shard_backward", "tcar_shard_finish", "tcar_step_session_backward", "tcar_scatter_add_rows_packed", "tcar_shard_begin", "tcar_shard_join", "tcar_shard_step_local", "tcar_colsum_det", "tcar_fold_slabs", "tcar_gather_clip_bwd_sqnorm", "tcar_graph_probe", "tcar_attn_pool_bwd_det", "tcar_attn_pool_fwd_slabs", "tcar_attn_pool_bwd_slabs", "tcar_small_tables_bwd_det", "tcar_small_det_ws_floats", "tcar_shard_pack_head", "tcar_shard_unpack_head", "tcar_shard_pack_ids", "tcar_step_forward",
           "tcar_step_backward_local", "tcar_step_finish", "tcar_step_update", "tcar_train_step", "tcar_train_step_deferred", "tcar_eval_step",
           "tcar_step_form", "tcar_shard_form", "tcar_step_dense_norms"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


HEADERS = [os.path.join(CSRC, "tcar_common.h"), os.path.join(CSRC, "tcar_bf16_layout.h"),
           os.path.join(PKG_DIR, "..", "include", "tcar_hip.h")]
_ID_MARK = b"TCAR_BUILD_ID="


def _digest(paths) -> str:
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:32]


def source_build_id() -> str:
    """Digest of every source the library is compiled from (csrc/*.hip, the two csrc headers, include/tcar_hip.h) and of the
    code-generation flags every translation unit must be built with (SAFE_FLAGS): a binary built without them is stale."""
    import hashlib
    base = _digest([os.path.join(CSRC, s) for s in SOURCES] + HEADERS)
    return hashlib.sha256((base + "|" + " ".join(SAFE_FLAGS)).encode()).hexdigest()[:32]


def binary_build_id(path: str = LIB_PATH):
    """The digest baked into a built libtcar_hip.so (tcar_build_id()), read from the file without loading it."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(_ID_MARK)
    if i < 0:
        return None
    j = blob.find(b"\0", i)
    return blob[i + len(_ID_MARK):j].decode(errors="replace")


# Every translation unit is compiled WITHOUT the SLP vectorizer.  hipcc 7.2 at -O3 packs adjacent scalar f32 FMAs into v_pk_fma_f32 with
# operand selects, and on gfx950 `v_pk_fma_f32 ... op_sel:[0,1,0]` (the low result takes the HIGH dword of the second source: what
# the vectorizer emits for `acc = fma(row, s[r], acc)` with four consecutive s) loses its low-half product in lanes 48-63, now and
# then, while another wave of the same SIMD issues MFMAs — DESIGN.md §7 observation 1; repro tools/micro/pkfma_lds.hip and
# tools/obs1_probe.py, evidence profiles/r05_obs1_erratum.txt.  The step co-schedules such kernels with the scoring GEMMs on
# purpose.  Measured perf-neutral (profiles/r05_ab_experiments.txt); tests/test_host_logic.py scans the built binary for the form.
SAFE_FLAGS = ["-fno-slp-vectorize"]


def have_sources() -> bool:
    return all(os.path.exists(os.path.join(CSRC, s)) for s in SOURCES) and all(os.path.exists(h) for h in HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -O3 -shared -fPIC csrc/*.hip -> libtcar_hip.so (in-tree).  The digest of the sources
    is compiled in (tcar_build_id); a library whose digest differs from the sources next to it is rebuilt.  Objects are
    cached per source file (digest of the file + headers), so an edit recompiles one translation unit."""
    if not have_sources():
        raise FileNotFoundError("HIP sources missing under " + CSRC)
    want = source_build_id()
    if os.environ.get("TCAR_LIB"):
        # a diagnostic binary named by TCAR_LIB is only ever VERIFIED here: build() never links over it (its flags differ from
        # SAFE_FLAGS on purpose — tools/micro/build_obs1.sh), a stale one is an error, not something to overwrite
        if not os.path.exists(LIB_PATH) or binary_build_id() != want:
            raise RuntimeError("TCAR_LIB=%s does not carry the digest of the sources next to it (%s): rebuild it with the script "
                               "that made it, or unset TCAR_LIB" % (LIB_PATH, want))
        return LIB_PATH
    if not force and os.path.exists(LIB_PATH) and binary_build_id() == want:
        return LIB_PATH
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(o)
        extra = os.environ.get("TCAR_HIPCC_FLAGS", "").split()      # e.g. -DTCAR_GEMM_DIAG (diagnostic kernel forms, tools/gemm_variants.sh)
        dig = _digest([src] + HEADERS) + (":" + want if s == BUILD_ID_TU else "") + ":" + " ".join(SAFE_FLAGS + extra)
        stamp = o + ".digest"
        if not force and os.path.exists(o) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"] + SAFE_FLAGS + extra + ["-c", src, "-o", o]
        if s == BUILD_ID_TU:
            cmd.insert(1, '-DTCAR_BUILD_ID="%s"' % want)
        procs.append((cmd, stamp, dig, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, stamp, dig, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
        with open(stamp, "w") as f:
            f.write(dig)
        if verbose and out:
            print(out.decode(errors="replace"))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(link), r.stdout.decode(errors="replace")))
    return LIB_PATH


# ------------------------------------------------------------------------------------------- ctypes mirrors
class Dims(C.Structure):
    _fields_ = [("n_items", C.c_int32), ("H", C.c_int32), ("Ht", C.c_int32), ("ldh", C.c_int32), ("ldt", C.c_int32)]


class Tables(C.Structure):
    _fields_ = [("E", C.c_void_p), ("pos", C.c_void_p), ("time", C.c_void_p * 5), ("dur", C.c_void_p)]


class Grads(C.Structure):
    _fields_ = [("g_item", C.c_void_p), ("g_pos", C.c_void_p), ("g_time", C.c_void_p * 5), ("g_dur", C.c_void_p),
                ("sqn", C.c_void_p), ("slot_item", C.c_int32), ("slot_pos", C.c_int32),
                ("slot_time", C.c_int32 * 5), ("slot_dur", C.c_int32), ("rows_out", C.c_void_p), ("norms_out", C.c_void_p), ("rows_ld", C.c_int64), ("skip_small", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("K", C.c_int32), ("seq", C.c_void_p),
                ("pub", C.c_void_p * 5), ("cw", C.c_void_p), ("ch", C.c_void_p), ("gap", C.c_void_p),
                ("label", C.c_void_p), ("neg", C.c_void_p)]


class Store(C.Structure):
    """mirror of tcar_store_t"""
    _fields_ = [("off", C.c_void_p), ("items", C.c_void_p), ("pub", C.c_void_p), ("clk", C.c_void_p),
                ("gap_active", C.c_void_p), ("gap_delta", C.c_void_p), ("n_examples", C.c_int64)]


class NegSrc(C.Structure):
    """mirror of tcar_negsrc_t"""
    _fields_ = [("mode", C.c_int32), ("off", C.c_void_p), ("flat", C.c_void_p), ("slot_of_example", C.c_void_p),
                ("n_lists", C.c_int64)]


class Shard(C.Structure):
    """mirror of tcar_shard_t"""
    _fields_ = ([("world", C.c_int32), ("cap", C.c_int32), ("n0", C.c_int32), ("n_loc", C.c_int32),
                 ("att_all", C.c_void_p), ("ld_att", C.c_int64)]
                + [(n, C.c_void_p) for n in ("lab_all", "logits", "stats", "lse", "ce", "a16h", "a16l", "ap16h", "ap16l",
                                             "dl16h", "dl16l", "slabs", "dx")]
                + [("head_K", C.c_int32), ("neg_all", C.c_void_p), ("coef_all", C.c_void_p), ("aps16h", C.c_void_p), ("scale2", C.c_void_p), ("n_total", C.c_int32)])


class Segments(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("off", C.c_int64 * NSLOT), ("len", C.c_int64 * NSLOT),
                ("slot", C.c_int32 * NSLOT)]


class GemmDesc(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("A", C.c_void_p * 3), ("B", C.c_void_p * 3), ("lda", C.c_int64 * 3),
                ("ldb", C.c_int64 * 3), ("K", C.c_int32 * 3), ("C", C.c_void_p), ("ldc", C.c_int64),
                ("bias", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("act", C.c_int32), ("beta", C.c_int32),
                ("splitk", C.c_int32), ("atomic", C.c_int32),
                ("dact", C.c_int32), ("dact_y", C.c_void_p), ("ld_dact_y", C.c_int64), ("colsum", C.c_void_p),
                ("plane_hi", C.c_void_p), ("plane_lo", C.c_void_p), ("plane_inner", C.c_int32), ("plane_col0", C.c_int32),
                ("pack_hi", C.c_void_p), ("pack_lo", C.c_void_p), ("pack_inner", C.c_int32), ("pack_c0", C.c_int32),
                ("pack_c1", C.c_int32)]


_WS = ["x_icp", "x_pt", "x_act", "click_t", "pre1", "pre2", "q1", "q", "alpha", "pooled", "attout", "logits", "ce",
       "neg_fb", "loss", "neg_coef", "negpart", "dattout", "dpooled", "dq", "dq1", "dclick", "slabs", "dx_icp", "dx_pt", "dx_act", "dpre1", "dpre2"]


class Ctx(C.Structure):
    """mirror of tcar_ctx_t (include/tcar_hip.h)"""
    _fields_ = ([("d", Dims), ("splitk", C.c_int32), ("slot_of", C.c_int32 * NVAR), ("slot_item", C.c_int32),
                 ("b1", C.c_float), ("b2", C.c_float), ("eps", C.c_float), ("clip", C.c_float),
                 ("neg_weight", C.c_float),
                 ("E", C.c_void_p), ("W", C.c_void_p), ("Gx", C.c_void_p), ("M", C.c_void_p), ("V", C.c_void_p),
                 ("arena_n", C.c_int64), ("off", C.c_int64 * NVAR),
                 ("big", C.c_void_p), ("Mi", C.c_void_p), ("Vi", C.c_void_p), ("sqn_dense", C.c_void_p),
                 ("use_dense", C.c_void_p), ("mwdhm", C.c_void_p), ("inv_n", C.c_void_p),
                 ("inv_off", C.c_void_p), ("ct_ws", C.c_void_p), ("segs_all", Segments), ("segs_dense", Segments)]
                + [(n, C.c_void_p) for n in _WS]
                + [("rank", C.c_void_p), ("topk", C.c_void_p), ("scoring", C.c_int32), ("scoring_bwd", C.c_int32)]
                + [(n, C.c_void_p) for n in ("e16h", "e16l", "a16h", "a16l", "ap16h", "ap16l", "dl16h", "dl16l")]
                + [("stream2", C.c_void_p), ("ev", C.c_void_p * 6), ("adam_bitmap", C.c_void_p), ("et_perm", C.c_void_p), ("ev_start", C.c_void_p), ("ev_stop", C.c_void_p),
                   ("ev_n", C.c_int32), ("ev_cursor", C.c_void_p), ("stream3", C.c_void_p), ("ev3", C.c_void_p),
                   ("segsum_ws", C.c_void_p), ("segsum_bytes", C.c_int64), ("gw_rows", C.c_void_p), ("wgrad_slabs", C.c_void_p), ("wgrad_slab_floats", C.c_int64),
                   ("proj_slabs", C.c_void_p), ("proj_slab_floats", C.c_int64),
                   ("ce_ws", C.c_void_p), ("ce_ws_floats", C.c_int64), ("ce_geo", C.c_void_p),
                   ("oh16", C.c_void_p), ("p16h", C.c_void_p), ("p16l", C.c_void_p),
                   ("tclip", C.c_void_p), ("dP", C.c_void_p), ("qz", C.c_void_p),
                   ("sig_dev", C.c_void_p), ("fork_host", C.c_void_p), ("sig_err_host", C.c_void_p), ("tune", C.c_void_p),
                   ("fold_scratch", C.c_void_p), ("fold_scratch_words", C.c_int32),
                   ("small_det_ws", C.c_void_p), ("small_det_ws_floats", C.c_int64),
                   ("ce_rowscale", C.c_void_p), ("aps16h", C.c_void_p), ("ce_form", C.c_void_p)])


TUNING_FIELDS = ["bf16_tile", "bf16_ks", "wgrad_ks", "gather_big_rows", "mha_mfma", "sort_scatter", "det_small", "fused_ce", "onehot_time",
                 "flag_fork", "ce_fold", "proj_split_rows"]


class Tuning(C.Structure):
    """mirror of tcar_tuning_t: a caller-owned copy of the TCAR_* switches (tcar_ctx_t.tune, the *_tuned entry points)"""
    _fields_ = [(n, C.c_int32) for n in TUNING_FIELDS]


def tuning(**overrides) -> Tuning:
    """The process-wide switch values (shipped defaults + TCAR_* environment) with `overrides` applied, e.g.
    tuning(TCAR_BF16_TILE=384) or tuning(bf16_tile=384).  The library itself keeps no mutable switch."""
    lib = load()
    t = Tuning()
    check(lib.tcar_tuning_defaults(C.byref(t)), "tcar_tuning_defaults")
    for k, v in overrides.items():
        if k.startswith("TCAR_"):
            if lib.tcar_tuning_set(C.byref(t), k.encode(), int(v)) == -2 ** 31:
                raise KeyError(k)
        else:
            if k not in TUNING_FIELDS:
                raise KeyError(k)
            setattr(t, k, int(v))
    return t


class TcarError(RuntimeError):
    pass


_LIB = None


def load() -> C.CDLL:
    """dlopen the in-tree library and check that every declared symbol is exported."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # The library must bind to the SAME HIP runtime instance that owns the caller's device pointers and streams.
    # PyTorch-ROCm bundles its own libamdhip64; importing torch first makes the dynamic loader resolve our
    # DT_NEEDED libamdhip64.so.N to that already-loaded copy (two runtimes in one process => hipErrorNoDevice).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise TcarError("libtcar_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
                        "there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise TcarError("libtcar_hip.so lacks symbols: %s" % missing)
    if lib.tcar_abi_version() != ABI_VERSION:
        raise TcarError("libtcar_hip.so has ABI %d, these bindings expect %d: rebuild (python -c 'import __graft_entry__ "
                        "as g; g.build()')" % (lib.tcar_abi_version(), ABI_VERSION))
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    P = C.POINTER
    lib.tcar_gather_clip_fwd.argtypes = [P(Dims), P(Tables), P(Batch), vp, vp, vp, vp, vp]
    lib.tcar_query_mlp.argtypes = [P(Dims), i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tcar_query_mlp_bwd.argtypes = [P(Dims), i32, vp, vp, vp, vp, vp, vp, vp]
    lib.tcar_flag_fork_selftest.argtypes = [vp, vp, vp, vp]
    lib.tcar_gather_clip_bwd.argtypes = [P(Dims), P(Tables), P(Batch), vp, vp, vp, vp, P(Grads), vp]
    lib.tcar_scatter_add_rows.argtypes = [P(Dims), vp, vp, i64, vp, vp]
    lib.tcar_cand_time_fwd.argtypes = [P(Dims), P(vp * 5), vp, vp, vp]
    lib.tcar_cand_time_bwd.argtypes = [P(Dims), P(vp * 5), vp, vp, P(Grads), vp]
    lib.tcar_cand_time_bwd_indexed.argtypes = [P(Dims), P(vp * 5), vp, vp, vp, i32, vp, P(Grads), vp]
    lib.tcar_cand_time_ws_floats.argtypes = [P(Dims)]
    lib.tcar_gemm_f32.argtypes = [i32, i32, i32, i32, vp, i64, vp, i64, vp, i64, vp, i32, i32, i32, vp]
    lib.tcar_gemm_f32_grouped.argtypes = [i32, i32, P(GemmDesc), vp]
    lib.tcar_gemm_x3_grouped.argtypes = [i32, i32, P(GemmDesc), vp]
    lib.tcar_gemm_bf16.argtypes = [i32, i32, i32, i32, vp, vp, i64, i64, vp, vp, i64, i64, vp, i64, vp, i64, i32, i32, i32,
                                   vp]
    lib.tcar_gemm_bf16_perm.argtypes = [i32, i32, i32, i32, vp, vp, i64, i64, vp, vp, i64, i64, vp, i64, vp, i64, i32, vp, i32,
                                        i32, i32, vp]
    lib.tcar_gemm_bf16_variant.argtypes = [i32, i32, i32, i32, i32, i32, C.c_char_p, i32]
    lib.tcar_split_bf16.argtypes = [vp, i64, i32, i32, vp, vp, i64, vp, vp, i64, i32, i32, vp]
    lib.tcar_splitk_reduce.argtypes = [vp, i32, i32, i32, i64, vp, vp]
    lib.tcar_gemm_splitk_effective.argtypes = [i32, i32]
    lib.tcar_attn_pool_fwd.argtypes = [P(Dims), i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tcar_attn_pool_bwd.argtypes = [P(Dims), i32, i32] + [vp] * 17
    lib.tcar_attn_pool_bwd_q.argtypes = [P(Dims), i32, i32] + [vp] * 18
    lib.tcar_attn_pool_bwd_det.argtypes = [P(Dims), i32, i32] + [vp] * 16
    lib.tcar_attn_pool_fwd_slabs.argtypes = [P(Dims), i32, i32, vp, vp, vp, i32, vp, i32, i64] + [vp] * 8
    lib.tcar_attn_pool_bwd_slabs.argtypes = [P(Dims), i32, i32] + [vp] * 9 + [i32, i32, i64] + [vp] * 7
    lib.tcar_colsum_det.argtypes = [i32, vp, vp]
    lib.tcar_graph_probe.argtypes = [P(Ctx), P(Batch), f32, i32, vp, vp]
    lib.tcar_small_tables_bwd_det.argtypes = [P(Dims), P(Tables), P(Batch), vp, vp, vp, vp, P(Grads), vp, vp]
    lib.tcar_softmax_ce.argtypes = [i32, i32, vp, i64, vp, vp, vp]
    lib.tcar_neg_term.argtypes = [P(Dims), i32, i32, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp]
    lib.tcar_neg_fwd.argtypes = [P(Dims), i32, i32, vp, vp, vp, f32, vp, vp, vp, vp]
    lib.tcar_neg_scatter.argtypes = [P(Dims), i32, i32, vp, vp, vp, vp, vp, vp, f32, vp, vp]
    lib.tcar_splitk_reduce_dact.argtypes = [vp, i32, i32, i32, i64, vp, i64, i32, vp, i64, i32, vp, vp, i32, vp, vp]
    lib.tcar_dact_colsum.argtypes = [i32, i32, i64, vp, vp, vp, i32, vp]
    lib.tcar_rank_topk.argtypes = [i32, i32, vp, i64, vp, i32, vp, vp, vp]
    lib.tcar_eval_rows.argtypes = [i32, i32, vp, i64, vp, i32, vp, vp, vp, vp]
    lib.tcar_eval_diversity.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tcar_sqnorm.argtypes = [vp, P(Segments), vp, vp]
    lib.tcar_clip_adam.argtypes = [vp, vp, vp, vp, P(Segments), vp, vp, vp, f32, f32, f32, f32, f32, vp]
    lib.tcar_clip_adam_all.argtypes = [vp, vp, vp, vp, P(Segments), vp, i64, vp, vp, vp, i64, i32, i32, vp, vp, vp, f32, f32,
                                       f32, f32, f32, vp, vp, i64, vp]
    lib.tcar_clip_adam_early.argtypes = [vp, vp, vp, vp, P(Segments), vp, i64, vp, vp, vp, i64, i32, i32, vp, vp, vp, f32, f32,
                                         f32, f32, f32, vp, vp, i64, vp, i64, vp, vp]
    lib.tcar_clip_adam_rest.argtypes = [vp, i64, vp, vp, vp, i64, i32, i32, vp, vp, vp, f32, f32, f32, f32, f32, vp, vp, i64,
                                        vp, vp]
    lib.tcar_gemm_bf16_ce.argtypes = [i32, i32, i32, vp, vp, i64, i64, vp, vp, i64, i64, i32, vp, vp, vp, i64, vp, i64, i64, vp, i64, vp,
                                      vp, i32, vp, vp, vp]
    lib.tcar_time_onehot.argtypes = [P(Dims), vp, vp, i64, vp]
    lib.tcar_time_scores.argtypes = [P(Dims), P(vp * 5), i32, vp, i64, vp, vp, i64, vp]
    lib.tcar_time_scores_clip.argtypes = [P(Dims), P(vp * 5), i32, vp, i64, vp, vp, i64, vp, vp]
    lib.tcar_attout_finish_scores.argtypes = [P(Dims), P(vp * 5), i32, vp, i32, i32, i64, vp, vp, vp, i64, vp, vp, i64, vp, vp, i64,
                                              vp, vp, i64, vp, vp]
    lib.tcar_gemm_bf16_dx_onehot.argtypes = [i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i64, vp, i64, i32, vp]
    lib.tcar_reduce_dact_onehot.argtypes = [vp, i32, i32, i32, i64, vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, vp]
    lib.tcar_gemm_bf16_de_qz.argtypes = [i32, i32, vp, i64, i64, vp, i64, i64, i32, vp, i64, vp, vp, vp, vp, i32, vp]
    lib.tcar_cand_time_bwd_onehot.argtypes = [P(Dims), i32, vp, vp, vp, vp, i64, vp, vp, P(Grads), vp]
    lib.tcar_ce_finish.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i64, vp]
    lib.tcar_ce_anchor_fold.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, i32, i64, vp]
    lib.tcar_gemm_bf16_ce_anchor.argtypes = [i32, i32, i32, vp, vp, i64, i64, vp, vp, i64, i64, i32, vp, vp, vp, i64, vp, i64, i64, vp, i64,
                                             vp, vp, i32, vp, vp, vp]
    lib.tcar_reduce_dact_onehot_scaled.argtypes = [vp, i32, i32, i32, i64, vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, vp, i64, vp, i32, vp]
    lib.tcar_ce_shard_stats.argtypes = [i32, i32, vp, vp, vp, i32, i32, vp, vp]
    lib.tcar_ce_rescale.argtypes = [i32, i32, i32, i32, vp, vp, vp, i32, i32, vp, i64, vp]
    lib.tcar_layernorm_fwd.argtypes = [i64, i32, vp, vp, vp, f32, vp, vp, vp]
    lib.tcar_layernorm_bwd.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tcar_mha_core_fwd.argtypes = [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tcar_mha_core_bwd.argtypes = [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tcar_clip_adam_2d.argtypes = [vp, i64, vp, vp, vp, i64, i32, i32, vp, vp, vp, f32, f32, f32, f32, f32, vp]
    lib.tcar_clip_adam_2d_bf16.argtypes = [vp, i64, vp, vp, vp, i64, i32, i32, vp, vp, vp, f32, f32, f32, f32, f32, vp, vp,
                                           i64, vp]
    lib.tcar_cand_time_fwd_bf16.argtypes = [P(Dims), P(vp * 5), vp, vp, vp, vp, vp]
    lib.tcar_softmax_ce_bf16.argtypes = [i32, i32, vp, i64, vp, vp, vp, vp, vp]
    lib.tcar_step_forward.argtypes = [P(Ctx), P(Batch), i32, vp]
    lib.tcar_step_backward_local.argtypes = [P(Ctx), P(Batch), vp]
    lib.tcar_step_finish.argtypes = [P(Ctx), P(Batch), vp]
    lib.tcar_step_update.argtypes = [P(Ctx), f32, vp]
    lib.tcar_train_step.argtypes = [P(Ctx), P(Batch), i32, f32, vp]
    lib.tcar_train_step_deferred.argtypes = [P(Ctx), P(Batch), i32, i32, f32, vp]
    lib.tcar_eval_step.argtypes = [P(Ctx), P(Batch), i32, i32, vp]
    lib.tcar_step_form.argtypes = [P(Ctx), P(Batch), vp]
    lib.tcar_shard_form.argtypes = [P(Ctx), P(Shard), vp]
    for s in SYMBOLS:
        getattr(lib, s).restype = C.c_int
    lib.tcar_segsum_ws_bytes.restype = C.c_int64
    lib.tcar_segsum_ws_bytes.argtypes = [P(Dims), i64]
    lib.tcar_segsum_rows_buffer.restype = C.c_void_p
    lib.tcar_segsum_rows_buffer.argtypes = [P(Dims), P(Batch), vp]
    lib.tcar_segsum_index.argtypes = [P(Dims), P(Batch), vp, i64, vp]
    lib.tcar_segsum_norms_buffer.restype = C.c_void_p
    lib.tcar_segsum_norms_buffer.argtypes = [P(Dims), P(Batch), vp]
    lib.tcar_segsum_apply.argtypes = [P(Dims), P(Batch), vp, i64, i32, vp, vp, vp, i64, vp, vp, vp, vp, vp, f32, vp, vp]
    lib.tcar_sqnorm_det.argtypes = [vp, i64, vp, i64, vp]
    lib.tcar_tuning_defaults.argtypes = [vp]
    lib.tcar_tuning_set.argtypes = [vp, C.c_char_p, i32]
    lib.tcar_fork_state_bytes.restype = C.c_int64
    lib.tcar_fork_state_bytes.argtypes = []
    lib.tcar_ctx_bytes.restype = C.c_int64
    lib.tcar_ctx_bytes.argtypes = []
    if lib.tcar_ctx_bytes() != C.sizeof(Ctx):
        raise RuntimeError("tcar_ctx_t is %d bytes in libtcar_hip.so and %d in the ctypes mirror (_lib.Ctx): the mirror is stale"
                           % (lib.tcar_ctx_bytes(), C.sizeof(Ctx)))
    lib.tcar_flag_poll_expire.argtypes = [vp, vp, vp]
    lib.tcar_gather_clip_fwd_tuned.argtypes = [vp] + lib.tcar_gather_clip_fwd.argtypes
    lib.tcar_gemm_bf16_tuned.argtypes = [vp] + lib.tcar_gemm_bf16.argtypes
    lib.tcar_gemm_bf16_dx_onehot_tuned.argtypes = [vp] + lib.tcar_gemm_bf16_dx_onehot.argtypes
    lib.tcar_mha_core_fwd_tuned.argtypes = [vp] + lib.tcar_mha_core_fwd.argtypes
    lib.tcar_mha_core_bwd_tuned.argtypes = [vp] + lib.tcar_mha_core_bwd.argtypes
    lib.tcar_step_session_forward.argtypes = [P(Ctx), P(Batch), vp]
    lib.tcar_shard_score.argtypes = [P(Ctx), P(Shard), i32, vp]
    lib.tcar_shard_backward.argtypes = [P(Ctx), P(Shard), vp, vp]
    lib.tcar_shard_finish.argtypes = [P(Ctx), P(Shard), i32, vp, vp, vp]
    lib.tcar_step_session_backward.argtypes = [P(Ctx), P(Batch), vp, vp, i64, i64, vp, vp]
    lib.tcar_step_dense_norms.argtypes = [P(Ctx), vp]
    lib.tcar_shard_begin.argtypes = [P(Ctx), P(Batch), i32, i32, vp, i64, i32, i32, f32, vp]
    lib.tcar_shard_join.argtypes = [P(Ctx), vp]
    lib.tcar_shard_step_local.argtypes = [P(Ctx), P(Ctx), P(Shard), P(Batch), i32, vp, i64, i32, f32, vp, i64, i64, P(Dims), f32, vp]
    lib.tcar_shard_pack_head.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i64, vp]
    lib.tcar_shard_unpack_head.argtypes = [i32, i32, i32, vp, i64, vp, vp, vp, vp]
    lib.tcar_shard_pack_ids.argtypes = [i64, i64, i32, vp, vp, i64, i32, vp, vp, f32, vp, vp]
    lib.tcar_scatter_add_rows_packed.argtypes = [P(Dims), vp, i64, i64, i32, vp, vp]
    lib.tcar_softmax_stats.argtypes = [i32, i32, vp, i64, vp, i32, vp, vp]
    lib.tcar_softmax_combine.argtypes = [i32, i32, vp, vp, vp, vp, vp]
    lib.tcar_softmax_combine_rowstat.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp]
    lib.tcar_softmax_grad.argtypes = [i32, i32, vp, i64, vp, vp, i32, vp, vp, vp]
    lib.tcar_neg_scatter_range.argtypes = [P(Dims), i64, i32, i32, i32, vp, vp, i64, vp, vp, vp]
    lib.tcar_form_batch.argtypes = [P(Dims), P(Store), P(NegSrc), vp, i32, i32, i32, i32, C.c_uint64, C.c_uint64, vp, vp]
    lib.tcar_build_id.restype = C.c_char_p
    lib.tcar_build_id.argtypes = []
    # a binary built from other sources than the ones next to it is stale (the build is digest-gated, not mtime-gated)
    if have_sources():
        got, want = lib.tcar_build_id().decode(), source_build_id()
        if got != want:
            raise TcarError("libtcar_hip.so was built from other sources (build id %s, sources %s): rebuild with "
                            "python -c 'import __graft_entry__ as g; g.build()'" % (got, want))
    _LIB = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise TcarError("%s failed with code %d (%s)" % (what, rc, {-1: "bad argument", -2: "launch error"}.get(rc, "?")))
