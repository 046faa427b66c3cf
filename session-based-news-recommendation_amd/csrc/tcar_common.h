// Shared device helpers for the TCAR gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tcar_hip.h"

#define TCAR_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) {                                  \
      fprintf(stderr, "tcar: HIP error %d (%s) at %s:%d\n", (int)e__, hipGetErrorString(e__), __FILE__, __LINE__); \
      return TCAR_E_LAUNCH;                                   \
    }                                                         \
  } while (0)

// hipGetLastError() is sticky per thread: clear whatever an earlier (foreign) runtime call left behind, so the
// check after the launch reports THIS launch only.
#define TCAR_LAUNCH(...)                 \
  do {                                   \
    (void)hipGetLastError();             \
    hipLaunchKernelGGL(__VA_ARGS__);     \
  } while (0)

static inline bool tcar_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// One-time, per-DEVICE kernel attributes (hipFuncSetAttribute for > 64 KB of dynamic LDS).  One bit per device ordinal:
// correct with several devices in one process and safe from several host threads (two threads may both set the
// attribute once: the call is idempotent).  This is the only mutable state the library keeps.
#include <atomic>
struct TcarOnce {
  std::atomic<unsigned long long> mask{0};
};
static inline bool tcar_first_on_device(TcarOnce& o) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (o.mask.load(std::memory_order_acquire) & bit) return false;
  o.mask.fetch_or(bit, std::memory_order_acq_rel);
  return true;
}
#define TCAR_SET_LDS_ONCE(kernel, bytes)                                                                     \
  do {                                                                                                        \
    static TcarOnce once__;                                                                                   \
    if (tcar_first_on_device(once__))                                                                         \
      (void)hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
  } while (0)

// Diagnostic tuning switches: tcar_tuning_t of include/tcar_hip.h.  tcar_tuning() is the PROCESS snapshot — shipped defaults
// overridden by the TCAR_* environment, read ONCE at first use (C++11 thread-safe static) and never written again; a context
// (tcar_ctx_t.tune) or a *_tuned entry point may carry its own copy instead.  Defined in step.hip.
typedef tcar_tuning_t TcarTuning;
// Choices that were TCAR_* switches until round 6 — each A/B'd at least twice (profiles/experiments_log.md) — are constants of the build
// now; tcar_tuning_t keeps the twelve switches that select between forms a test pins or a user may need.
namespace tcar_fixed {
constexpr int rest_grid = 512;     // grid cap of the deferred Adam rest pass (it runs beside the next step's session forward)
constexpr int fork_delay = 7;      // us the aux prologue's flag fork holds its consumer back behind the end of the logits GEMM
#ifndef TCAR_FIX_X3_ONESHOT
#define TCAR_FIX_X3_ONESHOT 4      // (diagnostic builds override it: tools/micro/build_x3ring.sh)
#endif
constexpr int x3_oneshot = TCAR_FIX_X3_ONESHOT;      // small-GEMM launches of at most this many 64-deep stages per workgroup keep two stages in flight
#ifndef TCAR_FIX_BWD_SMALL_HI
#define TCAR_FIX_BWD_SMALL_HI 0    // (diagnostic builds: tools/micro/build_x3ring.sh bwdhi)
#endif
constexpr int bwd_small_hi = TCAR_FIX_BWD_SMALL_HI;  // 1: with the hi-only backward precision (scoring_bwd == 1) the session-side BACKWARD GEMMs
                                                     // (dpooled, input gradients, weight gradients) contract plain bf16 operands too
#ifndef TCAR_FIX_X3_DEEP
#define TCAR_FIX_X3_DEEP 0         // (diagnostic builds: tools/micro/build_x3ring.sh deep)
#endif
constexpr int x3_deep = TCAR_FIX_X3_DEEP;            // 1: long-K small-GEMM launches of at most 256 workgroups take 128-deep stages
constexpr int gather_wg = 2;       // 1024-thread workgroups per CU of the gather's throughput form (2 x 78 KB of LDS fit)
}  // namespace tcar_fixed
const TcarTuning& tcar_tuning();

// Completion flag of a kernel (step.hip: fork_arm / fork_go).  A kernel that carries one publishes `epoch` to *flag when its
// LAST workgroup is through: the side stream's consumers sit behind a one-wave polling kernel instead of behind an event the main
// stream would have to record (measured, tools/micro/event_cost: a record between two kernels costs the recording stream 6.5 us,
// the flag costs it nothing and releases the consumer 0.4 us after the producer's end).  cnt == nullptr: no flag.
struct TcarSignal { unsigned* cnt; unsigned* flag; unsigned epoch; unsigned slot; };
// In-kernel form of the consumer side of a flag fork: instead of a polling kernel in front of it, the consuming kernel waits itself,
// at the point where it first needs the producer's bytes (everything before that point overlaps the wait).  EVERY lane of a wave
// calls tcar_wave_wait: lane 0 polls the flag word (relaxed agent-scope loads + s_sleep, bounded like poll_flag_kernel and counting
// a time-out in the same error words).  NO acquire fence follows (a buffer_inv sc1 per wave, in hundreds of workgroups, measured
// +17 us on the slab reduce and slowed every kernel beside it): the caller reads the producer's bytes — a few KB at most, stored
// write-through by a flag-capable kernel — with sc1 loads (ld4_sc1 / ld1_sc1: served by L2 / memory, never by this CU's L1), the
// form cdna_hip_programming.md Guideline 16 lists beside the acquire.  flag == nullptr: no wait.
struct TcarWait { const unsigned* flag; unsigned epoch; unsigned* err; unsigned* err_host; };
__device__ __forceinline__ void tcar_wave_wait(const TcarWait& w) {
  if (!w.flag) return;
  if ((threadIdx.x & 63) == 0) {
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(w.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - w.epoch) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 100000000LL) {            // 1 s of the 100-MHz wall clock
        atomicAdd(w.err, 1u);
        if (w.err_host) __hip_atomic_fetch_add(w.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;       // (the time-out is COUNTED and the engine raises after the step: what this launch computes meanwhile is void)
      }
    }
  }
  // compiler barrier: no load of the producer's bytes may be scheduled above the poll.  (Hardware ordering is the callers' contract:
  // EVERY producer byte is read with ld4_sc1 / ld1_sc1 — L2-served loads issued after this point —, never with a plain load that
  // this CU's L1 could serve stale.)
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ float4 ld4_sc1(const float* p) {
  typedef float f4_t __attribute__((ext_vector_type(4)));
  f4_t v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ld1_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Launch options of the internal (C++) forms of the entry points, `..._o`: the tuning copy to consult (NULL = process snapshot)
// and the completion flag THIS launch is to carry — passed explicitly, there is no per-thread "pending" state.  The launcher
// sets `carried` when the kernel form it chose publishes the flag (flag-capable forms: the latency form of the forward gather,
// the click-query MLP, small-GEMM launches WITHOUT bf16 plane outputs, the softmax-epilogue logits GEMM, sqnorm_seg, the
// small-table norm fold); a launch that cannot carry it leaves `carried` false and the driver forks with an event.
// Anchored softmax form, dX side (score.hip: ce_anchor_fold_kernel / reduce_dact_onehot_kernel).  The slabs hold the UNSCALED plane
// times [E | OH]; scale2[m] = (1 / S_m, r_m): the slab sums (and dP) of row m are multiplied by 1 / S_m, and r_m — the part of the
// one-hot's -1 that the bf16 label entry of the plane could not hold, already divided by S_m — comes back in fp32 as
// r_m * [E[label[m], :] | onehot(publish time of label[m])].  E: fp32 candidate rows [n_items, ldE] (item | content columns first).
constexpr int TCAR_ANCHOR_COLS = 8;      // columns 139 .. 146 of the one-hot K segment (embed.hip: time_onehot_kernel)
// lab_off: the label's row in E / mwdhm is label[m] - lab_off (catalog shard: E, mwdhm and n_items are the shard's; a label outside
// [0, n_items) has r_m = 0 and is only clamped for the address)
struct TcarRowFix {
  const float* scale2; const int32_t* label; const float* E; long ldE; const int32_t* mwdhm; int n_items; int lab_off = 0;
};
struct TcarOpt {
  const tcar_tuning_t* tune = nullptr;
  TcarSignal sig{};
  bool carried = false;
  TcarWait wait{};       // a flag the launch waits for IN the kernel (launchers that support it: the one-hot slab reduce)
  // START flag (round 6): the kernel's first workgroup publishes `start.epoch` to *start.flag as its first action.  A kernel that has
  // started has every predecessor of its stream behind it, COMPLETE and released — so a side stream that polls this word is ordered
  // behind the PREDECESSOR of this launch without an event record on this stream (6.5 us between two kernels) and without write-through
  // stores in the predecessor.  Launchers that support it (the one-hot dX GEMM) set `started`.
  TcarSignal start{};
  bool started = false;
  // label window of the softmax-epilogue logits GEMM (catalog-sharded step): the column of row m's label is label[m] - lab_off,
  // and a label outside [0, N) is in another shard — no label score is written for it (lab_window = 0: labels are clamped)
  int lab_off = 0, lab_window = 0;
  // ANCHORED softmax epilogue of the logits GEMM (round 6; step.hip: ce_anchored): the accumulators already are x - anchor[m] — the
  // anchor rides in the one-hot K segment (eight spare columns of the time-score planes hold minus its partial sums, the one-hot
  // plane has ones there: embed.hip) — so the plane is exp(accumulator) with NO group maximum, the statistics are (0, group sum) and
  // lab_logit is the label's accumulator (its score minus the anchor).  Every group of a row then shares ONE scale, the plane never
  // needs the rescale pass: dX is scaled per row in its slab reduce (rowfix), dE contracts the plane with per-row scaled attout
  // planes (tcar_ce_anchor_fold_o)
  bool anchored = false;
  const TcarRowFix* rowfix = nullptr;      // one-hot slab reduce of the anchored form (below)
  bool hi_only = false;                    // small-GEMM launches (tcar_gemm_x3_grouped_o): plain bf16 operands, ONE MFMA per product
  // zeroed device words a launch may use for an order-fixed last-arrival fold (tcar_sqnorm_o: word 0 = arrival counter, kept zero
  // between launches; then one float per 32,768-float chunk): tcar_ctx_t.fold_scratch
  unsigned* scratch = nullptr;
  int scratch_words = 0;
  const TcarTuning& tn() const { return tune ? *tune : tcar_tuning(); }
};
inline const TcarTuning& tcar_tn(const TcarOpt* o) { return o ? o->tn() : tcar_tuning(); }
inline TcarSignal tcar_sig(TcarOpt* o) {       // the flag a flag-capable launch carries (marks it carried)
  if (!o || !o->sig.cnt) return TcarSignal{};
  o->carried = true;
  return o->sig;
}
// ---- internal (C++ linkage) forms of entry points that consult a switch or can carry a completion flag; the extern "C" names of
// include/tcar_hip.h call them with the process snapshot and no flag
int tcar_ce_finish_o(int B, int N, int group_width, int ngroups, const float* stats, const float* lab_logit, const int32_t* label,
                     float* rowstat, float* ce, void* dl_hi, int64_t inner, void* stream, TcarOpt* o);
int tcar_ce_anchor_fold_o(int B, int N, int group_width, int ngroups, const float* stats, const float* lab_logit, const int32_t* label,
                          float* rowstat, float* ce, float* scale2, void* dl_hi, int64_t inner, const void* ap_hi, const void* ap_lo,
                          void* aps_hi, int ap_cols, int64_t ap_inner, void* stream, TcarOpt* o);
// catalog-sharded step, anchored form: tcar_ce_anchor_fold's second half with the row sums from the statistics exchange (rowstat =
// (lse, 1) of tcar_softmax_combine_rowstat, every group reference 0) and a label WINDOW; tcar_anchor_scores: P[b, 139] = -attout[b] .
// E[label[b]] over the item | content columns (hi / lo), zero for padding sessions (label < 0)
int tcar_ce_anchor_apply_o(int B, int N, const float* rowstat, const int32_t* label, int lab_off, void* dl_hi, int64_t inner,
                           const void* ap_hi, const void* ap_lo, void* aps_hi, int ap_cols, int64_t ap_inner, float* scale2, void* stream);
int tcar_ce_shard_stats_a(int B, int ngroups, const float* stats, const float* lab_logit, const int32_t* label, int n0, int n_loc,
                          float* out3, int anchored, void* stream);
int tcar_softmax_combine_anchored(int W, int B, const float* stats_all, const int32_t* label, float* lse, float* ce, float* rowstat,
                                  void* stream);
int tcar_anchor_scores(int ldh, int B, const float* attout, int64_t ld_att, const int32_t* label, const float* E, int64_t ldE,
                       int64_t n_rows, void* p_hi, void* p_lo, int64_t inner, void* stream);
int tcar_clip_adam_early_2(float* w, const float* g, float* m, float* v, const tcar_segments_t* segs, float* w2d, int64_t ldw,
                           const float* g2d, float* m2d, float* v2d, int64_t rows, int32_t cols, int32_t slot, const float* sqn_dense,
                           const float* sqn_pieces, const int32_t* use_dense, float clip, float lr_t, float b1, float b2, float eps,
                           void* e16_hi, void* e16_lo, int64_t ld16, const int32_t* ids, int64_t n_ids, const int32_t* ids2,
                           int64_t n_ids2, uint32_t* bitmap, void* stream);
int tcar_attout_finish_scores_a(const tcar_dims_t* d, const float* const time_tab[5], int B, const float* slabs, int nd_ic, int nd_pt,
                                int64_t stride, const float* bias_o, const float* bias_ot, float* attout, int64_t ld_out, void* a_hi,
                                void* a_lo, int64_t a_inner, void* ap_hi, void* ap_lo, int64_t ap_inner, void* p_hi, void* p_lo,
                                int64_t p_inner, float* tclip, const int32_t* label, const float* E, int64_t ldE, void* stream);
int tcar_gather_clip_fwd_o(const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt, float* x_icp, float* x_pt,
                           float* x_act, float* click_t, void* stream, TcarOpt* o);
int tcar_query_mlp_o(const tcar_dims_t* d, int B, const float* click_t, const float* q1_w, const float* q1_b, const float* q2_w,
                     const float* q2_b, float* q1, float* q, void* stream, TcarOpt* o);
int tcar_attn_pool_bwd_slabs_o(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt, const float* pre1,
                               const float* pre2, const float* q, const float* w_res1, const float* w_res2, const float* alpha,
                               const float* dpooled, int nd_ic, int nd_pt, int64_t dp_stride, float* dx_icp, float* dx_pt, float* dq,
                               float* dpre1, float* dpre2, float* gw_rows, void* stream, TcarOpt* o);
int tcar_query_mlp_bwd_o(const tcar_dims_t* d, int B, const float* dq, const float* q1, const float* q1_w, const float* q2_w, float* dq1,
                         float* dclick, void* stream, TcarOpt* o);
int tcar_small_tables_bwd_det_o(const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt, const float* dx_icp,
                                const float* dx_pt, const float* dx_act, const float* dclick, const tcar_grads_t* g, float* ws,
                                void* stream, TcarOpt* o, const float* cand_pc /* optional [139]: candidate-side norm pieces */,
                                int64_t ws_floats /* of ws: below tcar_small_det_ws_floats() the single-pass form only */);
int tcar_cand_time_bwd_onehot_w(const tcar_dims_t* d, int B, const int32_t* inv_off, const float* qz, const float* dP,
                                const float* attout, int64_t ld_att, const float* tclip, float* ws, const tcar_grads_t* g, void* stream,
                                const TcarWait& wait_dp, int with_pieces);
inline const float* tcar_cand_pieces(const tcar_dims_t* d, const float* ws) { return ws + (long)139 * 32 * (d->ldt + 4); }
int tcar_sqnorm_o(const float* g, const tcar_segments_t* segs, float* sqn_dense, void* stream, TcarOpt* o);
int tcar_colsum_sqnorm_o(const float* g, const tcar_segments_t* segs, int ncs, const tcar_colsum_t* cs, float* sqn_dense, void* stream,
                         TcarOpt* o);
int tcar_attn_pool_fwd_slabs_w(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt, const float* pre1_slabs,
                               int n1, const float* pre2_slabs, int n2, int64_t slab_stride, float* pre1, float* pre2, const float* q,
                               const float* w_res1, const float* w_res2, float* pooled, float* alpha, void* stream,
                               const TcarWait& wait_q);
int tcar_gemm_x3_grouped_o(int layout, int nprob, const tcar_gemm_desc_t* descs, void* stream, TcarOpt* o);
int tcar_gemm_bf16_perm_o(int layout, int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner, int64_t a_rows,
                          const void* B_hi, const void* B_lo, int64_t b_inner, int64_t b_rows, float* C, int64_t ldc, float* C2,
                          int64_t ldc2, int csplit, const int32_t* c2_perm, int c2_group, int nsplit, int splitk, void* stream,
                          TcarOpt* o);
int tcar_gemm_bf16_ce_o(int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner, int64_t a_rows, const void* B_hi,
                        const void* B_lo, int64_t b_inner, int64_t b_rows, int K1, const void* A2_hi, const void* A2_lo,
                        const void* B2_hi, int64_t inner2, void* p_hi, int64_t p_inner, int64_t p_rows, float* stats,
                        int64_t stats_floats, const int32_t* label, float* lab_logit, int nsplit, int32_t* group_width,
                        int32_t* ngroups, void* stream, TcarOpt* o);
int tcar_gemm_bf16_dx_onehot_o(int M, int N1, int K, const void* A_hi, int64_t a_inner, int64_t a_rows, const void* B_hi,
                               int64_t b_inner, int64_t b_rows, const void* B2_hi, int64_t inner2, float* C, int64_t ldc, int splitk,
                               void* stream, TcarOpt* o);
int tcar_gemm_bf16_de_qz_o(int M, int K, const void* A_hi, int64_t a_inner, int64_t a_rows, const void* B_hi, int64_t b_inner,
                           int64_t b_rows, int ldh, float* C, int64_t ldc, const int32_t* mwdhm, const int32_t* perm,
                           const float* tclip, float* qz, int tile, void* stream, TcarOpt* o);
int tcar_reduce_dact_onehot_o(const float* slabs, int splitk, int M, int ic, int64_t ld, const float* addend, int64_t ld_add,
                              const float* y, int64_t ldy, const float* tclip, float* out, int64_t ldo, float* dP, float* bias_grad0,
                              float* bias_grad1, void* stream, TcarOpt* o);
int tcar_neg_fwd_o(const tcar_dims_t* d, int B, int K, const float* E, const int32_t* neg, const float* attout, float weight,
                   float* neg_fb, float* coef, float* negpart, void* stream, TcarOpt* o);
int tcar_clip_adam_rest_keep_o(float* w2d, int64_t ldw, const float* g2d, float* m2d, float* v2d, int64_t rows, int32_t cols,
                               int32_t slot, const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense, float clip,
                               float lr_t, float b1, float b2, float eps, void* e16_hi, void* e16_lo, int64_t ld16,
                               uint32_t* bitmap, void* stream, int rest_grid);
int tcar_softmax_ce_bf16_o(int B, int N, float* logits, int64_t ld, const int32_t* label, float* ce, void* dl_hi, void* dl_lo,
                           void* stream);

// 16-byte write-through store (sc1): the bytes bypass the write-back state of this XCD's L2, so a consumer behind a completion
// flag needs no release fence / L2 write-back from the producer (cdna_hip_programming.md Guideline 16, R1).  The compiler does not
// count this store: drain with an explicit s_waitcnt vmcnt(0) before signalling (tcar_signal_done does).
__device__ __forceinline__ void st4_sc1(float* p, float4 v) {
  typedef float f4_t __attribute__((ext_vector_type(4)));
  const f4_t x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
// Every thread of every workgroup calls this as the kernel's last statement (no early returns ahead of it).  The workgroup's
// stores are drained into its XCD's L2 (vmcnt) before it is counted; NO release fence here — an agent-scope release in every
// workgroup is an L2 write-back per workgroup (measured: the CE-rescale kernel 24 -> 176 us, and every kernel beside it slower).
// The write-back of the eight L2s is done ONCE per XCD by the polling kernel of the consumer stream (step.hip poll_flag_kernel),
// after the flag and before the consumer's first kernel starts.
__device__ __forceinline__ void tcar_signal_done(const TcarSignal& s) {
  if (!s.cnt) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    if (__hip_atomic_fetch_add(s.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y * gridDim.z - 1u) {
      __hip_atomic_store(s.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // reusable by the slot's next launch
      __hip_atomic_store(s.flag, s.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// sum over aligned groups of `width` consecutive lanes (width = 16, 32 or 64)
__device__ __forceinline__ float group_sum(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// explicit fma chain: hipcc contracts a plain a*b + c*d + ... differently from one inlining context to the next, and two
// kernels that must agree bit for bit (the two forms of the forward gather) would round their row norms differently
__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming (non-temporal) forms: data that is touched once per step by an HBM-bound pass should not evict what the
// latency-bound kernels beside it keep in L2 / the Infinity Cache
typedef float tcar_v4f_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* p) {
  const tcar_v4f_nt v = __builtin_nontemporal_load(reinterpret_cast<const tcar_v4f_nt*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4_stream(float* p, float4 v) {
  const tcar_v4f_nt x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<tcar_v4f_nt*>(p));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float4 a, float s, float4 c) {
  return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}
__device__ __forceinline__ void atomic_add4(float* p, float4 v) {
  atomicAdd(p + 0, v.x);
  atomicAdd(p + 1, v.y);
  atomicAdd(p + 2, v.z);
  atomicAdd(p + 3, v.w);
}

// tf.clip_by_norm(row, 1.0): y = x / max(||x||, 1).  scale for a row with sum of squares ss.
__device__ __forceinline__ float clip_scale(float ss) { return ss > 1.0f ? 1.0f / sqrtf(ss) : 1.0f; }
// backward of the row clip: gx = gy/n - x (x.gy)/n^3 when n > 1, identity otherwise.
// returns (a, b) with gx = a*gy - b*x
__device__ __forceinline__ void clip_bwd_coef(float ss, float d, float& a, float& b) {
  if (ss > 1.0f) {
    float inv = 1.0f / sqrtf(ss);
    a = inv;
    b = d * inv * inv * inv;
  } else {
    a = 1.0f;
    b = 0.0f;
  }
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
