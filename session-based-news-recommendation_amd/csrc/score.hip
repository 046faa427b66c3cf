// Scoring-side kernels for gfx950: softmax cross-entropy over the full catalog, the sampled negative-feedback
// term, activation/bias backward, and the evaluation rank / top-k.
//
// Reference: model_combine.py:138-147 (logits, sparse softmax CE, neg_logits / neg_feedback, loss),
// modules.py:52-54 (bias + activation), util.py:8-18 and model_combine.py:301 (rank, top-20).
// All four are HBM/L2-bandwidth bound passes over [B, N] or gathered rows; they use 16-byte accesses with one
// workgroup (or wave) per session and wave-shuffle reductions.
#include <utility>
#include "tcar_common.h"
#include "tcar_bf16_layout.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_s;

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  const float r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_max_256(float v, float* sh) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  const float r = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  return r;
}

// ---- sparse softmax cross entropy + gradient (model_combine.py:145) ------------------------------------
// One workgroup per session.  Pass 1: online (max, sum-exp) over the row.  Pass 2: overwrite the row with
// softmax - onehot (the gradient of the SUM of the per-session losses, model_combine.py:147,156).
// dh != NULL: the gradient is written as bf16 hi / lo planes [B, ld] (operands of gemm_bf16.hip) and the logits are
// left untouched; otherwise it overwrites the logits in fp32.
__global__ __launch_bounds__(256) void softmax_ce_kernel(int B, int N, float* __restrict__ logits, long ld,
                                                         const int32_t* __restrict__ label, float* __restrict__ ce,
                                                         __bf16* __restrict__ dh, __bf16* __restrict__ dl) {
  __shared__ float sh[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= B) {   // padding rows of the KB32 planes (grid covers ceil128(B) rows): zero, they are k-rows of dE
    for (int i = tid; i < (int)(ld >> 2); i += 256) {
      const long o = kb32_off(b, i * 4, (int)(ld >> 5));
      bf16x4_s z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
      *reinterpret_cast<bf16x4_s*>(dh + o) = z;
      if (dl) *reinterpret_cast<bf16x4_s*>(dl + o) = z;
    }
    return;
  }
  float* row = logits + (long)b * ld;
  const int n4 = (N + 3) >> 2;
  float m = -INFINITY, s = 0.f;
  for (int i = tid; i < n4; i += 256) {
    float4 v = ld4(row + i * 4);
    const int c = i * 4;
    if (c + 1 >= N) v.y = -INFINITY;
    if (c + 2 >= N) v.z = -INFINITY;
    if (c + 3 >= N) v.w = -INFINITY;
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    const float mn = fmaxf(m, mx);
    s = s * expf(m - mn) + expf(v.x - mn) + expf(v.y - mn) + expf(v.z - mn) + expf(v.w - mn);
    m = mn;
  }
  const float gm = block_max_256(m, sh);
  s = (m == -INFINITY) ? 0.f : s * expf(m - gm);
  const float gs = block_sum_256(s, sh);
  const float lse = gm + logf(gs);
  const int lab = clampi(label[b], 0, N - 1);
  if (tid == 0) ce[b] = lse - row[lab];
  __syncthreads();                       // the label logit is read before the row is overwritten
  const int l4 = (int)(ld >> 2);
  for (int i = tid; i < l4; i += 256) {
    const int c = i * 4;
    float4 v = ld4(row + c);
    float4 o;
    o.x = (c + 0 < N) ? expf(v.x - lse) - (c + 0 == lab ? 1.f : 0.f) : 0.f;
    o.y = (c + 1 < N) ? expf(v.y - lse) - (c + 1 == lab ? 1.f : 0.f) : 0.f;
    o.z = (c + 2 < N) ? expf(v.z - lse) - (c + 2 == lab ? 1.f : 0.f) : 0.f;
    o.w = (c + 3 < N) ? expf(v.w - lse) - (c + 3 == lab ? 1.f : 0.f) : 0.f;
    if (dh) {
      const float ov[4] = {o.x, o.y, o.z, o.w};
      bf16x4_s h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) { h[j] = (__bf16)ov[j]; l[j] = (__bf16)(ov[j] - (float)h[j]); }
      const long o = kb32_off(b, c, (int)(ld >> 5));            // KB32 blocked planes [ceil128(B), ld]
      *reinterpret_cast<bf16x4_s*>(dh + o) = h;
      if (dl) *reinterpret_cast<bf16x4_s*>(dl + o) = l;
    } else {
      st4(row + c, o);
    }
  }
}

// Row-resident variant: the whole logits row lives in registers between the two passes (NT threads x R float4, all
// loads issued up front), so the row is read from memory ONCE: 1 read + 1 write of [B, N] instead of 2 reads + 1 write,
// and one exp per element (the pass-1 exponentials are kept and rescaled by 1 / sum).
template <int NT, int R>
__global__ __launch_bounds__(NT) void softmax_ce_rows_kernel(int B, int N, float* __restrict__ logits, long ld,
                                                              const int32_t* __restrict__ label, float* __restrict__ ce,
                                                              __bf16* __restrict__ dh, __bf16* __restrict__ dl) {
  __shared__ float sh[NT / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (b >= B) {   // padding rows of the KB32 planes
    for (int i = tid; i < (int)(ld >> 2); i += NT) {
      const long o = kb32_off(b, i * 4, (int)(ld >> 5));
      bf16x4_s z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
      *reinterpret_cast<bf16x4_s*>(dh + o) = z;
      if (dl) *reinterpret_cast<bf16x4_s*>(dl + o) = z;
    }
    return;
  }
  float* row = logits + (long)b * ld;
  float4 v[R];
  const float ninf = -INFINITY;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int c = (tid + r * NT) * 4;
    v[r] = (c < (int)ld) ? ld4(row + c) : make_float4(ninf, ninf, ninf, ninf);
  }
  float m = ninf;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int c = (tid + r * NT) * 4;
    if (c + 0 >= N) v[r].x = ninf;
    if (c + 1 >= N) v[r].y = ninf;
    if (c + 2 >= N) v[r].z = ninf;
    if (c + 3 >= N) v[r].w = ninf;
    m = fmaxf(m, fmaxf(fmaxf(v[r].x, v[r].y), fmaxf(v[r].z, v[r].w)));
  }
  m = wave_max(m);
  if (lane == 0) sh[w] = m;
  __syncthreads();
  float gm = sh[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) gm = fmaxf(gm, sh[i]);
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    v[r].x = expf(v[r].x - gm); v[r].y = expf(v[r].y - gm); v[r].z = expf(v[r].z - gm); v[r].w = expf(v[r].w - gm);
    s += (v[r].x + v[r].y) + (v[r].z + v[r].w);
  }
  s = wave_sum(s);
  if (lane == 0) sh[w] = s;
  __syncthreads();
  float gs = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) gs += sh[i];
  const float lse = gm + logf(gs), inv = 1.0f / gs;
  const int lab = clampi(label[b], 0, N - 1);
  if (tid == 0) ce[b] = lse - row[lab];
  __syncthreads();                       // the label logit is read before the row is overwritten (fp32 mode)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int c = (tid + r * NT) * 4;
    if (c >= (int)ld) continue;
    float ov[4] = {v[r].x * inv, v[r].y * inv, v[r].z * inv, v[r].w * inv};     // masked columns hold exp(-inf) = 0
#pragma unroll
    for (int j = 0; j < 4; ++j) ov[j] -= (c + j == lab) ? 1.f : 0.f;
    if (dh) {
      bf16x4_s h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) { h[j] = (__bf16)ov[j]; l[j] = (__bf16)(ov[j] - (float)h[j]); }
      const long o = kb32_off(b, c, (int)(ld >> 5));
      *reinterpret_cast<bf16x4_s*>(dh + o) = h;
      if (dl) *reinterpret_cast<bf16x4_s*>(dl + o) = l;
    } else {
      st4(row + c, make_float4(ov[0], ov[1], ov[2], ov[3]));
    }
  }
}

// ---- softmax cross entropy from the logits GEMM's softmax epilogue (tcar_gemm_bf16_ce, gemm_bf16.hip) -----------------
// The epilogue left, per session row and column group g, (m_g, s_g) = (group maximum, sum of exp(x - m_g)), the plane
// e[b, n] = bf16(exp(x - m_g(n))) and the label's score.  Combine: M = max_g m_g, S = sum_g s_g exp(m_g - M),
// lse = M + log S, ce = lse - x_label (model_combine.py:145).  One wave per row.
__global__ __launch_bounds__(256) void ce_combine_kernel(int B, int ngroups, const float* __restrict__ stats,
                                                         const float* __restrict__ lab_logit, float* __restrict__ rowstat,
                                                         float* __restrict__ ce) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float2* st = reinterpret_cast<const float2*>(stats) + (long)b * ngroups;
  float m = -INFINITY;
  for (int g = lane; g < ngroups; g += 64) m = fmaxf(m, st[g].x);
  m = wave_max(m);
  float s = 0.f;
  for (int g = lane; g < ngroups; g += 64) {
    const float2 v = st[g];
    if (v.x != -INFINITY) s += v.y * expf(v.x - m);
  }
  s = wave_sum(s);
  if (lane == 0) {
    rowstat[2 * b] = m;
    rowstat[2 * b + 1] = 1.0f / s;
    ce[b] = m + logf(s) - lab_logit[b];
  }
}
// Rescale in place: plane[b, n] = bf16( e[b, n] * exp(m_g - M) / S - [n == label_b] ) = softmax - onehot, the gradient of the
// SUM of the per-session losses (model_combine.py:147,156); rows [B, ceil128(B)) are zeroed (they are k-rows of dE).
// One thread per 16-byte piece position (row, q) of CE_KPT consecutive 32-column blocks: a wave instruction moves 1 KB of
// CONTIGUOUS plane (16 rows x 64 bytes) and the CE_KPT loads of a thread are independent (round 4: one thread per 64-byte row
// read its four pieces with four instructions of 64-byte lane stride — 26 us for the 94 MB of the Globo plane).
constexpr int CE_KPT = 4;
// (a launch that carries a completion flag — the fork to the dE GEMM's stream — stores the plane write-through; tcar_common.h)
__device__ __forceinline__ void ce_store16(uint4* p, uint4 v, bool wt) {
  if (wt) {
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    const u4_t x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
  } else {
    *p = v;
  }
}
template <int GW>
__device__ __forceinline__ void ce_rescale_body(int B, int N, int in32, int ngroups, const float* __restrict__ stats,
                                                const float* __restrict__ rowstat, const int32_t* __restrict__ label,
                                                __bf16* __restrict__ plane, long nunits, int lab_off, int lab_window, bool wt) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nunits) return;
  const int q = (int)(i & 3), r = (int)((i >> 2) & 127);
  const long chunk = i >> 9;
  const int nch = (in32 + CE_KPT - 1) / CE_KPT;
  const long rb = chunk / nch;
  const int kb0 = (int)(chunk - rb * nch) * CE_KPT;
  const long row = rb * 128 + r;
  uint4* p = reinterpret_cast<uint4*>(plane + (rb * in32 + kb0) * 4096 + r * 32 + q * 8);      // block kb0 + j: p + 512 j
  if (row >= B) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int j = 0; j < CE_KPT; ++j)
      if (kb0 + j < in32) ce_store16(p + 512 * j, z, wt);
    return;
  }
  uint4 v[CE_KPT];
#pragma unroll
  for (int j = 0; j < CE_KPT; ++j)
    if (kb0 + j < in32) v[j] = p[512 * j];
  const float2 rs = *reinterpret_cast<const float2*>(rowstat + 2 * row);
  // label's column, if inside the catalog (label window — catalog shard —: a label of another shard matches nothing)
  const int lraw = label[row] - lab_off;
  const int labc = lab_window ? ((lraw >= 0 && lraw < N) ? lraw : -(1 << 30)) : clampi(lraw, 0, N - 1);
  const int piece = q ^ ((r >> 2) & 3);                           // storage position q holds logical piece q ^ sw
#pragma unroll
  for (int j = 0; j < CE_KPT; ++j) {
    if (kb0 + j >= in32) break;
    const int col0 = (kb0 + j) * 32;
    // (padding blocks of the plane — columns at or beyond N, or beyond the groups the GEMM wrote statistics for — scale to zero:
    //  no statistic outside [0, ngroups) is ever read)
    const int gi = col0 / GW;
    const float mg = (gi < ngroups && col0 < N) ? stats[((long)row * ngroups + gi) * 2] : -INFINITY;
    const float c = (mg == -INFINITY) ? 0.f : expf(mg - rs.x) * rs.y;
    const int lab = labc - col0 - piece * 8;                      // the label among this piece's eight columns, if 0 <= lab < 8
    unsigned w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // two bf16 per dword: low half = even element
      float lo = __uint_as_float(w[e] << 16) * c, hi = __uint_as_float(w[e] & 0xffff0000u) * c;
      if (2 * e == lab) lo -= 1.f;
      if (2 * e + 1 == lab) hi -= 1.f;
      const __bf16 bl = (__bf16)lo, bh = (__bf16)hi;
      w[e] = (unsigned)__builtin_bit_cast(unsigned short, bl) | ((unsigned)__builtin_bit_cast(unsigned short, bh) << 16);
    }
    ce_store16(p + 512 * j, make_uint4(w[0], w[1], w[2], w[3]), wt);
  }
}
template <int GW>
__global__ __launch_bounds__(256) void ce_rescale_kernel(int B, int N, int in32, int ngroups, const float* __restrict__ stats,
                                                         const float* __restrict__ rowstat, const int32_t* __restrict__ label,
                                                         __bf16* __restrict__ plane, long nunits, int lab_off, int lab_window,
                                                         const TcarSignal sig) {
  ce_rescale_body<GW>(B, N, in32, ngroups, stats, rowstat, label, plane, nunits, lab_off, lab_window, sig.cnt != nullptr);
  tcar_signal_done(sig);        // (the body is a function: its early returns end up here)
}
// ---- ANCHORED form (round 6): the plane holds exp(x - anchor[b]) with ONE reference per row — subtracted INSIDE the logits GEMM's
// contraction (anchor columns of its one-hot K segment: embed.hip; tcar_gemm_bf16_ce_o with TcarOpt::anchored), so that the GEMM's
// epilogue neither computes group maxima nor loads anything —, a row's softmax is plane / S_b with S_b = the sum of its group sums, and
// NO pass over the [B, N] plane follows the logits GEMM: this launch (one wave per session row) folds the row's group sums in a fixed
// order — ce = log S_b - (x_label - anchor), the label's accumulator as the GEMM leaves it in lab_logit — and prepares the two
// consumers, which scale PER ROW:
//   * the label's -1 goes INTO the plane as v = bf16(e_l - S_b).  Rounded like that it carries the one-hot — the LARGEST term of a
//     row's gradient while the softmax is flat — with up to 2^-8 relative error (measured: 1.6e-3 instead of 1e-4 norm-wise on the
//     time-side gradients against the fp64 oracle), so the residual d = (e_l - S_b) - v is kept in fp32;
//   * dX is linear in the plane's rows: its slab reduce multiplies by scale2[b].x = 1 / S_b and adds scale2[b].y = d / S_b times the
//     label's candidate row in fp32 (TcarRowFix) — softmax part scaled exactly, one-hot part exact;
//   * dE contracts over b, so the scale rides on attout's rows: aps[b, :] = bf16((ap_hi + ap_lo)[b, :] / S') with S' = e_l - v, for which
//     v / S' = e_l / S' - 1 EXACTLY — the one-hot part of dE (and of its (q, z) pairs, which are not linear) is exact, the rounding of
//     v becomes a common factor S_b / S' = 1 +- 2^-8 on the row's softmax part.
// 94 MB of plane traffic and ~23 us of the step's critical chain become 0.6 MB and one small launch.  The anchor is the label's
// score up to rounding (embed.hip: attout_finish_kernel), hence S_b >= ~1 (no underflow for any logits); the GEMM clamps its
// exponent at 2^100 (a logit > 69 nats above the anchor saturates): nothing overflows.
__global__ __launch_bounds__(256) void ce_anchor_fold_kernel(int B, int N, int in32, int ngroups, const float* __restrict__ stats,
                                                             const float* __restrict__ lab_logit, const int32_t* __restrict__ label,
                                                             float* __restrict__ rowstat, float* __restrict__ ce,
                                                             float* __restrict__ scale2, __bf16* __restrict__ plane,
                                                             const __bf16* __restrict__ ap_hi, const __bf16* __restrict__ ap_lo,
                                                             __bf16* __restrict__ aps, int ap_cols, int ap_in32) {
  const int lane = threadIdx.x & 63;
  const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) {      // padding rows up to the next multiple of 32 are k-rows of dE: their attout rows are ZERO (the plane's own padding
    //                  rows hold finite leftovers of earlier batches — the epilogue's exponent is clamped —, and finite times zero is zero)
    if (b < ((B + 31) & ~31))
      for (int c = lane * 4; c < ap_cols; c += 256) {
        bf16x4_s z;
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = (__bf16)0.f;
        *reinterpret_cast<bf16x4_s*>(aps + kb32_off(b, c, ap_in32)) = z;
      }
    return;
  }
  const float2* st = reinterpret_cast<const float2*>(stats) + b * ngroups;
  // per lane: groups lane, lane + 64, ... in order; then the shuffle tree (a fixed order: the same bits in every run)
  // (s: the sum of the plane's ROUNDED entries — the gradient's scale, consistent with what the GEMMs contract; st: the sum of the
  //  exponentials themselves — the loss)
  constexpr int NG = 8;
  float s = 0.f, strue = 0.f;
  if (ngroups <= 64 * NG) {
    float2 pr[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int g = lane + 64 * j;
      pr[j] = g < ngroups ? st[g] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) { s += pr[j].x; strue += pr[j].y; }
  } else {
    for (int g = lane; g < ngroups; g += 64) { s += st[g].x; strue += st[g].y; }
  }
  s = wave_sum(s);
  strue = wave_sum(strue);
  const int lab = clampi(label[b], 0, N - 1);
  __bf16* q = plane + kb32_off(b, lab, in32);
  const float el = (float)*q;                     // (every lane: the same address, the same bits)
  const float t = el - s;
  const __bf16 v = (__bf16)t;
  const float inv = 1.0f / s, inv_e = 1.0f / (el - (float)v);
  if (lane == 0) {
    if (rowstat) { rowstat[2 * b] = 0.f; rowstat[2 * b + 1] = 1.0f / strue; }         // (reference of the exponentials: the anchor = 0 here)
    ce[b] = logf(strue) - lab_logit[b];
    scale2[2 * b] = inv;
    scale2[2 * b + 1] = (t - (float)v) * inv;
    *q = v;                                       // (v depends on the wave's one load of the entry: it has returned for every lane)
  }
  for (int c = lane * 4; c < ap_cols; c += 256) {
    const long o = kb32_off(b, c, ap_in32);
    const bf16x4_s h = *reinterpret_cast<const bf16x4_s*>(ap_hi + o), l = *reinterpret_cast<const bf16x4_s*>(ap_lo + o);
    bf16x4_s y;
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = (__bf16)(inv_e * ((float)h[j] + (float)l[j]));
    *reinterpret_cast<bf16x4_s*>(aps + o) = y;
  }
}
// ... for the catalog-sharded step: the row sums crossed the ranks (rowstat[b].x = the sum over the shards of their planes' rounded
// entries, from tcar_softmax_combine_anchored), the label lives in ONE shard (window: label - lab_off
// inside [0, N) or no label entry here: no patch, no residual, plain 1 / S_b on attout's row), and a padding session (label < 0,
// lse = +inf) gets an exactly zero gradient row: scale 0 on both consumers
__global__ __launch_bounds__(256) void ce_anchor_apply_kernel(int B, int N, int in32, const float* __restrict__ rowstat,
                                                              const int32_t* __restrict__ label, int lab_off,
                                                              float* __restrict__ scale2, __bf16* __restrict__ plane,
                                                              const __bf16* __restrict__ ap_hi, const __bf16* __restrict__ ap_lo,
                                                              __bf16* __restrict__ aps, int ap_cols, int ap_in32) {
  const int lane = threadIdx.x & 63;
  const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) {      // (padding k-rows of dE: zero attout rows, as in ce_anchor_fold_kernel)
    if (b < ((B + 31) & ~31))
      for (int c = lane * 4; c < ap_cols; c += 256) {
        bf16x4_s z;
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = (__bf16)0.f;
        *reinterpret_cast<bf16x4_s*>(aps + kb32_off(b, c, ap_in32)) = z;
      }
    return;
  }
  const int lraw = label[b];
  const int lab = lraw - lab_off;
  float inv = 0.f, inv_e = 0.f, resid = 0.f;
  if (lraw >= 0) {
    const float s = rowstat[2 * b];               // (anchored convention: the sum over the shards of the planes' ROUNDED entries)
    inv = 1.0f / s;
    inv_e = inv;
    if (lab >= 0 && lab < N) {
      __bf16* q = plane + kb32_off(b, lab, in32);
      const float el = (float)*q;
      const float t = el - s;
      const __bf16 v = (__bf16)t;
      inv_e = 1.0f / (el - (float)v);
      resid = (t - (float)v) * inv;
      if (lane == 0) *q = v;
    }
  }
  if (lane == 0) { scale2[2 * b] = inv; scale2[2 * b + 1] = resid; }
  for (int c = lane * 4; c < ap_cols; c += 256) {
    const long o = kb32_off(b, c, ap_in32);
    const bf16x4_s h = *reinterpret_cast<const bf16x4_s*>(ap_hi + o), l = *reinterpret_cast<const bf16x4_s*>(ap_lo + o);
    bf16x4_s y;
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = (__bf16)(inv_e * ((float)h[j] + (float)l[j]));
    *reinterpret_cast<bf16x4_s*>(aps + o) = y;
  }
}
// ce_combine + ce_rescale in ONE launch (round 5; TCAR_CE_FOLD): a workgroup owns 16 session rows x one slice of the plane's
// column blocks.  It folds ITS rows' (max, sum) pairs itself — ngroups * 8 bytes per row out of L2, the same lane-strided loops and
// shuffle trees as ce_combine_kernel, so lse / ce / the scale come out in the same bits — and then streams its slice: a wave
// instruction is one 128 x 32 block's 16 rows x 64 bytes = 1 KB contiguous (lane = (row, 16-byte piece)), CE_FKPT independent
// loads in flight per wave, issued BEFORE the fold (they do not depend on it).  Slice 0 also writes rowstat and ce.  The launch
// saves the combine kernel (7.6 us) and its boundary on the step's critical chain; the pairs are re-read once per slice.
constexpr int CE_FKPT = 8;
template <int GW>
__global__ __launch_bounds__(256) void ce_fold_rescale_kernel(int B, int N, int in32, int ngroups, const float* __restrict__ stats,
                                                              const float* __restrict__ lab_logit, const int32_t* __restrict__ label,
                                                              float* __restrict__ rowstat, float* __restrict__ ce,
                                                              __bf16* __restrict__ plane, int nslice, int lab_off, int lab_window,
                                                              const TcarSignal sig) {
  __shared__ float sM[16], sR[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int rg = blockIdx.x / nslice, sl = blockIdx.x - rg * nslice;
  const int row0 = rg * 16;                                    // 16 rows of one 128-row block (16 divides 128)
  const int per = (in32 + nslice - 1) / nslice;
  const int kb_lo = sl * per, kb_hi = min(in32, kb_lo + per);
  const bool wt = sig.cnt != nullptr;
  const int r = (row0 & 127) + (lane >> 2), q = lane & 3;
  const long row = row0 + (lane >> 2);
  // block kb of this wave's 16 rows: 16 bytes per lane, 1 KB per wave instruction
  uint4* p0 = reinterpret_cast<uint4*>(plane + ((long)(row0 >> 7) * in32) * 4096 + r * 32 + q * 8);
  const int kb_first = kb_lo + wv;                             // wave w takes blocks kb_lo + w, + 4, ...
  if (row0 >= B) {                                             // pad rows of the last 128-row block: zero (they are k-rows of dE)
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int kb = kb_first; kb < kb_hi; kb += 4) ce_store16(p0 + 512L * kb, z, wt);
    tcar_signal_done(sig);
    return;
  }
  uint4 v[CE_FKPT];
#pragma unroll
  for (int j = 0; j < CE_FKPT; ++j) {
    const int kb = kb_first + 4 * j;
    if (kb < kb_hi) v[j] = p0[512L * kb];
  }
  // ---- fold: wave w takes rows row0 + 4 w .. + 3.  ce_combine_kernel's arithmetic (per lane: groups lane, lane + 64, ... in order;
  // then the shuffle trees), but every pair of the wave's four rows is LOADED before the first use: one L2 round trip per workgroup
  // (the first version walked two dependent loops per row — 64 serial round trips, 46 us for the launch)
  constexpr int NG = 8;                                         // pairs per lane and row held in registers: ngroups <= 512
  if (ngroups <= 64 * NG) {
    float2 pr[4][NG];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = min(row0 + 4 * wv + i, B - 1);
      const float2* st = reinterpret_cast<const float2*>(stats) + (long)b * ngroups;
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        const int g = lane + 64 * j;
        pr[i][j] = g < ngroups ? st[g] : make_float2(-INFINITY, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = row0 + 4 * wv + i;
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < NG; ++j) m = fmaxf(m, pr[i][j].x);
      m = wave_max(m);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NG; ++j)
        if (pr[i][j].x != -INFINITY) s += pr[i][j].y * expf(pr[i][j].x - m);
      s = wave_sum(s);
      if (lane == 0 && b < B) {
        const float inv = 1.0f / s;
        sM[4 * wv + i] = m;
        sR[4 * wv + i] = inv;
        if (sl == 0) {
          if (rowstat) { rowstat[2 * b] = m; rowstat[2 * b + 1] = inv; }
          ce[b] = m + logf(s) - lab_logit[b];
        }
      }
    }
  } else {
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      const int b = row0 + 4 * wv + i;
      if (b >= B) break;
      const float2* st = reinterpret_cast<const float2*>(stats) + (long)b * ngroups;
      float m = -INFINITY;
      for (int g = lane; g < ngroups; g += 64) m = fmaxf(m, st[g].x);
      m = wave_max(m);
      float s = 0.f;
      for (int g = lane; g < ngroups; g += 64) {
        const float2 x = st[g];
        if (x.x != -INFINITY) s += x.y * expf(x.x - m);
      }
      s = wave_sum(s);
      if (lane == 0) {
        const float inv = 1.0f / s;
        sM[4 * wv + i] = m;
        sR[4 * wv + i] = inv;
        if (sl == 0) {
          if (rowstat) { rowstat[2 * b] = m; rowstat[2 * b + 1] = inv; }
          ce[b] = m + logf(s) - lab_logit[b];
        }
      }
    }
  }
  __syncthreads();
  const bool live = row < B;
  const float rsx = live ? sM[lane >> 2] : 0.f, rsy = live ? sR[lane >> 2] : 0.f;
  int labc = -(1 << 30);
  if (live) {
    const int lraw = label[row] - lab_off;
    labc = lab_window ? ((lraw >= 0 && lraw < N) ? lraw : -(1 << 30)) : clampi(lraw, 0, N - 1);
  }
  const int piece = q ^ ((r >> 2) & 3);                         // storage position q holds logical piece q ^ sw
  const float* strow = stats + (long)(live ? row : 0) * ngroups * 2;
  auto rescale1 = [&](uint4 x, int kb) -> uint4 {
    const int col0 = kb * 32;
    const int gi = col0 / GW;
    const float mg = (live && gi < ngroups && col0 < N) ? strow[2 * gi] : -INFINITY;
    const float c = (mg == -INFINITY) ? 0.f : expf(mg - rsx) * rsy;
    const int lab = labc - col0 - piece * 8;
    unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float lo = __uint_as_float(w[e] << 16) * c, hi = __uint_as_float(w[e] & 0xffff0000u) * c;
      if (2 * e == lab) lo -= 1.f;
      if (2 * e + 1 == lab) hi -= 1.f;
      const __bf16 bl = (__bf16)lo, bh = (__bf16)hi;
      w[e] = (unsigned)__builtin_bit_cast(unsigned short, bl) | ((unsigned)__builtin_bit_cast(unsigned short, bh) << 16);
    }
    return live ? make_uint4(w[0], w[1], w[2], w[3]) : make_uint4(0u, 0u, 0u, 0u);
  };
  for (int kb0 = kb_first; kb0 < kb_hi; kb0 += 4 * CE_FKPT) {
    if (kb0 != kb_first) {                                     // (the first trip's loads were issued ahead of the fold)
#pragma unroll
      for (int j = 0; j < CE_FKPT; ++j) {
        const int kb = kb0 + 4 * j;
        if (kb < kb_hi) v[j] = p0[512L * kb];
      }
    }
#pragma unroll
    for (int j = 0; j < CE_FKPT; ++j) {
      const int kb = kb0 + 4 * j;
      if (kb < kb_hi) ce_store16(p0 + 512L * kb, rescale1(v[j], kb), wt);
    }
  }
  tcar_signal_done(sig);
}

// Catalog-sharded step: the shard's (max, sum exp, label score) per session from the epilogue's per-group pairs — the row of the
// statistics all-gather (shard.hip: softmax_combine).  The label's score is 0 unless the label lies in [n0, n0 + n_loc).
__global__ __launch_bounds__(256) void ce_shard_stats_kernel(int B, int ngroups, const float* __restrict__ stats,
                                                             const float* __restrict__ lab_logit, const int32_t* __restrict__ label,
                                                             int n0, int n_loc, float* __restrict__ out3, int anchored) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float2* st = reinterpret_cast<const float2*>(stats) + (long)b * ngroups;
  if (anchored) {      // pairs (sum of the plane's rounded entries, sum of the exponentials), one reference per row: plain sums
    float sr = 0.f, s2 = 0.f;
    for (int g = lane; g < ngroups; g += 64) { const float2 v = st[g]; sr += v.x; s2 += v.y; }
    sr = wave_sum(sr);
    s2 = wave_sum(s2);
    if (lane == 0) {
      const int l = label[b] - n0;
      out3[3L * b] = sr;
      out3[3L * b + 1] = s2;
      out3[3L * b + 2] = (l >= 0 && l < n_loc) ? lab_logit[b] : 0.f;
    }
    return;
  }
  float m = -INFINITY;
  for (int g = lane; g < ngroups; g += 64) m = fmaxf(m, st[g].x);
  m = wave_max(m);
  float s = 0.f;
  for (int g = lane; g < ngroups; g += 64) {
    const float2 v = st[g];
    if (v.x != -INFINITY) s += v.y * expf(v.x - m);
  }
  s = wave_sum(s);
  if (lane == 0) {
    const int l = label[b] - n0;
    out3[3L * b] = m;
    out3[3L * b + 1] = s;
    out3[3L * b + 2] = (l >= 0 && l < n_loc) ? lab_logit[b] : 0.f;
  }
}

// ---- negative-feedback term (model_combine.py:142-143) ---------------------------------------------------
// One wave per session: gathers K item|content rows of E (2 x 16 B per lane per row), dots them with attout_ic.
template <int NCH>
__global__ __launch_bounds__(256) void neg_term_kernel(int B, int K, int n_items, int ldh, int ek,
                                                       const float* __restrict__ E, const int32_t* __restrict__ neg,
                                                       const float* __restrict__ attout, float weight,
                                                       float* __restrict__ neg_fb, float* __restrict__ dattout,
                                                       float* __restrict__ g_item, const float* __restrict__ ce,
                                                       float* __restrict__ loss, float* __restrict__ coef_out,
                                                       long datt_ld, int datt_overwrite, TcarSignal sig) {
  // one WORKGROUP per session: the 4 waves split the K negatives (independent row gathers in flight), partial
  // dot / row sums meet in LDS, every wave then scatters its own negatives' gradient rows
  __shared__ __attribute__((aligned(16))) float part[4 * 2 * 512];   // [wave][item|content sums]
  __shared__ float px[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.x;
  const int ic = 2 * ldh;
  float4 ua[NCH], ub[NCH], sa[NCH], sb[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
    const bool ok = col < ldh;
    ua[c] = ok ? ld4(attout + (long)b * ek + col) : zero4();
    ub[c] = ok ? ld4(attout + (long)b * ek + ldh + col) : zero4();
    sa[c] = zero4(); sb[c] = zero4();
  }
  float x = 0.f;
  constexpr int U = 4;                  // rows in flight per wave: all loads of a group are issued before the first use
  for (int kb = w; kb < K; kb += 4 * U) {
    float4 ra[U][NCH], rb[U][NCH];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = kb + 4 * u;
      const bool live = k < K;
      const int n = live ? clampi(neg[(long)b * K + k], 0, n_items - 1) : 0;
      const float* e = E + (long)n * ek;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        const bool ok = live && col < ldh;
        ra[u][c] = ok ? ld4(e + col) : zero4();
        rb[u][c] = ok ? ld4(e + ldh + col) : zero4();
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        x += dot4(ra[u][c], ua[c]) + dot4(rb[u][c], ub[c]);
        sa[c] = add4(sa[c], ra[u][c]); sb[c] = add4(sb[c], rb[u][c]);
      }
  }
  x = wave_sum(x);
  if (lane == 0) px[w] = x;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
    if (col < ldh) { st4(part + w * ic + col, sa[c]); st4(part + w * ic + ldh + col, sb[c]); }
  }
  __syncthreads();
  x = px[0] + px[1] + px[2] + px[3];                   // sum over K BEFORE the sigmoid (model_combine.py:142)
  const float sg = 1.0f / (1.0f + expf(-x));
  const float om = 1.0f - sg;
  const float fb = -logf(om + 1e-24f);
  if (tid == 0 && neg_fb) neg_fb[b] = fb;
  if (tid == 0 && loss) loss[b] = ce[b] + weight * fb;          // model_combine.py:147
  const float coef = weight * sg * om / (om + 1e-24f);  // weight * d/dx[-log(1 - sigmoid(x) + 1e-24)]
  if (tid == 0 && coef_out) coef_out[b] = coef;
  if (dattout) {
    for (int col = tid * 4; col < ic; col += 1024) {
      const float4 s = add4(add4(ld4(part + col), ld4(part + ic + col)), add4(ld4(part + 2 * ic + col), ld4(part + 3 * ic + col)));
      float* p = dattout + (long)b * datt_ld + col;
      // (behind a completion flag the other stream reads this row: write-through, tcar_common.h)
      if (sig.cnt) st4_sc1(p, datt_overwrite ? scale4(s, coef) : fma4(s, coef, ld4(p)));
      else st4(p, datt_overwrite ? scale4(s, coef) : fma4(s, coef, ld4(p)));
    }
  }
  if (g_item) {
    for (int k = w; k < K; k += 4) {
      const int n = clampi(neg[(long)b * K + k], 0, n_items - 1);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < ldh) atomic_add4(g_item + (long)n * ldh + col, scale4(ua[c], coef));
      }
    }
  }
  tcar_signal_done(sig);
}

// ---- negative rows of the item-table gradient: g_item[neg[b,k], :] += coef[b] * attout[b, 0:ldh] --------------
// One wave per (session, negative); lanes walk consecutive columns so each atomic instruction covers 256 contiguous
// bytes.  The first wave of a session also writes the training loss (model_combine.py:147).
__global__ __launch_bounds__(256) void neg_scatter_kernel(int B, int K, int n_items, int ldh, int ek,
                                                          const int32_t* __restrict__ neg, const float* __restrict__ attout,
                                                          const float* __restrict__ coef, float* __restrict__ g_item,
                                                          const float* __restrict__ neg_fb, const float* __restrict__ ce,
                                                          float weight, float* __restrict__ loss) {
  const int lane = threadIdx.x & 63;
  const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wv >= (long)B * K) return;
  const int b = (int)(wv / K), k = (int)(wv - (long)b * K);
  if (k == 0 && lane == 0 && loss) loss[b] = ce[b] + weight * neg_fb[b];
  const int n = clampi(neg[wv], 0, n_items - 1);
  const float cf = coef[b];
  if (cf == 0.f) return;                        // saturated sigmoid (S8): no gradient
  const float* a = attout + (long)b * ek;
  float* gdst = g_item + (long)n * ldh;
  for (int col = lane; col < ldh; col += 64) atomicAdd(gdst + col, cf * a[col]);
}

// ---- dattout = sum of the split-K slabs [+ negative part], times act'(attout), plus the bias gradients -----------
// One pass instead of four launches (slab reduce, negative term, two activation backward passes): tile = 16 rows x 64
// columns (416 workgroups at B = 512, ek = 832), thread = (float4 column group, row); twelve slab loads are in flight per
// thread before the first add (the pass is a 61-MB read at the Globo shape: 36 slabs of [512, 832]).
__global__ __launch_bounds__(256) void reduce_dact_kernel(const float* __restrict__ slabs, int S, int M, int N, long ld,
                                                          const float* __restrict__ addend, long ld_add, int n_add,
                                                          const float* __restrict__ y, long ldy, int act,
                                                          float* __restrict__ out, float* __restrict__ bg0, int split_col,
                                                          float* __restrict__ bg1) {
  __shared__ float4 sh[256];
  const int tid = threadIdx.x, cg = tid & 15, rp = tid >> 4;
  const int col = blockIdx.x * 64 + cg * 4;
  const int r0 = blockIdx.y * 16;
  float4 cs = zero4();
  if (col < N) {
    {
      const int row = r0 + rp;
      if (row < M) {
      float4 acc = (addend && col < n_add) ? ld4(addend + (long)row * ld_add + col) : zero4();
      const float* sp = slabs + (long)row * ld + col;
      int k = 0;
      for (; k + 12 <= S; k += 12) {
        float4 t[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) t[j] = ld4(sp + (long)(k + j) * M * ld);
#pragma unroll
        for (int j = 0; j < 12; ++j) acc = add4(acc, t[j]);
      }
      for (; k < S; ++k) acc = add4(acc, ld4(sp + (long)k * M * ld));
      if (act) {
        const float4 yy = ld4(y + (long)row * ldy + col);
        if (act == 1) {
          acc.x = yy.x > 0.f ? acc.x : 0.f; acc.y = yy.y > 0.f ? acc.y : 0.f;
          acc.z = yy.z > 0.f ? acc.z : 0.f; acc.w = yy.w > 0.f ? acc.w : 0.f;
        } else {
          acc.x *= 1.f - yy.x * yy.x; acc.y *= 1.f - yy.y * yy.y; acc.z *= 1.f - yy.z * yy.z; acc.w *= 1.f - yy.w * yy.w;
        }
      }
      st4(out + (long)row * ld + col, acc);
      cs = add4(cs, acc);
      }
    }
  }
  sh[tid] = cs;
  __syncthreads();
  if (tid < 64) {                                // thread = one column of the tile: sum its 16 row phases
    const int g4 = tid >> 2, j = tid & 3;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float4 t = sh[q * 16 + g4];
      v += (j == 0) ? t.x : (j == 1) ? t.y : (j == 2) ? t.z : t.w;
    }
    const int c = blockIdx.x * 64 + tid;
    if (c < N && v != 0.f) {
      if (c < split_col) { if (bg0) atomicAdd(bg0 + c, v); }
      else if (bg1) atomicAdd(bg1 + (c - split_col), v);
    }
  }
}

// DPP row_share: lane I of every 16-lane row broadcast to its row (full-rate VALU, no LDS crossbar)
template <int I>
__device__ __forceinline__ float row_share(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + I, 0xf, 0xf, false));
}
template <int... I>
__device__ __forceinline__ void expand16(float4& acc, const float4 (&xr)[16], float v, std::integer_sequence<int, I...>) {
  ((acc = fma4(xr[I], row_share<I>(v), acc)), ...);
}

// ---- the same for the ONE-HOT form of dX (round 4): slabs [S][M][lds] hold dlogits [E_item | E_content | OH] -------------------
// Column blocks 0 .. ic/64 - 1: as reduce_dact_kernel.  Block ic/64 + k (k = 0..4): dP[m, rows of table k] = sum of the slabs'
// one-hot columns (written to dP [M, 160] for the candidate-side table gradients, embed.hip), expanded on the spot to the time
// columns of dattout: d attout_t,k[m, :] = sum_r dP[m, r] clip(table_k row r)  (dlogits E_time = (dlogits OH) T_clip), then tanh'.
__global__ __launch_bounds__(256) void reduce_dact_onehot_kernel(const float* __restrict__ slabs, int S, int M, int ic, long lds_,
                                                                 const float* __restrict__ addend, long ld_add,
                                                                 const float* __restrict__ y, long ldy, const float* __restrict__ tclip,
                                                                 float* __restrict__ out, long ldo, float* __restrict__ dP,
                                                                 float* __restrict__ bg0, float* __restrict__ bg1, TcarSignal sig,
                                                                 const TcarWait wait_add, const TcarRowFix fix) {
  __shared__ float4 sh[256];           // (bias column sums only: the atomic mode)
  const int tid = threadIdx.x, cg = tid & 15, rp = tid >> 4;
  const int nic = ic >> 6;
  const int r0 = blockIdx.y * 16, row = r0 + rp;
  // anchored softmax form (ce_anchor_fold_kernel, TcarRowFix): the slabs hold (unscaled plane) [E | OH]; the row's 1 / S_b and the
  // fp32 residual of its one-hot come in here
  const bool fx = fix.scale2 != nullptr;
  float rs = 1.f, rd = 0.f;
  int flab = 0;
  float4 fe = zero4();       // the label's candidate row / publish-time index of this thread's columns: in flight beside the slab loads
  int fk = -1;
  if (fx && row < M) {
    rs = fix.scale2[2 * row]; rd = fix.scale2[2 * row + 1];
    flab = clampi(fix.label[row] - fix.lab_off, 0, fix.n_items - 1);
    if ((int)blockIdx.x < nic) fe = ld4(fix.E + (long)flab * fix.ldE + blockIdx.x * 64 + cg * 4);
    else fk = fix.mwdhm[(long)flab * 5 + (blockIdx.x - nic)];
  }
  float4 cs = zero4();
  int col;                                   // output column of this thread's float4
  if ((int)blockIdx.x < nic) {
    col = blockIdx.x * 64 + cg * 4;
    float4 acc = zero4();
    if (row < M) {
      const float* sp = slabs + (long)row * lds_ + col;
      int k = 0;
      for (; k + 9 <= S; k += 9) {           // (18 slabs by default: two rounds of nine loads in flight, no dependent tail)
        float4 t[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) t[j] = ld4(sp + (long)(k + j) * M * lds_);
#pragma unroll
        for (int j = 0; j < 9; ++j) acc = add4(acc, t[j]);
      }
      for (; k < S; ++k) acc = add4(acc, ld4(sp + (long)k * M * lds_));
      if (fx) acc = make_float4(fmaf(rd, fe.x, acc.x * rs), fmaf(rd, fe.y, acc.y * rs), fmaf(rd, fe.z, acc.z * rs), fmaf(rd, fe.w, acc.w * rs));
    }
    // the negative term's part comes from the aux stream: behind its flag, waited for HERE (after the slab sum), or an event
    if (addend) tcar_wave_wait(wait_add);
    if (row < M) {
      if (addend) acc = add4(wait_add.flag ? ld4_sc1(addend + (long)row * ld_add + col) : ld4(addend + (long)row * ld_add + col), acc);
      if (y) {          // (y = NULL, catalog-sharded step: the partial sums leave before tanh' — it follows the exchange)
        const float4 yy = ld4(y + (long)row * ldy + col);
        acc.x *= 1.f - yy.x * yy.x; acc.y *= 1.f - yy.y * yy.y; acc.z *= 1.f - yy.z * yy.z; acc.w *= 1.f - yy.w * yy.w;
      }
      st4(out + (long)row * ldo + col, acc);
      cs = acc;
    }
  } else {
    const int k = blockIdx.x - nic;
    const int off = k == 0 ? 0 : k == 1 ? 13 : k == 2 ? 45 : k == 3 ? 53 : 78;
    const int nk = k == 0 ? 13 : k == 1 ? 32 : k == 2 ? 8 : k == 3 ? 25 : 61;
    col = ic + k * 64 + cg * 4;
    // dP tile: 16 rows x nk one-hot columns, summed over the slabs in slab order: thread (row rp, lane cg of its 16-lane group)
    // holds the columns cg, cg + 16, cg + 32, cg + 48 of its row in registers
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cg + 16 * j;
      v[j] = 0.f;
      if (row < M && c < nk) {
        const float* sp = slabs + (long)row * lds_ + ic + off + c;
        int q = 0;
        for (; q + 6 <= S; q += 6) {
          float t[6];
#pragma unroll
          for (int u = 0; u < 6; ++u) t[u] = sp[(long)(q + u) * M * lds_];
#pragma unroll
          for (int u = 0; u < 6; ++u) v[j] += t[u];
        }
        for (; q < S; ++q) v[j] += sp[(long)q * M * lds_];
        if (fx) v[j] = fmaf(v[j], rs, (clampi(fk, 0, nk - 1) == c) ? rd : 0.f);
        if (sig.cnt) __hip_atomic_store(dP + (long)row * 160 + off + c, v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else dP[(long)row * 160 + off + c] = v[j];
      }
    }
    // expansion d attout_t,k[row, 4 cg ..] = sum_r dP[row, r] clip(table_k row r)[4 cg ..]: dP[row, r] comes out of the registers of
    // the row's own 16-lane group (a cross-lane read), the clipped rows (<= 61 x 256 B, written by tcar_time_scores_clip in the
    // forward pass) straight from L1 / L2 — every row group of the chip reads the same 15 KB.  NO LDS in this kernel: a first
    // version staged both through LDS and, running beside the dE GEMM's LDS-DMA workgroups in the step, dropped single terms of
    // single rows now and then (bit-for-bit repeatable alone; tools/det_probe.py, profiles/r04_ab_experiments.txt)
    const int lane = tid & 63, g0 = lane & 48;
    float4 acc = zero4();
    const float* tp = tclip + (long)off * 64 + cg * 4;
    (void)g0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (16 * j >= nk) break;                                    // (workgroup-uniform)
      float4 xr[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) xr[i] = (16 * j + i < nk) ? ld4(tp + (long)(16 * j + i) * 64) : zero4();      // 16 rows in flight
      expand16(acc, xr, v[j], std::make_integer_sequence<int, 16>{});
    }
    if (row < M) {
      if (y) {
        const float4 yy = ld4(y + (long)row * ldy + col);
        acc.x *= 1.f - yy.x * yy.x; acc.y *= 1.f - yy.y * yy.y; acc.z *= 1.f - yy.z * yy.z; acc.w *= 1.f - yy.w * yy.w;
      }
      st4(out + (long)row * ldo + col, acc);
      cs = acc;
    }
  }
  if (bg0 || bg1) {
    sh[tid] = cs;
    __syncthreads();
    if (tid < 64) {                                // thread = one column of the tile: sum its 16 row phases
      const int g4 = tid >> 2, j = tid & 3;
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 t = sh[q * 16 + g4];
        v += (j == 0) ? t.x : (j == 1) ? t.y : (j == 2) ? t.z : t.w;
      }
      const int c = (col - cg * 4) + tid;
      if (v != 0.f) {
        if (c < ic) { if (bg0) atomicAdd(bg0 + c, v); }
        else if (bg1) atomicAdd(bg1 + (c - ic), v);
      }
    }
  }
  tcar_signal_done(sig);
}

#ifdef TCAR_OBS1_DIAG
// DIAGNOSTIC ONLY (-DTCAR_OBS1_DIAG, tools/obs1_probe.py): round 4's FIRST form of the kernel above (git f933989), which staged the
// clipped rows and the dP tile through LDS and — in the step, beside the dE GEMM — "dropped single terms of single rows now and
// then" (observation 1).  Kept out of the product build; selected at run time by TCAR_OBS1_LDS=1 in a diagnostic build.
__global__ __launch_bounds__(256) void reduce_dact_onehot_lds_kernel(const float* __restrict__ slabs, int S, int M, int ic, long lds_,
                                                                     const float* __restrict__ addend, long ld_add,
                                                                     const float* __restrict__ y, long ldy, const float* __restrict__ tclip,
                                                                     float* __restrict__ out, long ldo, float* __restrict__ dP, TcarSignal sig) {
  __shared__ float dpl[16 * 64];
  __shared__ __attribute__((aligned(16))) float tl[61 * 64];
  const int tid = threadIdx.x, cg = tid & 15, rp = tid >> 4;
  const int nic = ic >> 6;
  const int r0 = blockIdx.y * 16, row = r0 + rp;
  int col;
  if ((int)blockIdx.x < nic) {
    col = blockIdx.x * 64 + cg * 4;
    if (row < M) {
      float4 acc = addend ? ld4(addend + (long)row * ld_add + col) : zero4();
      const float* sp = slabs + (long)row * lds_ + col;
      int k = 0;
      for (; k + 12 <= S; k += 12) {
        float4 t[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) t[j] = ld4(sp + (long)(k + j) * M * lds_);
#pragma unroll
        for (int j = 0; j < 12; ++j) acc = add4(acc, t[j]);
      }
      for (; k < S; ++k) acc = add4(acc, ld4(sp + (long)k * M * lds_));
      const float4 yy = ld4(y + (long)row * ldy + col);
      acc.x *= 1.f - yy.x * yy.x; acc.y *= 1.f - yy.y * yy.y; acc.z *= 1.f - yy.z * yy.z; acc.w *= 1.f - yy.w * yy.w;
      st4(out + (long)row * ldo + col, acc);
    }
  } else {
    const int k = blockIdx.x - nic;
    const int off = k == 0 ? 0 : k == 1 ? 13 : k == 2 ? 45 : k == 3 ? 53 : 78;
    const int nk = k == 0 ? 13 : k == 1 ? 32 : k == 2 ? 8 : k == 3 ? 25 : 61;
    col = ic + k * 64 + cg * 4;
    for (int i = tid; i < nk * 16; i += 256) st4(tl + i * 4, ld4(tclip + (long)off * 64 + i * 4));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cg + 16 * j;
      float v = 0.f;
      if (row < M && c < nk) {
        const float* sp = slabs + (long)row * lds_ + ic + off + c;
        int q = 0;
        for (; q + 6 <= S; q += 6) {
          float t[6];
#pragma unroll
          for (int u = 0; u < 6; ++u) t[u] = sp[(long)(q + u) * M * lds_];
#pragma unroll
          for (int u = 0; u < 6; ++u) v += t[u];
        }
        for (; q < S; ++q) v += sp[(long)q * M * lds_];
        if (sig.cnt) __hip_atomic_store(dP + (long)row * 160 + off + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else dP[(long)row * 160 + off + c] = v;
      }
      dpl[rp * 64 + c] = v;
    }
    __syncthreads();
    if (row < M) {
      float4 acc = zero4();
      for (int r = 0; r < nk; ++r) acc = fma4(*reinterpret_cast<const float4*>(tl + r * 64 + cg * 4), dpl[rp * 64 + r], acc);
      const float4 yy = ld4(y + (long)row * ldy + col);
      acc.x *= 1.f - yy.x * yy.x; acc.y *= 1.f - yy.y * yy.y; acc.z *= 1.f - yy.z * yy.z; acc.w *= 1.f - yy.w * yy.w;
      st4(out + (long)row * ldo + col, acc);
    }
  }
  tcar_signal_done(sig);
}
#endif

// ---- dz = dy * act'(y), bias_grad += column sums (modules.py:52-54 backward) ------------------------------
// grid = (ncol/64 column blocks, row chunks); 256 threads = 64 columns x 4 row phases; one atomic per column
// and workgroup into the (zeroed) bias gradient.
__global__ __launch_bounds__(256) void dact_colsum_kernel(int M, int ncol, long ld, const float* __restrict__ y,
                                                          float* __restrict__ dy, float* __restrict__ bias_grad, int act) {
  __shared__ float sh[256];
  const int tid = threadIdx.x, cl = tid & 63, rp = tid >> 6;
  const int col = blockIdx.x * 64 + cl;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float s = 0.f;
  if (col < ncol) {
    for (int r = r0 + rp; r < r1; r += 4) {
      const long i = (long)r * ld + col;
      const float yy = y[i];
      float g = dy[i];
      g = (act == 1) ? (yy > 0.f ? g : 0.f) : (act == 2 ? g * (1.f - yy * yy) : g);
      dy[i] = g;
      s += g;
    }
  }
  sh[tid] = s;
  __syncthreads();
  if (rp == 0 && col < ncol && bias_grad) {
    const float v = sh[cl] + sh[64 + cl] + sh[128 + cl] + sh[192 + cl];
    if (v != 0.f) atomicAdd(bias_grad + col, v);
  }
}

// ---- rank of the label and top-k (util.py:13-17, model_combine.py:301) -----------------------------------
// One workgroup per session.  rank = 1 + #{n: x[n] > x[label]}.  top-k by k rounds of block-wide arg-max over
// keys ordered by (score desc, index desc) — the order of np.argsort(x)[::-1]; the row stays in L2 between
// rounds.
__global__ __launch_bounds__(256) void rank_topk_kernel(int N, const float* __restrict__ logits, long ld,
                                                        const int32_t* __restrict__ label, int k,
                                                        int32_t* __restrict__ rank, int32_t* __restrict__ topk) {
  __shared__ float shv[4];
  __shared__ int shi[4];
  __shared__ float sh[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* row = logits + (long)b * ld;
  const int lab = clampi(label[b], 0, N - 1);
  const float xl = row[lab];
  float cnt = 0.f;
  for (int i = tid; i < N; i += 256) cnt += (row[i] > xl) ? 1.f : 0.f;
  const float tot = block_sum_256(cnt, sh);
  if (tid == 0) rank[b] = (int)tot + 1;
  float pv = INFINITY;
  int pi = N;                       // previous pick; next key must be strictly "smaller" than (pv, pi)
  const int kk = k < N ? k : N;
  for (int r = 0; r < k; ++r) {
    if (r >= kk) { if (tid == 0) topk[(long)b * k + r] = -1; continue; }
    float bv = -INFINITY;
    int bi = -1;
    for (int i = tid; i < N; i += 256) {
      const float v = row[i];
      const bool eligible = (v < pv) || (v == pv && i < pi);
      const bool better = (v > bv) || (v == bv && i > bi);
      if (eligible && better) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi > bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { shv[w] = bv; shi[w] = bi; }
    __syncthreads();
    float fv = shv[0];
    int fi = shi[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
      if (shv[j] > fv || (shv[j] == fv && shi[j] > fi)) { fv = shv[j]; fi = shi[j]; }
    __syncthreads();
    if (tid == 0) topk[(long)b * k + r] = fi;
    pv = fv; pi = fi;
  }
}

// Row-resident rank / top-k (catalogs up to NT*4*R items): the row is read from memory ONCE into registers.  Every thread
// caches the best of its own elements under the total order (score desc, index desc); a round is one block-wide arg-max
// over the cached bests, after which only the WINNER rescans its 4*R elements below the extracted key.  21 passes over a
// 184-KB row (1.34 ms per launch at B = 512, N = 46,033) become one read + k cheap rounds.
template <int NT, int R>
__global__ __launch_bounds__(NT) void rank_topk_rows_kernel(int N, const float* __restrict__ logits, long ld,
                                                            const int32_t* __restrict__ label, int k,
                                                            int32_t* __restrict__ rank, int32_t* __restrict__ topk,
                                                            float* __restrict__ ce) {
  constexpr int NWV = NT / 64;
  __shared__ float shv[NWV];
  __shared__ int shi[NWV];
  __shared__ float shs[NWV];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* row = logits + (long)b * ld;
  const int lab = clampi(label[b], 0, N - 1);
  const float ninf = -INFINITY;
  float4 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int c = (tid + r * NT) * 4;
    v[r] = (c < (int)ld) ? ld4(row + c) : make_float4(ninf, ninf, ninf, ninf);
  }
  const float xl = row[lab];
  float cnt = 0.f, m = ninf;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int c = (tid + r * NT) * 4;
    if (c + 0 >= N) v[r].x = ninf;
    if (c + 1 >= N) v[r].y = ninf;
    if (c + 2 >= N) v[r].z = ninf;
    if (c + 3 >= N) v[r].w = ninf;
    cnt += (v[r].x > xl ? 1.f : 0.f) + (v[r].y > xl ? 1.f : 0.f) + (v[r].z > xl ? 1.f : 0.f) + (v[r].w > xl ? 1.f : 0.f);
    m = fmaxf(m, fmaxf(fmaxf(v[r].x, v[r].y), fmaxf(v[r].z, v[r].w)));
  }
  cnt = wave_sum(cnt);
  m = wave_max(m);
  if (lane == 0) { shs[w] = cnt; shv[w] = m; }
  __syncthreads();
  float tot = 0.f, gm = shv[0];
#pragma unroll
  for (int i = 0; i < NWV; ++i) { tot += shs[i]; gm = fmaxf(gm, shv[i]); }
  if (tid == 0) rank[b] = (int)tot + 1;
  __syncthreads();
  if (ce) {                         // sparse softmax cross entropy of the same row (model_combine.py:145)
    float se = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) se += (expf(v[r].x - gm) + expf(v[r].y - gm)) + (expf(v[r].z - gm) + expf(v[r].w - gm));
    se = wave_sum(se);
    if (lane == 0) shs[w] = se;
    __syncthreads();
    float gs = 0.f;
#pragma unroll
    for (int i = 0; i < NWV; ++i) gs += shs[i];
    if (tid == 0) ce[b] = gm + logf(gs) - xl;
    __syncthreads();
  }
  // best of this thread's elements strictly below the key (pv, pi) in the order (score desc, index desc)
  auto scan = [&](float pv, int pi, float& bv, int& bi) {
    bv = ninf; bi = -1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int c = (tid + r * NT) * 4;
      const float e[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = c + j;
        const bool eligible = (e[j] < pv) || (e[j] == pv && i < pi);
        const bool better = (e[j] > bv) || (e[j] == bv && i > bi);
        if (i < N && eligible && better) { bv = e[j]; bi = i; }
      }
    }
  };
  float cbv;
  int cbi;
  scan(INFINITY, N, cbv, cbi);
  const int kk = k < N ? k : N;
  // block-wide arg-max of one (value, index) key per thread under (score desc, index desc); result to every thread
  auto block_best = [&](float bv, int bi, float& fv, int& fi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi > bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { shv[w] = bv; shi[w] = bi; }
    __syncthreads();
    fv = shv[0]; fi = shi[0];
#pragma unroll
    for (int j = 1; j < NWV; ++j)
      if (shv[j] > fv || (shv[j] == fv && shi[j] > fi)) { fv = shv[j]; fi = shi[j]; }
    __syncthreads();
  };
  // Phase A: the kk-th largest of the per-thread bests, L.  At least kk elements are >= L, so the whole top-kk is.
  float lv = INFINITY;
  int li = N;
  {
    float av = cbv;
    int ai = cbi;
    for (int r = 0; r < kk; ++r) {
      float fv; int fi;
      block_best(av, ai, fv, fi);
      if (fi < 0) { lv = ninf; li = -1; break; }        // fewer than kk threads hold anything: every element qualifies
      lv = fv; li = fi;
      if (ai == fi) { av = ninf; ai = -1; }
    }
  }
  // Phase B: compact the elements >= L into LDS
  constexpr int CAP = 2 * NT;
  __shared__ float candv[CAP];
  __shared__ int candi[CAP];
  __shared__ int ncand;
  if (tid == 0) ncand = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int c = (tid + r * NT) * 4;
    const float e[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = c + j;
      if (i < N && (e[j] > lv || (e[j] == lv && i >= li))) {
        const int pos = atomicAdd(&ncand, 1);
        if (pos < CAP) { candv[pos] = e[j]; candi[pos] = i; }
      }
    }
  }
  __syncthreads();
  const int nc = ncand;
  if (nc <= CAP) {
    // Phase C: kk rounds over <= 2 candidates per thread
    float c0v = tid < nc ? candv[tid] : ninf, c1v = tid + NT < nc ? candv[tid + NT] : ninf;
    int c0i = tid < nc ? candi[tid] : -1, c1i = tid + NT < nc ? candi[tid + NT] : -1;
    for (int r = 0; r < k; ++r) {
      if (r >= kk) { if (tid == 0) topk[(long)b * k + r] = -1; continue; }
      const bool first = (c0v > c1v) || (c0v == c1v && c0i > c1i);
      float fv; int fi;
      block_best(first ? c0v : c1v, first ? c0i : c1i, fv, fi);
      if (tid == 0) topk[(long)b * k + r] = fi;
      if (c0i == fi) { c0v = ninf; c0i = -1; }
      if (c1i == fi) { c1v = ninf; c1i = -1; }
    }
    return;
  }
  // Fallback (more than CAP elements tie with or exceed L, e.g. a constant row): extract one element per round; only the
  // owner of the extracted element rescans its registers
  for (int r = 0; r < k; ++r) {
    if (r >= kk) { if (tid == 0) topk[(long)b * k + r] = -1; continue; }
    float fv; int fi;
    block_best(cbv, cbi, fv, fi);
    if (tid == 0) topk[(long)b * k + r] = fi;
    if (fi >= 0 && ((fi >> 2) % NT) == tid) scan(fv, fi, cbv, cbi);
  }
}

}  // namespace

extern "C" int tcar_softmax_ce(int B, int N, float* logits, int64_t ld, const int32_t* label, float* ce, void* stream) {
  return tcar_softmax_ce_bf16(B, N, logits, ld, label, ce, nullptr, nullptr, stream);
}

extern "C" int tcar_softmax_ce_bf16(int B, int N, float* logits, int64_t ld, const int32_t* label, float* ce,
                                    void* dl_hi, void* dl_lo, void* stream) {
  return tcar_softmax_ce_bf16_o(B, N, logits, ld, label, ce, dl_hi, dl_lo, stream);
}
int tcar_softmax_ce_bf16_o(int B, int N, float* logits, int64_t ld, const int32_t* label, float* ce, void* dl_hi, void* dl_lo,
                           void* stream) {
  if (B <= 0) return TCAR_OK;
  if (N <= 0 || ld < N || (ld & 3) || !tcar_aligned16(logits) || !label || !ce || (dl_hi && (ld & 31)) || (dl_lo && !dl_hi))
    return TCAR_E_ARG;
  const int grid = dl_hi ? ((B + 127) & ~127) : B;
  if (ld <= 512L * 4 * 24) {      // the row fits the register-resident variant (catalogs up to 49,152 items)
    if (ld <= 512L * 4 * 8)
      TCAR_LAUNCH((softmax_ce_rows_kernel<512, 8>), dim3(grid), dim3(512), 0, (hipStream_t)stream, B, N, logits, (long)ld,
                  label, ce, (__bf16*)dl_hi, (__bf16*)dl_lo);
    else
      TCAR_LAUNCH((softmax_ce_rows_kernel<512, 24>), dim3(grid), dim3(512), 0, (hipStream_t)stream, B, N, logits, (long)ld,
                  label, ce, (__bf16*)dl_hi, (__bf16*)dl_lo);
    TCAR_CHECK_LAUNCH();
    return TCAR_OK;
  }
  TCAR_LAUNCH(softmax_ce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, N, logits, (long)ld, label, ce,
              (__bf16*)dl_hi, (__bf16*)dl_lo);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

int tcar_ce_rescale_o(int B, int N, int group_width, int ngroups, const float* stats, const float* rowstat, const int32_t* label,
                      int lab_off, int lab_window, void* dl_hi, int64_t inner, void* stream, TcarOpt* o);
// (flag-capable: with a completion flag the rescale stores the plane write-through — what the flag's consumer, the dE GEMM on the aux
//  stream, reads)
int tcar_ce_finish_o(int B, int N, int group_width, int ngroups, const float* stats, const float* lab_logit, const int32_t* label,
                     float* rowstat, float* ce, void* dl_hi, int64_t inner, void* stream, TcarOpt* o) {
  if (B <= 0) return TCAR_OK;
  if (N <= 0 || !stats || !lab_logit || !label || !rowstat || !ce || !dl_hi || (inner & 31) || inner < N || ngroups <= 0 ||
      (group_width != 64 && group_width != 96) || (long)ngroups * group_width < N || ((uintptr_t)rowstat & 7))
    return TCAR_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  // TCAR_CE_FOLD = w > 0 (default 1024): ONE launch of about w workgroups, each folding its own 16 rows' pairs (ce_fold_rescale_kernel);
  // 0: the combine launch + the rescale launch (same bits either way)
  // (every slice re-reads its rows' pairs: fine while the [B, ngroups, 2] statistics sit in the L2s — 2 MB at the Globo shape —, a
  //  multiple of the plane's own traffic beyond; a 10 M-item catalog keeps the two launches)
  const int fold = tcar_tn(o).ce_fold;
  if (fold > 0 && (int64_t)B * ngroups * 8 <= (8LL << 20)) {
    const int in32 = (int)(inner >> 5);
    const int rgroups = (((B + 127) >> 7) << 7) / 16;
    int nslice = fold / rgroups;
    if (nslice > in32 / 4) nslice = in32 / 4;
    if (nslice < 1) nslice = 1;
    const TcarSignal sig = tcar_sig(o);
    if (group_width == 96)
      TCAR_LAUNCH(ce_fold_rescale_kernel<96>, dim3(rgroups * nslice), dim3(256), 0, st, B, N, in32, ngroups, stats, lab_logit, label,
                  rowstat, ce, (__bf16*)dl_hi, nslice, 0, 0, sig);
    else
      TCAR_LAUNCH(ce_fold_rescale_kernel<64>, dim3(rgroups * nslice), dim3(256), 0, st, B, N, in32, ngroups, stats, lab_logit, label,
                  rowstat, ce, (__bf16*)dl_hi, nslice, 0, 0, sig);
    TCAR_CHECK_LAUNCH();
    return TCAR_OK;
  }
  TCAR_LAUNCH(ce_combine_kernel, dim3((B + 3) / 4), dim3(256), 0, st, B, ngroups, stats, lab_logit, rowstat, ce);
  TCAR_CHECK_LAUNCH();
  return tcar_ce_rescale_o(B, N, group_width, ngroups, stats, rowstat, label, 0, 0, dl_hi, inner, stream, o);
}
extern "C" int tcar_ce_finish(int B, int N, int group_width, int ngroups, const float* stats, const float* lab_logit,
                              const int32_t* label, float* rowstat, float* ce, void* dl_hi, int64_t inner, void* stream) {
  return tcar_ce_finish_o(B, N, group_width, ngroups, stats, lab_logit, label, rowstat, ce, dl_hi, inner, stream, nullptr);
}

// anchored form of tcar_ce_finish (see ce_anchor_fold_kernel); any B (planes of ceil128(B) rows; the fold zeroes the scaled attout rows
// [B, ceil32(B)), the k-rows of dE beyond the batch); ap_* / aps: the packed attout planes of the dE GEMM [B, ap_cols] with inner dimension ap_inner, KB32 layout
int tcar_ce_anchor_fold_o(int B, int N, int group_width, int ngroups, const float* stats, const float* lab_logit, const int32_t* label,
                          float* rowstat, float* ce, float* scale2, void* dl_hi, int64_t inner, const void* ap_hi, const void* ap_lo,
                          void* aps_hi, int ap_cols, int64_t ap_inner, void* stream, TcarOpt* o) {
  (void)o;
  if (B <= 0) return TCAR_OK;
  if (N <= 0 || !stats || !lab_logit || !label || !ce || !scale2 || ((uintptr_t)scale2 & 7) || !dl_hi || (inner & 31) || inner < N || ngroups <= 0 ||
      (group_width != 64 && group_width != 96) || (long)ngroups * group_width < N || (rowstat && ((uintptr_t)rowstat & 7)) || !ap_hi ||
      !ap_lo || !aps_hi || ap_cols <= 0 || (ap_cols & 3) || (ap_inner & 31) || ap_inner < ap_cols || ((uintptr_t)stats & 7))
    return TCAR_E_ARG;
  TCAR_LAUNCH(ce_anchor_fold_kernel, dim3((((B + 31) & ~31) + 3) / 4), dim3(256), 0, (hipStream_t)stream, B, N, (int)(inner >> 5), ngroups, stats, lab_logit,
              label, rowstat, ce, scale2, (__bf16*)dl_hi, (const __bf16*)ap_hi, (const __bf16*)ap_lo, (__bf16*)aps_hi, ap_cols,
              (int)(ap_inner >> 5));
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
extern "C" int tcar_ce_anchor_fold(int B, int N, int group_width, int ngroups, const float* stats, const float* lab_logit,
                                   const int32_t* label, float* rowstat, float* ce, float* scale2, void* dl_hi, int64_t inner,
                                   const void* ap_hi, const void* ap_lo, void* aps_hi, int ap_cols, int64_t ap_inner, void* stream) {
  return tcar_ce_anchor_fold_o(B, N, group_width, ngroups, stats, lab_logit, label, rowstat, ce, scale2, dl_hi, inner, ap_hi, ap_lo,
                               aps_hi, ap_cols, ap_inner, stream, nullptr);
}

int tcar_ce_anchor_apply_o(int B, int N, const float* rowstat, const int32_t* label, int lab_off, void* dl_hi, int64_t inner,
                           const void* ap_hi, const void* ap_lo, void* aps_hi, int ap_cols, int64_t ap_inner, float* scale2, void* stream) {
  if (B <= 0) return TCAR_OK;
  if (N <= 0 || !rowstat || !label || !dl_hi || (inner & 31) || inner < N || !ap_hi || !ap_lo || !aps_hi || ap_cols <= 0 ||
      (ap_cols & 3) || (ap_inner & 31) || ap_inner < ap_cols || !scale2 || ((uintptr_t)scale2 & 7))
    return TCAR_E_ARG;
  TCAR_LAUNCH(ce_anchor_apply_kernel, dim3((((B + 31) & ~31) + 3) / 4), dim3(256), 0, (hipStream_t)stream, B, N, (int)(inner >> 5), rowstat, label, lab_off,
              scale2, (__bf16*)dl_hi, (const __bf16*)ap_hi, (const __bf16*)ap_lo, (__bf16*)aps_hi, ap_cols, (int)(ap_inner >> 5));
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// second half of tcar_ce_finish on its own (catalog-sharded step: the row statistics come from the statistics exchange):
// plane[b, n] = e[b, n] * exp(m_g - rowstat[b].x) * rowstat[b].y - [n == label[b] - lab_off]; lab_window != 0: a label outside
// [lab_off, lab_off + N) belongs to another shard and nothing is subtracted
int tcar_ce_rescale_o(int B, int N, int group_width, int ngroups, const float* stats, const float* rowstat, const int32_t* label,
                      int lab_off, int lab_window, void* dl_hi, int64_t inner, void* stream, TcarOpt* o) {
  if (B <= 0) return TCAR_OK;
  if (N <= 0 || !stats || !label || !rowstat || !dl_hi || (inner & 31) || inner < N || ngroups <= 0 ||
      (group_width != 64 && group_width != 96) || (long)ngroups * group_width < N || ((uintptr_t)rowstat & 7))
    return TCAR_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int in32 = (int)(inner >> 5);
  const long nunits = (((long)B + 127) >> 7) * ((in32 + CE_KPT - 1) / CE_KPT) * 512;
  const unsigned grid = (unsigned)((nunits + 255) / 256);
  const TcarSignal sig = tcar_sig(o);
  if (group_width == 96)
    TCAR_LAUNCH(ce_rescale_kernel<96>, dim3(grid), dim3(256), 0, st, B, N, in32, ngroups, stats, rowstat, label, (__bf16*)dl_hi, nunits,
                lab_off, lab_window, sig);
  else
    TCAR_LAUNCH(ce_rescale_kernel<64>, dim3(grid), dim3(256), 0, st, B, N, in32, ngroups, stats, rowstat, label, (__bf16*)dl_hi, nunits,
                lab_off, lab_window, sig);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
extern "C" int tcar_ce_rescale(int B, int N, int group_width, int ngroups, const float* stats, const float* rowstat,
                               const int32_t* label, int lab_off, int lab_window, void* dl_hi, int64_t inner, void* stream) {
  return tcar_ce_rescale_o(B, N, group_width, ngroups, stats, rowstat, label, lab_off, lab_window, dl_hi, inner, stream, nullptr);
}

extern "C" int tcar_ce_shard_stats(int B, int ngroups, const float* stats, const float* lab_logit, const int32_t* label, int n0,
                                   int n_loc, float* out3, void* stream) {
  return tcar_ce_shard_stats_a(B, ngroups, stats, lab_logit, label, n0, n_loc, out3, 0, stream);
}
// anchored != 0: the pairs of the anchored epilogue -> out3[b] = (sum of the plane's rounded entries, sum of the exponentials, label's
// accumulator), to be combined by tcar_softmax_combine_anchored
int tcar_ce_shard_stats_a(int B, int ngroups, const float* stats, const float* lab_logit, const int32_t* label, int n0, int n_loc,
                          float* out3, int anchored, void* stream) {
  if (B <= 0) return TCAR_OK;
  if (ngroups <= 0 || !stats || !lab_logit || !label || !out3 || ((uintptr_t)stats & 7)) return TCAR_E_ARG;
  TCAR_LAUNCH(ce_shard_stats_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, B, ngroups, stats, lab_logit, label, n0,
              n_loc, out3, anchored);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_neg_term(const tcar_dims_t* d, int B, int K, const float* E, const int32_t* neg,
                             const float* attout, float weight, float* neg_fb, float* dattout, float* g_item,
                             const float* ce, float* loss, void* stream) {
  if (!d || B <= 0 || K <= 0) return TCAR_OK;
  if (!E || !neg || !attout || (d->ldh & 63) || d->ldh > 512 || (loss && !ce)) return TCAR_E_ARG;
  const int ek = 2 * d->ldh + 5 * d->ldt;
  const int grid = B;          // one workgroup per session
  if (d->ldh <= 256)
    TCAR_LAUNCH(neg_term_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, K, d->n_items, d->ldh, ek, E,
                       neg, attout, weight, neg_fb, dattout, g_item, ce, loss, (float*)nullptr, (long)ek, 0, TcarSignal{});
  else
    TCAR_LAUNCH(neg_term_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, K, d->n_items, d->ldh, ek, E,
                       neg, attout, weight, neg_fb, dattout, g_item, ce, loss, (float*)nullptr, (long)ek, 0, TcarSignal{});
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_neg_fwd(const tcar_dims_t* d, int B, int K, const float* E, const int32_t* neg, const float* attout,
                            float weight, float* neg_fb, float* coef, float* negpart, void* stream) {
  return tcar_neg_fwd_o(d, B, K, E, neg, attout, weight, neg_fb, coef, negpart, stream, nullptr);
}
// (flag-capable: negpart — what the main chain's slab reduce reads — leaves write-through when the launch carries a flag)
int tcar_neg_fwd_o(const tcar_dims_t* d, int B, int K, const float* E, const int32_t* neg, const float* attout, float weight,
                   float* neg_fb, float* coef, float* negpart, void* stream, TcarOpt* o) {
  if (!d || B <= 0 || K <= 0) return TCAR_OK;
  if (!E || !neg || !attout || !coef || !negpart || (d->ldh & 63) || d->ldh > 512) return TCAR_E_ARG;
  const int ek = 2 * d->ldh + 5 * d->ldt;
  const TcarSignal sg = tcar_sig(o);
  if (d->ldh <= 256)
    TCAR_LAUNCH(neg_term_kernel<1>, dim3(B), dim3(256), 0, (hipStream_t)stream, B, K, d->n_items, d->ldh, ek, E, neg, attout,
                weight, neg_fb, negpart, (float*)nullptr, (const float*)nullptr, (float*)nullptr, coef, (long)(2 * d->ldh), 1, sg);
  else
    TCAR_LAUNCH(neg_term_kernel<2>, dim3(B), dim3(256), 0, (hipStream_t)stream, B, K, d->n_items, d->ldh, ek, E, neg, attout,
                weight, neg_fb, negpart, (float*)nullptr, (const float*)nullptr, (float*)nullptr, coef, (long)(2 * d->ldh), 1, sg);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_neg_scatter(const tcar_dims_t* d, int B, int K, const int32_t* neg, const float* attout,
                                const float* coef, float* g_item, const float* neg_fb, const float* ce, float weight,
                                float* loss, void* stream) {
  if (!d || B <= 0 || K <= 0) return TCAR_OK;
  if (!neg || !attout || !coef || !g_item || (loss && (!ce || !neg_fb))) return TCAR_E_ARG;
  const int ek = 2 * d->ldh + 5 * d->ldt;
  const long waves = (long)B * K;
  TCAR_LAUNCH(neg_scatter_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, B, K, d->n_items,
              d->ldh, ek, neg, attout, coef, g_item, neg_fb, ce, weight, loss);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_splitk_reduce_dact(const float* slabs, int splitk, int M, int N, int64_t ld, const float* addend,
                                       int64_t ld_add, int n_add, const float* y, int64_t ldy, int act, float* out,
                                       float* bias_grad0, int split_col, float* bias_grad1, void* stream) {
  if (M <= 0 || N <= 0) return TCAR_OK;
  if ((N & 3) || (ld & 3) || !tcar_aligned16(slabs) || !tcar_aligned16(out) || splitk < 1 || (act && (!y || (ldy & 3))) ||
      (addend && ((ld_add & 3) || (n_add & 3) || !tcar_aligned16(addend))))
    return TCAR_E_ARG;
  TCAR_LAUNCH(reduce_dact_kernel, dim3((N + 63) / 64, (M + 15) / 16), dim3(256), 0, (hipStream_t)stream, slabs, splitk, M, N,
              (long)ld, addend, (long)ld_add, n_add, y, (long)ldy, act, out, bias_grad0, split_col, bias_grad1);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// tcar_splitk_reduce_dact for the one-hot form of dX (tcar_gemm_bf16_dx_onehot): slabs [splitk, M, ld] with ld >= ic + 160;
// out [M, ldo] gets all ic + 5 * 64 columns of d attout (through tanh' of y = attout), dP [M, 160] the summed one-hot columns.
// addend [M, ic] (the negative term's part) may be NULL; bias_grad0 / 1 as tcar_splitk_reduce_dact (NULL: order-fixed column sums
// elsewhere).  ldt = 64.
int tcar_reduce_dact_onehot_o(const float* slabs, int splitk, int M, int ic, int64_t ld, const float* addend, int64_t ld_add,
                              const float* y, int64_t ldy, const float* tclip, float* out, int64_t ldo, float* dP, float* bias_grad0,
                              float* bias_grad1, void* stream, TcarOpt* o) {
  if (M <= 0) return TCAR_OK;
  if (!slabs || splitk <= 0 || ic <= 0 || (ic & 63) || ld < ic + 160 || (ld & 3) || (ldy & 3) || (ldo & 3) || (addend && (ld_add & 3)) ||
      !tclip || !out || !dP || !tcar_aligned16(slabs) || (y && !tcar_aligned16(y)) || !tcar_aligned16(out) || !tcar_aligned16(tclip) ||
      (addend && !tcar_aligned16(addend)))
    return TCAR_E_ARG;
#ifdef TCAR_OBS1_DIAG
  if (getenv("TCAR_OBS1_LDS") && y && !bias_grad0 && !bias_grad1 && !(o && o->wait.flag)) {
    TCAR_LAUNCH(reduce_dact_onehot_lds_kernel, dim3(ic / 64 + 5, (M + 15) / 16), dim3(256), 0, (hipStream_t)stream, slabs, splitk, M, ic,
                (long)ld, addend, (long)ld_add, y, (long)ldy, tclip, out, (long)ldo, dP, tcar_sig(o));
    TCAR_CHECK_LAUNCH();
    return TCAR_OK;
  }
#endif
  TCAR_LAUNCH(reduce_dact_onehot_kernel, dim3(ic / 64 + 5, (M + 15) / 16), dim3(256), 0, (hipStream_t)stream, slabs, splitk, M, ic,
              (long)ld, addend, (long)ld_add, y, (long)ldy, tclip, out, (long)ldo, dP, bias_grad0, bias_grad1, tcar_sig(o),
              o ? o->wait : TcarWait{}, (o && o->rowfix) ? *o->rowfix : TcarRowFix{});
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
// ... of the anchored softmax form: scale2 [M, 2] = (1 / S_m, residual) as tcar_ce_anchor_fold leaves them, label [M], E fp32
// candidate rows [n_items, ldE], mwdhm [n_items, 5] (TcarRowFix); no bias column sums
extern "C" int tcar_reduce_dact_onehot_scaled(const float* slabs, int splitk, int M, int ic, int64_t ld, const float* addend,
                                              int64_t ld_add, const float* y, int64_t ldy, const float* tclip, float* out, int64_t ldo,
                                              float* dP, const float* scale2, const int32_t* label, const float* E, int64_t ldE,
                                              const int32_t* mwdhm, int n_items, void* stream) {
  if (!scale2 || !label || !E || !mwdhm || n_items <= 0 || ldE < ic || (ldE & 3) || !tcar_aligned16(E)) return TCAR_E_ARG;
  const TcarRowFix fix{scale2, label, E, (long)ldE, mwdhm, n_items};
  TcarOpt o;
  o.rowfix = &fix;
  return tcar_reduce_dact_onehot_o(slabs, splitk, M, ic, ld, addend, ld_add, y, ldy, tclip, out, ldo, dP, nullptr, nullptr, stream, &o);
}
extern "C" int tcar_reduce_dact_onehot(const float* slabs, int splitk, int M, int ic, int64_t ld, const float* addend, int64_t ld_add,
                                       const float* y, int64_t ldy, const float* tclip, float* out, int64_t ldo, float* dP,
                                       float* bias_grad0, float* bias_grad1, void* stream) {
  return tcar_reduce_dact_onehot_o(slabs, splitk, M, ic, ld, addend, ld_add, y, ldy, tclip, out, ldo, dP, bias_grad0, bias_grad1, stream,
                                   nullptr);
}

extern "C" int tcar_dact_colsum(int M, int ncol, int64_t ld, const float* y, float* dy, float* bias_grad, int act,
                                void* stream) {
  if (M <= 0 || ncol <= 0) return TCAR_OK;
  if (!y || !dy) return TCAR_E_ARG;
  int chunks = (M + 31) / 32;
  if (chunks > 16) chunks = 16;
  TCAR_LAUNCH(dact_colsum_kernel, dim3((ncol + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, M, ncol, (long)ld, y,
              dy, bias_grad, act);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_rank_topk(int B, int N, const float* logits, int64_t ld, const int32_t* label, int k,
                              int32_t* rank, int32_t* topk, void* stream) {
  if (B <= 0) return TCAR_OK;
  if (N <= 0 || k < 0 || !logits || !label || !rank || (k > 0 && !topk)) return TCAR_E_ARG;
  return tcar_eval_rows(B, N, logits, ld, label, k, rank, topk, nullptr, stream);
}

extern "C" int tcar_eval_rows(int B, int N, const float* logits, int64_t ld, const int32_t* label, int k, int32_t* rank,
                              int32_t* topk, float* ce, void* stream) {
  if (B <= 0) return TCAR_OK;
  if (N <= 0 || k < 0 || ld < N || !logits || !label || !rank || (k > 0 && !topk)) return TCAR_E_ARG;
  if ((ld & 3) == 0 && tcar_aligned16(logits) && ld <= 512L * 4 * 24) {
    if (ld <= 512L * 4 * 8)
      TCAR_LAUNCH((rank_topk_rows_kernel<512, 8>), dim3(B), dim3(512), 0, (hipStream_t)stream, N, logits, (long)ld, label, k,
                  rank, topk, ce);
    else
      TCAR_LAUNCH((rank_topk_rows_kernel<512, 24>), dim3(B), dim3(512), 0, (hipStream_t)stream, N, logits, (long)ld, label, k,
                  rank, topk, ce);
    TCAR_CHECK_LAUNCH();
    return TCAR_OK;
  }
  if (ce) return TCAR_E_ARG;        // the streaming fallback has no fused CE: call tcar_softmax_ce
  TCAR_LAUNCH(rank_topk_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, N, logits, (long)ld, label, k, rank, topk);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// ---- diversity metrics of the evaluation loop (model_combine.py:174-194,301-313) on the device ---------------------------
// getILD: ordered pairs (i != j) of the top-k list whose categories differ; getUnexp: (recommended, input click) pairs whose
// categories differ; resultItemDict: the set of recommended items.  One wave per session, lane = one recommended item
// (k <= 64): its category against the k - 1 others (cross-lane reads) and against the T input clicks.  The kernel leaves the
// INTEGER counts (the reference divides Python ints: score / (n (n - 1)), score / (n len(inSeq)) — the host divides the same
// ints in double precision, so the metrics are bit-exact) and marks seen[item] = 1 (a byte map: idempotent plain stores, and
// ranks can union it with a MAX all-reduce).  Entries < 0 of topk (a catalog shorter than k) are not in the list.
__global__ __launch_bounds__(256) void eval_diversity_kernel(int B, int T, int k, int n_items, const int32_t* __restrict__ topk,
                                                             const int32_t* __restrict__ seq, const int32_t* __restrict__ cat,
                                                             int32_t* __restrict__ ild_cnt, int32_t* __restrict__ unexp_cnt,
                                                             int32_t* __restrict__ n_rec, uint8_t* __restrict__ seen) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int item = lane < k ? topk[(long)b * k + lane] : -1;
  const bool valid = item >= 0 && item < n_items;
  const int c = valid ? cat[item] : 0;
  if (valid && seen) seen[item] = 1;
  int ild = 0;
  for (int j = 0; j < k; ++j) {
    const int cj = __shfl(c, j);
    const int vj = __shfl(valid ? 1 : 0, j);
    ild += (valid && vj && j != lane && c != cj) ? 1 : 0;
  }
  int un = 0;
  for (int t = 0; t < T; ++t) {
    // 1-based (sampler.py:68): category of item id - 1 (model_combine.py:192).  Ids outside [1, n_items] cannot reach this kernel
    // through the engine (upload() raises on them, as the reference's dict lookup would); a raw caller's out-of-range id is
    // CLAMPED to the nearest item — id 0 compares as item 0, not as Python's reverse_item[-1]
    const int id = seq[(long)b * T + t];
    const int ct = cat[clampi(id - 1, 0, n_items - 1)];
    un += (valid && c != ct) ? 1 : 0;
  }
  int nv = valid ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ild += __shfl_xor(ild, o);
    un += __shfl_xor(un, o);
    nv += __shfl_xor(nv, o);
  }
  if (lane == 0) {
    ild_cnt[b] = ild;
    unexp_cnt[b] = un;
    if (n_rec) n_rec[b] = nv;
  }
}

extern "C" int tcar_eval_diversity(int B, int T, int k, int n_items, const int32_t* topk, const int32_t* seq, const int32_t* cat,
                                   int32_t* ild_cnt, int32_t* unexp_cnt, int32_t* n_rec, uint8_t* seen, void* stream) {
  if (B <= 0) return TCAR_OK;
  if (T <= 0 || k <= 0 || k > 64 || n_items <= 0 || !topk || !seq || !cat || !ild_cnt || !unexp_cnt) return TCAR_E_ARG;
  TCAR_LAUNCH(eval_diversity_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, B, T, k, n_items, topk, seq,
              cat, ild_cnt, unexp_cnt, n_rec, seen);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
