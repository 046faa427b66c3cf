// Session attention pools (forward + backward) for gfx950.
//
// Reference: count_alpha_m / multi_attention_layer (modules.py:103-152), count_alpha_s / single_attention_layer
// (modules.py:72-101), normalizer (util.py:92-100: exp(x) / (sum exp(x) + 1e-9), NO max subtraction).
//
//   e1[t] = sum_j sigmoid(pre1[t,j]) * w_res1[j]         alpha1 = expnorm_t(e1)      modules.py:132-135
//   e2[t] = X_ic[t] . q                                  alpha2 = expnorm_t(e2)      modules.py:140-141
//   e3[t] = sum_j sigmoid(pre2[t,j]) * w_res2[j]         alpha_t = expnorm_t(e3)     modules.py:97-100
//   pooled_ic = sum_t (alpha1+alpha2)[t] X_ic[t]         pooled_t = sum_t alpha_t[t] X_pt[t]   (:116-117, :82-83)
//
// A batch holds sessions of ONE length T <= 40 (sampler.py:40-49): no padding, no mask.  One 64-lane wave owns
// one session; the T scores live one per lane, the exp-normalisation is a wave reduction, and the weighted sum
// re-reads the T rows (L2 hits: they were just read for the scores).  Everything stays in registers — no LDS,
// no cross-wave traffic.  This is the only softmax-attention on the executed graph (a single query per session),
// far too small for MFMA: it is bound by the row reads.
#include <type_traits>
#include "tcar_common.h"

namespace {

struct PoolArgs {
  int B, T, H, ldh, ldt;
  const float* x_icp; const float* x_pt; const float* pre1; const float* pre2; const float* q;
  const float* w1; const float* w2; const float* alpha_in; const float* dpooled;
  float* pooled; float* alpha;
  float* dx_icp; float* dx_pt; float* dq; float* dpre1; float* dpre2; float* g_w1; float* g_w2;
  // split-K slabs (tcar_attn_pool_fwd_slabs / tcar_attn_pool_bwd_slabs): pre1 / pre2 arrive as n1 / n2 partial products of
  // stride pre_stride floats — summed here in slab order, the sums written to pre1_out / pre2_out (the backward pass re-reads
  // them); dpooled arrives as nd_ic (item | content columns) / nd_pt (time columns) slabs of stride dp_stride
  int n1, n2, nd_ic, nd_pt; long pre_stride, dp_stride; float* pre1_out; float* pre2_out;
  float* g_qb;     // optional: bias gradient of query_trans2; dq then leaves multiplied by tanh'(q) (modules.py:139 backward)
  float* gw_rows;  // optional [B, 2 * ldh]: the per-session d w_res1 | d w_res2 rows are WRITTEN here (and dq leaves through
                   // tanh') instead of any atomic sum: tcar_colsum_det adds the columns up in a fixed order
  TcarSignal sig;  // backward, optional completion flag: dq (what the other stream's click-query backward reads) leaves write-through
  TcarWait wait_q; // forward, optional: the click query q comes from another stream behind a completion flag — every wave waits
                   // for it itself, after the part of its work that does not need q (the slab fold, alpha1 / alpha_t scores)
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float4 sig4(float4 v) {
  return make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
}
// zero the lanes of a float4 whose column index is >= H (padding columns j in [H, ldh))
__device__ __forceinline__ float4 mask4(float4 v, int col, int H) {
  return make_float4(col + 0 < H ? v.x : 0.f, col + 1 < H ? v.y : 0.f, col + 2 < H ? v.z : 0.f, col + 3 < H ? v.w : 0.f);
}

template <int NCH, bool SLABS = true>  // NCH = ceil(ldh/256); SLABS: the projections arrive as split-K slabs (else: finished rows)
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(const PoolArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const int T = a.T, H = a.H, ldh = a.ldh, ldt = a.ldt;
  const int ic = 2 * ldh, pt = 5 * ldt, ek = ic + pt;
  const int BT = a.B * T;
  const int ptl = pt >> 2;   // float4 lanes of a publish-time row (80 for ldt = 64)

  // per-lane constants
  float4 w1[NCH], w2[NCH], qa[NCH], qb[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
    const bool ok = col < ldh;
    w1[c] = ok ? ld4(a.w1 + col) : zero4();
    w2[c] = ok ? ld4(a.w2 + col) : zero4();
    qa[c] = zero4(); qb[c] = zero4();
  }
  float e1 = 0.f, e2 = 0.f, e3 = 0.f;     // lane t keeps the scores of position t
  const bool late_q = a.wait_q.flag != nullptr;      // (kernel-uniform) q arrives behind a flag: its scores in a second loop
  if (!late_q) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < ldh) { qa[c] = ld4(a.q + (long)b * ic + col); qb[c] = ld4(a.q + (long)b * ic + ldh + col); }
    }
  }
  // Round 4: every load of a pair of positions — up to MS split-K slabs of each projection, the item | content row, the publish-time
  // row — is ISSUED before the first is used (the slab fold was a loop of 7 + 5 dependent round trips per position: 10 us alone and
  // 24-38 us beside the Adam rest pass, whose traffic lengthens each trip); the rows of the first TC positions stay in registers for
  // the weighted sums below.  The sums keep their order: slabs in slab order, positions in position order.
  // (Round 5: finished rows — the un-split projections of the long buckets — need one load per projection and position: FOUR positions
  //  per trip there, and every loop over the positions behind the cached ones below walks them four at a time, loads first.)
  constexpr int MS = SLABS ? 8 : 1, TU = (NCH == 1) ? (SLABS ? 2 : 4) : 1, TC = (NCH == 1) ? 4 : 0;
  float4 xa[TC > 0 ? TC : 1][NCH], xb[TC > 0 ? TC : 1][NCH], xp[TC > 0 ? TC : 1][2];
  const int n1 = (SLABS && a.pre1_out) ? a.n1 : 1, n2 = (SLABS && a.pre1_out) ? a.n2 : 1;
  for (int t0 = 0; t0 < T; t0 += TU) {
    float4 t1[TU][NCH][MS], t2[TU][NCH][MS], ya[TU][NCH], yb[TU][NCH], yp[TU][2];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int t = t0 + u;
      const long row = (long)b * T + t;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        const bool ok = t < T && col < ldh;
#pragma unroll
        for (int sl = 0; sl < MS; ++sl) {
          t1[u][c][sl] = (ok && sl < n1) ? ld4(a.pre1 + sl * a.pre_stride + row * ldh + col) : zero4();
          t2[u][c][sl] = (ok && sl < n2) ? ld4(a.pre2 + sl * a.pre_stride + row * ldh + col) : zero4();
        }
        ya[u][c] = ok ? ld4(a.x_icp + row * ic + col) : zero4();
        yb[u][c] = ok ? ld4(a.x_icp + row * ic + ldh + col) : zero4();
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int l4 = c * 64 + lane;
        yp[u][c] = (TC > 0 && t < T && l4 < ptl) ? ld4(a.x_pt + row * pt + l4 * 4) : zero4();
      }
    }
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int t = t0 + u;
      if (t >= T) break;
      const long row = (long)b * T + t;
      float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < ldh) {
          float4 p1 = t1[u][c][0], p2 = t2[u][c][0];
          if (SLABS && a.pre1_out) {       // split-K partial products, folded in slab order
#pragma unroll
            for (int sl = 1; sl < MS; ++sl) if (sl < n1) p1 = add4(p1, t1[u][c][sl]);
            for (int sl = MS; sl < n1; ++sl) p1 = add4(p1, ld4(a.pre1 + sl * a.pre_stride + row * ldh + col));
#pragma unroll
            for (int sl = 1; sl < MS; ++sl) if (sl < n2) p2 = add4(p2, t2[u][c][sl]);
            for (int sl = MS; sl < n2; ++sl) p2 = add4(p2, ld4(a.pre2 + sl * a.pre_stride + row * ldh + col));
            st4(a.pre1_out + row * ldh + col, p1);
            st4(a.pre2_out + row * ldh + col, p2);
          }
          s1 += dot4(mask4(sig4(p1), col, H), w1[c]);
          s3 += dot4(mask4(sig4(p2), col, H), w2[c]);
          if (!late_q) s2 += dot4(ya[u][c], qa[c]) + dot4(yb[u][c], qb[c]);
        }
      }
      s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
      if (lane == t) { e1 = s1; e2 = s2; e3 = s3; }
      if constexpr (TC > 0) {
#pragma unroll
        for (int k = 0; k < TC; ++k)      // (t is a runtime value: a compare chain instead of a dynamically indexed register array)
          if (k == t) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) { xa[k][c] = ya[u][c]; xb[k][c] = yb[u][c]; }
            xp[k][0] = yp[u][0]; xp[k][1] = yp[u][1];
          }
      }
    }
  }
  if (late_q) {       // alpha2 scores X_ic . q (modules.py:140-141), behind the producer's flag: same sums in the same order
    tcar_wave_wait(a.wait_q);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < ldh) { qa[c] = ld4_sc1(a.q + (long)b * ic + col); qb[c] = ld4_sc1(a.q + (long)b * ic + ldh + col); }
    }
#pragma unroll
    for (int t = 0; t < TC; ++t) {          // (rows still in registers)
      if (t >= T) break;
      float s2 = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < ldh) s2 += dot4(xa[t][c], qa[c]) + dot4(xb[t][c], qb[c]);
      }
      s2 = wave_sum(s2);
      if (lane == t) e2 = s2;
    }
    for (int t0 = TC; t0 < T; t0 += 4) {          // four positions per trip, loads first
      float4 ra[4][NCH], rb[4][NCH];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long row = (long)b * T + t0 + u;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int col = c * 256 + lane * 4;
          const bool ok = t0 + u < T && col < ldh;
          ra[u][c] = ok ? ld4(a.x_icp + row * ic + col) : zero4();
          rb[u][c] = ok ? ld4(a.x_icp + row * ic + ldh + col) : zero4();
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u;
        if (t >= T) break;
        float s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int col = c * 256 + lane * 4;
          if (col < ldh) s2 += dot4(ra[u][c], qa[c]) + dot4(rb[u][c], qb[c]);
        }
        s2 = wave_sum(s2);
        if (lane == t) e2 = s2;
      }
    }
  }
  const bool on = lane < T;
  const float x1 = on ? expf(e1) : 0.f, x2 = on ? expf(e2) : 0.f, x3 = on ? expf(e3) : 0.f;
  const float a1 = x1 / (wave_sum(x1) + 1e-9f);
  const float a2 = x2 / (wave_sum(x2) + 1e-9f);
  const float a3 = x3 / (wave_sum(x3) + 1e-9f);
  if (on) {
    a.alpha[(long)b * T + lane] = a1;
    a.alpha[(long)BT + (long)b * T + lane] = a2;
    a.alpha[2L * BT + (long)b * T + lane] = a3;
  }
  const float a12 = a1 + a2;
  float4 pa[NCH], pb[NCH];
  float4 pp[2];                // publish-time row: up to 2 float4 per lane (pt <= 512 floats)
#pragma unroll
  for (int c = 0; c < NCH; ++c) { pa[c] = zero4(); pb[c] = zero4(); }
  pp[0] = zero4(); pp[1] = zero4();
#pragma unroll
  for (int t = 0; t < TC; ++t) {             // positions whose rows are still in registers
    if (t >= T) break;
    const float wt = __shfl(a12, t), wt3 = __shfl(a3, t);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < ldh) { pa[c] = fma4(xa[t][c], wt, pa[c]); pb[c] = fma4(xb[t][c], wt, pb[c]); }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int l4 = c * 64 + lane;
      if (l4 < ptl) pp[c] = fma4(xp[t][c], wt3, pp[c]);
    }
  }
  for (int t0 = TC; t0 < T; t0 += 4) {            // four positions per trip, loads first; the sums stay in position order
    float4 ra[4][NCH], rb[4][NCH], rp[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long row = (long)b * T + t0 + u;
      const bool in = t0 + u < T;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        const bool ok = in && col < ldh;
        ra[u][c] = ok ? ld4(a.x_icp + row * ic + col) : zero4();
        rb[u][c] = ok ? ld4(a.x_icp + row * ic + ldh + col) : zero4();
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int l4 = c * 64 + lane;
        rp[u][c] = (in && l4 < ptl) ? ld4(a.x_pt + row * pt + l4 * 4) : zero4();
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u;
      if (t >= T) break;
      const float wt = __shfl(a12, t), wt3 = __shfl(a3, t);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < ldh) { pa[c] = fma4(ra[u][c], wt, pa[c]); pb[c] = fma4(rb[u][c], wt, pb[c]); }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int l4 = c * 64 + lane;
        if (l4 < ptl) pp[c] = fma4(rp[u][c], wt3, pp[c]);
      }
    }
  }
  float* o = a.pooled + (long)b * ek;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
    if (col < ldh) { st4(o + col, pa[c]); st4(o + ldh + col, pb[c]); }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int l4 = c * 64 + lane;
    if (l4 < ptl) st4(o + ic + l4 * 4, pp[c]);
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void attn_pool_bwd_kernel(const PoolArgs a) {
  __shared__ float gw_lds[4 * 512];     // per-workgroup partial sums of d w_res1 | d w_res2 | d q-bias (2 * ldh)
  const int tid = threadIdx.x, lane = tid & 63;
  const int b = blockIdx.x * 4 + (tid >> 6);
  const int T = a.T, H = a.H, ldh = a.ldh, ldt = a.ldt;
  const int ic = 2 * ldh, pt = 5 * ldt, ek = ic + pt;
  const int BT = a.B * T;
  const int ptl = pt >> 2;
  const int nlds = a.gw_rows ? 0 : (a.g_qb ? 4 * ldh : 2 * ldh);
  for (int i = tid; i < nlds; i += 256) gw_lds[i] = 0.f;
  __syncthreads();
  if (b < a.B) {
    float4 w1[NCH], w2[NCH], qa[NCH], qb[NCH], da[NCH], db[NCH];
    float4 dp[2];
    const float* dpo = a.dpooled + (long)b * ek;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      const bool ok = col < ldh;
      w1[c] = ok ? ld4(a.w1 + col) : zero4();
      w2[c] = ok ? ld4(a.w2 + col) : zero4();
      qa[c] = ok ? ld4(a.q + (long)b * ic + col) : zero4();
      qb[c] = ok ? ld4(a.q + (long)b * ic + ldh + col) : zero4();
      // split-K partial products of the output transform's input gradient, folded in slab order; the first MS slabs' loads are
      // issued together (round 4: the fold was a chain of dependent round trips)
      constexpr int MS = 8;
      float4 ta[MS], tb[MS];
#pragma unroll
      for (int sl = 0; sl < MS; ++sl) {
        const bool on = ok && (sl == 0 || sl < a.nd_ic);
        ta[sl] = on ? ld4(dpo + sl * a.dp_stride + col) : zero4();
        tb[sl] = on ? ld4(dpo + sl * a.dp_stride + ldh + col) : zero4();
      }
      da[c] = ta[0]; db[c] = tb[0];
#pragma unroll
      for (int sl = 1; sl < MS; ++sl)
        if (ok && sl < a.nd_ic) { da[c] = add4(da[c], ta[sl]); db[c] = add4(db[c], tb[sl]); }
      for (int sl = MS; sl < a.nd_ic; ++sl) {
        if (ok) { da[c] = add4(da[c], ld4(dpo + sl * a.dp_stride + col)); db[c] = add4(db[c], ld4(dpo + sl * a.dp_stride + ldh + col)); }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int l4 = c * 64 + lane;
      constexpr int MS = 8;
      float4 tp[MS];
#pragma unroll
      for (int sl = 0; sl < MS; ++sl) tp[sl] = (l4 < ptl && (sl == 0 || sl < a.nd_pt)) ? ld4(dpo + sl * a.dp_stride + ic + l4 * 4) : zero4();
      dp[c] = tp[0];
#pragma unroll
      for (int sl = 1; sl < MS; ++sl)
        if (l4 < ptl && sl < a.nd_pt) dp[c] = add4(dp[c], tp[sl]);
      for (int sl = MS; sl < a.nd_pt; ++sl)
        if (l4 < ptl) dp[c] = add4(dp[c], ld4(dpo + sl * a.dp_stride + ic + l4 * 4));
    }
    const bool on = lane < T;
    const float a1 = on ? a.alpha_in[(long)b * T + lane] : 0.f;
    const float a2 = on ? a.alpha_in[(long)BT + (long)b * T + lane] : 0.f;
    const float a3 = on ? a.alpha_in[2L * BT + (long)b * T + lane] : 0.f;
    // the rows of the first TC positions — item | content, publish time, both pre-activations — are fetched ONCE, all loads issued
    // together, and serve both passes (round 4: each pass walked the positions with dependent loads)
    constexpr int TC = (NCH == 1) ? 4 : 0;
    float4 xa[TC > 0 ? TC : 1][NCH], xb[TC > 0 ? TC : 1][NCH], xp[TC > 0 ? TC : 1][2], r1[TC > 0 ? TC : 1][NCH], r2[TC > 0 ? TC : 1][NCH];
#pragma unroll
    for (int t = 0; t < TC; ++t) {
      const long row = (long)b * T + t;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        const bool ok = t < T && col < ldh;
        xa[t][c] = ok ? ld4(a.x_icp + row * ic + col) : zero4();
        xb[t][c] = ok ? ld4(a.x_icp + row * ic + ldh + col) : zero4();
        r1[t][c] = ok ? ld4(a.pre1 + row * ldh + col) : zero4();
        r2[t][c] = ok ? ld4(a.pre2 + row * ldh + col) : zero4();
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int l4 = c * 64 + lane;
        xp[t][c] = (t < T && l4 < ptl) ? ld4(a.x_pt + row * pt + l4 * 4) : zero4();
      }
    }
    // pass A: d alpha[t] = dpooled . X[t]   (alpha1 and alpha2 share it: alpha = alpha1 + alpha2)
    float dal = 0.f, dal3 = 0.f;
#pragma unroll
    for (int t = 0; t < TC; ++t) {
      if (t >= T) break;
      float s = 0.f, s3 = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < ldh) s += dot4(xa[t][c], da[c]) + dot4(xb[t][c], db[c]);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int l4 = c * 64 + lane;
        if (l4 < ptl) s3 += dot4(xp[t][c], dp[c]);
      }
      s = wave_sum(s); s3 = wave_sum(s3);
      if (lane == t) { dal = s; dal3 = s3; }
    }
    for (int t0 = TC; t0 < T; t0 += 4) {          // (round 5) four positions per trip, loads first
      float4 ra[4][NCH], rb[4][NCH], rp[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long row = (long)b * T + t0 + u;
        const bool in = t0 + u < T;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int col = c * 256 + lane * 4;
          const bool ok = in && col < ldh;
          ra[u][c] = ok ? ld4(a.x_icp + row * ic + col) : zero4();
          rb[u][c] = ok ? ld4(a.x_icp + row * ic + ldh + col) : zero4();
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int l4 = c * 64 + lane;
          rp[u][c] = (in && l4 < ptl) ? ld4(a.x_pt + row * pt + l4 * 4) : zero4();
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u;
        if (t >= T) break;
        float s = 0.f, s3 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int col = c * 256 + lane * 4;
          if (col < ldh) s += dot4(ra[u][c], da[c]) + dot4(rb[u][c], db[c]);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int l4 = c * 64 + lane;
          if (l4 < ptl) s3 += dot4(rp[u][c], dp[c]);
        }
        s = wave_sum(s); s3 = wave_sum(s3);
        if (lane == t) { dal = s; dal3 = s3; }
      }
    }
    // exp-normaliser backward: de = alpha * (dalpha - sum_s dalpha_s alpha_s)   (epsilon included exactly)
    const float de1 = a1 * (dal - wave_sum(dal * a1));
    const float de2 = a2 * (dal - wave_sum(dal * a2));
    const float de3 = a3 * (dal3 - wave_sum(dal3 * a3));
    const float a12 = a1 + a2;
    // pass B
    float4 dqa[NCH], dqb[NCH], gw1[NCH], gw2[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { dqa[c] = zero4(); dqb[c] = zero4(); gw1[c] = zero4(); gw2[c] = zero4(); }
    // (the rows of position t: from the register cache, or pre-loaded by the caller — pr_* — two positions at a time)
    auto pass_b = [&](int t, auto cached_, const float4 (&pr_xa)[NCH], const float4 (&pr_xb)[NCH], const float4 (&pr_r1)[NCH],
                      const float4 (&pr_r2)[NCH]) __attribute__((always_inline)) {
      constexpr bool CACHED = decltype(cached_)::value;
      const int tc = CACHED ? t : 0;
      const long row = (long)b * T + t;
      const float wt = __shfl(a12, t), wt3 = __shfl(a3, t);
      const float g1 = __shfl(de1, t), g2 = __shfl(de2, t), g3 = __shfl(de3, t);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < ldh) {
          const float4 xa_ = CACHED ? xa[tc][c] : pr_xa[c];
          const float4 xb_ = CACHED ? xb[tc][c] : pr_xb[c];
          st4(a.dx_icp + row * ic + col, fma4(qa[c], g2, scale4(da[c], wt)));
          st4(a.dx_icp + row * ic + ldh + col, fma4(qb[c], g2, scale4(db[c], wt)));
          dqa[c] = fma4(xa_, g2, dqa[c]);
          dqb[c] = fma4(xb_, g2, dqb[c]);
          const float4 s1 = mask4(sig4(CACHED ? r1[tc][c] : pr_r1[c]), col, H);
          const float4 s2 = mask4(sig4(CACHED ? r2[tc][c] : pr_r2[c]), col, H);
          gw1[c] = fma4(s1, g1, gw1[c]);
          gw2[c] = fma4(s2, g3, gw2[c]);
          // dpre = de * w * sig * (1 - sig)   (0 in padding columns: w = 0 there and sig is masked)
          const float4 dp1 = make_float4(g1 * w1[c].x * s1.x * (1.f - s1.x), g1 * w1[c].y * s1.y * (1.f - s1.y),
                                         g1 * w1[c].z * s1.z * (1.f - s1.z), g1 * w1[c].w * s1.w * (1.f - s1.w));
          const float4 dp2 = make_float4(g3 * w2[c].x * s2.x * (1.f - s2.x), g3 * w2[c].y * s2.y * (1.f - s2.y),
                                         g3 * w2[c].z * s2.z * (1.f - s2.z), g3 * w2[c].w * s2.w * (1.f - s2.w));
          // (a launch that carries a completion flag stores what the flag's consumers read write-through: dq below for the fused
          //  click-query backward, dpre1 / dpre2 for the weight gradients forked off this launch — tcar_common.h)
          if (a.sig.cnt) { st4_sc1(a.dpre1 + row * ldh + col, dp1); st4_sc1(a.dpre2 + row * ldh + col, dp2); }
          else { st4(a.dpre1 + row * ldh + col, dp1); st4(a.dpre2 + row * ldh + col, dp2); }
        }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int l4 = c * 64 + lane;
        if (l4 < ptl) st4(a.dx_pt + row * pt + l4 * 4, scale4(dp[c], wt3));
      }
    };
    {
      float4 none[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) none[c] = zero4();
#pragma unroll
      for (int t = 0; t < TC; ++t) {
        if (t >= T) break;
        pass_b(t, std::true_type{}, none, none, none, none);
      }
    }
    for (int t0 = TC; t0 < T; t0 += 2) {          // (round 5) two positions per trip, their four rows loaded first
      float4 ra[2][NCH], rb[2][NCH], q1[2][NCH], q2[2][NCH];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long row = (long)b * T + t0 + u;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int col = c * 256 + lane * 4;
          const bool ok = t0 + u < T && col < ldh;
          ra[u][c] = ok ? ld4(a.x_icp + row * ic + col) : zero4();
          rb[u][c] = ok ? ld4(a.x_icp + row * ic + ldh + col) : zero4();
          q1[u][c] = ok ? ld4(a.pre1 + row * ldh + col) : zero4();
          q2[u][c] = ok ? ld4(a.pre2 + row * ldh + col) : zero4();
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (t0 + u < T) pass_b(t0 + u, std::false_type{}, ra[u], rb[u], q1[u], q2[u]);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < ldh) {
        if (a.g_qb || a.gw_rows) {   // q = tanh(.): the gradient leaves through tanh' = 1 - q^2, and its column sums are the bias gradient
          dqa[c] = make_float4(dqa[c].x * (1.f - qa[c].x * qa[c].x), dqa[c].y * (1.f - qa[c].y * qa[c].y),
                               dqa[c].z * (1.f - qa[c].z * qa[c].z), dqa[c].w * (1.f - qa[c].w * qa[c].w));
          dqb[c] = make_float4(dqb[c].x * (1.f - qb[c].x * qb[c].x), dqb[c].y * (1.f - qb[c].y * qb[c].y),
                               dqb[c].z * (1.f - qb[c].z * qb[c].z), dqb[c].w * (1.f - qb[c].w * qb[c].w));
          if (!a.gw_rows) {
            atomic_add4(gw_lds + 2 * ldh + col, dqa[c]);
            atomic_add4(gw_lds + 3 * ldh + col, dqb[c]);
          }
        }
        if (a.sig.cnt) { st4_sc1(a.dq + (long)b * ic + col, dqa[c]); st4_sc1(a.dq + (long)b * ic + ldh + col, dqb[c]); }
        else { st4(a.dq + (long)b * ic + col, dqa[c]); st4(a.dq + (long)b * ic + ldh + col, dqb[c]); }
        if (a.gw_rows) {
          st4(a.gw_rows + (long)b * ic + col, gw1[c]);
          st4(a.gw_rows + (long)b * ic + ldh + col, gw2[c]);
        } else {
          atomic_add4(gw_lds + col, gw1[c]);
          atomic_add4(gw_lds + ldh + col, gw2[c]);
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < nlds; i += 256) {
    const float v = gw_lds[i];
    if (v != 0.f) atomicAdd((i < ldh ? a.g_w1 + i : i < 2 * ldh ? a.g_w2 + (i - ldh) : a.g_qb + (i - 2 * ldh)), v);
  }
  tcar_signal_done(a.sig);
}

}  // namespace

extern "C" int tcar_attn_pool_fwd(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt,
                                  const float* pre1, const float* pre2, const float* q, const float* w_res1,
                                  const float* w_res2, float* pooled, float* alpha, void* stream) {
  if (!d || B <= 0 || T <= 0 || T > TCAR_POS_VOCAB || (d->ldh & 63) || d->ldh > 512 || 5 * d->ldt > 512) return TCAR_E_ARG;
  PoolArgs a{};
  a.B = B; a.T = T; a.H = d->H; a.ldh = d->ldh; a.ldt = d->ldt;
  a.x_icp = x_icp; a.x_pt = x_pt; a.pre1 = pre1; a.pre2 = pre2; a.q = q; a.w1 = w_res1; a.w2 = w_res2;
  a.pooled = pooled; a.alpha = alpha;
  const int grid = (B + 3) / 4;
  if (d->ldh <= 256) TCAR_LAUNCH((attn_pool_fwd_kernel<1, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else TCAR_LAUNCH((attn_pool_fwd_kernel<2, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_attn_pool_fwd_slabs(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt,
                                        const float* pre1_slabs, int n1, const float* pre2_slabs, int n2, int64_t slab_stride,
                                        float* pre1, float* pre2, const float* q, const float* w_res1, const float* w_res2,
                                        float* pooled, float* alpha, void* stream) {
  return tcar_attn_pool_fwd_slabs_w(d, B, T, x_icp, x_pt, pre1_slabs, n1, pre2_slabs, n2, slab_stride, pre1, pre2, q, w_res1, w_res2,
                                    pooled, alpha, stream, TcarWait{});
}
// wait_q: q is produced on another stream behind a completion flag; the kernel waits for it itself (tcar_wave_wait)
int tcar_attn_pool_fwd_slabs_w(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt, const float* pre1_slabs,
                               int n1, const float* pre2_slabs, int n2, int64_t slab_stride, float* pre1, float* pre2, const float* q,
                               const float* w_res1, const float* w_res2, float* pooled, float* alpha, void* stream,
                               const TcarWait& wait_q) {
  if (!d || B <= 0 || T <= 0 || T > TCAR_POS_VOCAB || (d->ldh & 63) || d->ldh > 512 || 5 * d->ldt > 512) return TCAR_E_ARG;
  if (n1 < 1 || n2 < 1 || !pre1 || !pre2 || slab_stride < (int64_t)B * T * d->ldh) return TCAR_E_ARG;
  PoolArgs a{};
  a.wait_q = wait_q;
  a.B = B; a.T = T; a.H = d->H; a.ldh = d->ldh; a.ldt = d->ldt;
  a.x_icp = x_icp; a.x_pt = x_pt; a.pre1 = pre1_slabs; a.pre2 = pre2_slabs; a.q = q; a.w1 = w_res1; a.w2 = w_res2;
  a.n1 = n1; a.n2 = n2; a.pre_stride = slab_stride; a.pre1_out = pre1; a.pre2_out = pre2;
  a.pooled = pooled; a.alpha = alpha;
  const int grid = (B + 3) / 4;
  if (d->ldh <= 256) TCAR_LAUNCH(attn_pool_fwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else TCAR_LAUNCH(attn_pool_fwd_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_attn_pool_bwd(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt,
                                  const float* pre1, const float* pre2, const float* q, const float* w_res1,
                                  const float* w_res2, const float* alpha, const float* dpooled, float* dx_icp,
                                  float* dx_pt, float* dq, float* dpre1, float* dpre2, float* g_wres1,
                                  float* g_wres2, void* stream) {
  return tcar_attn_pool_bwd_q(d, B, T, x_icp, x_pt, pre1, pre2, q, w_res1, w_res2, alpha, dpooled, dx_icp, dx_pt, dq, dpre1,
                              dpre2, g_wres1, g_wres2, nullptr, stream);
}

extern "C" int tcar_attn_pool_bwd_q(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt,
                                    const float* pre1, const float* pre2, const float* q, const float* w_res1,
                                    const float* w_res2, const float* alpha, const float* dpooled, float* dx_icp,
                                    float* dx_pt, float* dq, float* dpre1, float* dpre2, float* g_wres1,
                                    float* g_wres2, float* g_qbias, void* stream) {
  if (!d || B <= 0 || T <= 0 || T > TCAR_POS_VOCAB || (d->ldh & 63) || d->ldh > 512 || 5 * d->ldt > 512) return TCAR_E_ARG;
  PoolArgs a{};
  a.B = B; a.T = T; a.H = d->H; a.ldh = d->ldh; a.ldt = d->ldt;
  a.x_icp = x_icp; a.x_pt = x_pt; a.pre1 = pre1; a.pre2 = pre2; a.q = q; a.w1 = w_res1; a.w2 = w_res2;
  a.alpha_in = alpha; a.dpooled = dpooled;
  a.dx_icp = dx_icp; a.dx_pt = dx_pt; a.dq = dq; a.dpre1 = dpre1; a.dpre2 = dpre2; a.g_w1 = g_wres1; a.g_w2 = g_wres2;
  a.g_qb = g_qbias;
  const int grid = (B + 3) / 4;
  if (d->ldh <= 256) TCAR_LAUNCH(attn_pool_bwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else TCAR_LAUNCH(attn_pool_bwd_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// Order-fixed form: dq leaves through tanh'(q) as in tcar_attn_pool_bwd_q, but nothing is summed atomically — the per-session
// rows d w_res1 | d w_res2 go to gw_rows [B, 2 * ldh]; tcar_colsum_det adds up their columns (and those of dq: the bias gradient
// of query_trans2) in a fixed order.
extern "C" int tcar_attn_pool_bwd_det(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt,
                                      const float* pre1, const float* pre2, const float* q, const float* w_res1,
                                      const float* w_res2, const float* alpha, const float* dpooled, float* dx_icp,
                                      float* dx_pt, float* dq, float* dpre1, float* dpre2, float* gw_rows, void* stream) {
  return tcar_attn_pool_bwd_slabs(d, B, T, x_icp, x_pt, pre1, pre2, q, w_res1, w_res2, alpha, dpooled, 1, 1, 0, dx_icp, dx_pt, dq,
                                  dpre1, dpre2, gw_rows, stream);
}

extern "C" int tcar_attn_pool_bwd_slabs(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt,
                                        const float* pre1, const float* pre2, const float* q, const float* w_res1,
                                        const float* w_res2, const float* alpha, const float* dpooled, int nd_ic, int nd_pt,
                                        int64_t dp_stride, float* dx_icp, float* dx_pt, float* dq, float* dpre1, float* dpre2,
                                        float* gw_rows, void* stream) {
  return tcar_attn_pool_bwd_slabs_o(d, B, T, x_icp, x_pt, pre1, pre2, q, w_res1, w_res2, alpha, dpooled, nd_ic, nd_pt, dp_stride, dx_icp,
                                    dx_pt, dq, dpre1, dpre2, gw_rows, stream, nullptr);
}
// (flag-capable: dq leaves write-through when the launch carries a flag — the click-query backward on another stream reads it)
int tcar_attn_pool_bwd_slabs_o(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt, const float* pre1,
                               const float* pre2, const float* q, const float* w_res1, const float* w_res2, const float* alpha,
                               const float* dpooled, int nd_ic, int nd_pt, int64_t dp_stride, float* dx_icp, float* dx_pt, float* dq,
                               float* dpre1, float* dpre2, float* gw_rows, void* stream, TcarOpt* o) {
  if (!d || B <= 0 || T <= 0 || T > TCAR_POS_VOCAB || (d->ldh & 63) || d->ldh > 512 || 5 * d->ldt > 512 || !gw_rows) return TCAR_E_ARG;
  if (nd_ic < 1 || nd_pt < 1) return TCAR_E_ARG;
  PoolArgs a{};
  a.sig = tcar_sig(o);
  a.nd_ic = nd_ic; a.nd_pt = nd_pt; a.dp_stride = dp_stride;
  a.B = B; a.T = T; a.H = d->H; a.ldh = d->ldh; a.ldt = d->ldt;
  a.x_icp = x_icp; a.x_pt = x_pt; a.pre1 = pre1; a.pre2 = pre2; a.q = q; a.w1 = w_res1; a.w2 = w_res2;
  a.alpha_in = alpha; a.dpooled = dpooled;
  a.dx_icp = dx_icp; a.dx_pt = dx_pt; a.dq = dq; a.dpre1 = dpre1; a.dpre2 = dpre2; a.gw_rows = gw_rows;
  const int grid = (B + 3) / 4;
  if (d->ldh <= 256) TCAR_LAUNCH(attn_pool_bwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else TCAR_LAUNCH(attn_pool_bwd_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

