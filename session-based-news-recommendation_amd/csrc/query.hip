// The click-query MLP of the multi-attention pool in ONE launch (modules.py:138-139):
//     q1 = relu(click_t Wq1 + b1)   [B, 256]      q = tanh(q1 Wq2 + b2)   [B, 512]
// for the reference's hidden sizes (padded H = 256, time hidden 64: click_t has 128 columns).  As two grouped small GEMMs the two
// layers are two dependent launches of 32 and 64 workgroups on the chain gather -> q1 -> q -> pools (22 + 30 us inside a step);
// the work is 0.17 GFLOP.  Here a workgroup of 512 threads owns a few (QS) sessions and walks both layers in fp32 FMAs: every
// thread keeps all of its weight loads of a phase in flight at once (16, then 2 x 32 sixteen-byte loads: one memory round trip
// per phase instead of one per 64-deep GEMM stage), partial sums over the K groups are folded through LDS in a fixed order.
// The step driver runs it on a side stream beside the input projections (step.hip: session_forward); both outputs are stored
// write-through (sc1) so that the consumer behind the completion flag needs no L2 write-back.
#include "tcar_common.h"

namespace {
constexpr int QS = 4;                       // sessions per workgroup (8: 42 us beside the projections — the FMAs of QS sessions per thread; 4: see DESIGN.md)
constexpr int CT = 128, H1 = 256, H2 = 512; // click_t columns, q1 columns, q columns
struct QMlpArgs {
  const float* click; const float* w1; const float* b1; const float* w2; const float* b2;
  float* q1; float* q;
  int B;
  TcarSignal sig;
};
__device__ __forceinline__ void st4_wt(float* p, float4 v, bool write_through) {
  if (write_through) st4_sc1(p, v);
  else st4(p, v);
}

__global__ __launch_bounds__(512) void query_mlp_kernel(const QMlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* clk = lds;                         // [QS][CT]
  float* q1s = clk + QS * CT;               // [QS][H1]
  float* part = q1s + QS * H1;              // [8][QS][H1] floats, then [4][QS][H2]
  const int t = threadIdx.x;
  const int b0 = blockIdx.x * QS;
  const bool wt = a.sig.cnt != nullptr;
  // ---- layer 1: thread = (column group cg of 64, K group kq of 8 x 16 rows)
  const int cg = t & 63, kq = t >> 6;
  float4 w[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) w[j] = ld4(a.w1 + (long)(kq * 16 + j) * H1 + cg * 4);
  if (t < QS * CT / 4) {
    const int s = t >> 5, c = t & 31;
    st4(clk + s * CT + c * 4, b0 + s < a.B ? ld4(a.click + (long)(b0 + s) * CT + c * 4) : zero4());
  }
  __syncthreads();
  {
    float4 acc[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) acc[s] = zero4();
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
#pragma unroll
      for (int s = 0; s < QS; ++s) {
        const float4 c = *reinterpret_cast<const float4*>(clk + s * CT + kq * 16 + j4 * 4);     // wave-uniform address
        acc[s] = fma4(w[j4 * 4 + 0], c.x, acc[s]);
        acc[s] = fma4(w[j4 * 4 + 1], c.y, acc[s]);
        acc[s] = fma4(w[j4 * 4 + 2], c.z, acc[s]);
        acc[s] = fma4(w[j4 * 4 + 3], c.w, acc[s]);
      }
    }
#pragma unroll
    for (int s = 0; s < QS; ++s) st4(part + (kq * QS + s) * H1 + cg * 4, acc[s]);
  }
  __syncthreads();
  if (t < QS * 64) {   // q1[s][4 cg ..] = relu(b1 + sum over the 8 K groups, in group order)
    const int s = t >> 6;
    float4 v = ld4(a.b1 + cg * 4);
#pragma unroll
    for (int k = 0; k < 8; ++k) v = add4(v, *reinterpret_cast<const float4*>(part + (k * QS + s) * H1 + cg * 4));
    v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    st4(q1s + s * H1 + cg * 4, v);
    if (b0 + s < a.B) st4_wt(a.q1 + (long)(b0 + s) * H1 + cg * 4, v, wt);
  }
  __syncthreads();
  // ---- layer 2: thread = (column group c2 of 128, K group k2 of 4 x 64 rows), two half groups of 32 rows
  const int c2 = t & 127, k2 = t >> 7;
  float4 acc2[QS];
#pragma unroll
  for (int s = 0; s < QS; ++s) acc2[s] = zero4();
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    const int r0 = k2 * 64 + half * 32;
    float4 u[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) u[j] = ld4(a.w2 + (long)(r0 + j) * H2 + c2 * 4);
#pragma unroll
    for (int j4 = 0; j4 < 8; ++j4) {
#pragma unroll
      for (int s = 0; s < QS; ++s) {
        const float4 c = *reinterpret_cast<const float4*>(q1s + s * H1 + r0 + j4 * 4);
        acc2[s] = fma4(u[j4 * 4 + 0], c.x, acc2[s]);
        acc2[s] = fma4(u[j4 * 4 + 1], c.y, acc2[s]);
        acc2[s] = fma4(u[j4 * 4 + 2], c.z, acc2[s]);
        acc2[s] = fma4(u[j4 * 4 + 3], c.w, acc2[s]);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < QS; ++s) st4(part + (k2 * QS + s) * H2 + c2 * 4, acc2[s]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < QS / 4; ++i) {   // q[s][4 c ..] = tanh(b2 + sum over the 4 K groups)
    const int idx = t + i * 512, s = idx >> 7, c = idx & 127;
    float4 v = ld4(a.b2 + c * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) v = add4(v, *reinterpret_cast<const float4*>(part + (k * QS + s) * H2 + c * 4));
    v = make_float4(tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w));
    if (b0 + s < a.B) st4_wt(a.q + (long)(b0 + s) * H2 + c * 4, v, wt);
  }
  tcar_signal_done(a.sig);
}

// ---- backward of both layers in ONE launch (round 4) ----------------------------------------------------------------------------------
//     dq1 = (dq Wq2^T) * relu'(q1)   [B, 256]       dclick = dq1 Wq1^T   [B, 128]        (modules.py:138-139 backward; dq arrives
// already through tanh' — the pool backward applies it).  As small GEMMs these are two DEPENDENT launches of 32 and 16 workgroups
// with 8 and 4 serial 64-deep stages (32 + 18-22 us) in front of the small tables' pass and the weight gradients.  Both products
// contract along the CONTIGUOUS dimension of the weight (row j of Wq2, row m of Wq1), so a wave takes whole rows — 16 bytes per
// lane and 64 (128) float4 per row: fully coalesced — keeps the QS sessions' gradient rows in registers, and reduces its
// rows x sessions partial dots across the 64 lanes with a butterfly reduce-scatter (the value count halves at every exchange:
// V - 1 exchanges for V values instead of 6 V), a fixed order: bit-for-bit repeatable.  fp32 FMAs throughout.
struct QBwdArgs {
  const float* dq; const float* q1; const float* w1; const float* w2;
  float* dq1; float* dclick;
  int B;
  TcarSignal sig;
};
// v[0 .. V) per lane -> the sum over the 64 lanes of v[i] ends in lane (i * 64 / V) .. (each index owned by 64 / V adjacent lanes,
// all of which hold it); V a power of two <= 64
template <int N, int M, int V>
__device__ __forceinline__ void butterfly_step(float (&v)[V], int lane) {
  if constexpr (N > 1) {
    const bool up = (lane & M) != 0;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const float keep = up ? v[i + N / 2] : v[i];
      const float send = up ? v[i] : v[i + N / 2];
      v[i] = keep + __shfl_xor(send, M);
    }
    butterfly_step<N / 2, M / 2, V>(v, lane);
  }
}
template <int V>
__device__ __forceinline__ float butterfly_reduce(float (&v)[V], int lane) {
  butterfly_step<V, 32, V>(v, lane);
  float r = v[0];
  // the remaining exchanges (64 / V > 1 lanes own the same index): plain pair sums
#pragma unroll
  for (int m = 32 / V; m >= 1; m >>= 1) r += __shfl_xor(r, m);
  return r;
}

__global__ __launch_bounds__(512) void query_mlp_bwd_kernel(const QBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float dq1s[QS * H1];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b0 = blockIdx.x * QS;
  const bool wt = a.sig.cnt != nullptr;
  // ---- layer 2 backward: wave w owns rows j = 32 w .. 32 w + 31 of Wq2 [256, 512]; lane holds columns 4 lane .. and 256 + 4 lane ..
  // (a.dq == NULL: dq1 is an INPUT — written by the main chain's grouped GEMM — and only the layer-1 half below runs)
  if (a.dq) {
    float4 g0[QS], g1[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
      const bool ok = b0 + s < a.B;
      g0[s] = ok ? ld4(a.dq + (long)(b0 + s) * H2 + lane * 4) : zero4();
      g1[s] = ok ? ld4(a.dq + (long)(b0 + s) * H2 + 256 + lane * 4) : zero4();
    }
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {            // 8 rows per trip: 16 independent 16-byte loads per lane in flight
      const int j0 = wave * 32 + ch * 8;
      float4 wa[8], wb[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        wa[r] = ld4(a.w2 + (long)(j0 + r) * H2 + lane * 4);
        wb[r] = ld4(a.w2 + (long)(j0 + r) * H2 + 256 + lane * 4);
      }
      float v[8 * QS];
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int s = 0; s < QS; ++s) v[r * QS + s] = dot4(wa[r], g0[s]) + dot4(wb[r], g1[s]);
      const float sum = butterfly_reduce<8 * QS>(v, lane);          // index (r, s) = lane >> 1
      const int idx = lane >> 1, r = idx / QS, s = idx - r * QS, j = j0 + r;
      const bool live = b0 + s < a.B;
      const float y = live ? a.q1[(long)(b0 + s) * H1 + j] : 0.f;
      const float d = y > 0.f ? sum : 0.f;                           // relu'
      if ((lane & 1) == 0) {
        dq1s[s * H1 + j] = d;
        if (live) {
          if (wt) __hip_atomic_store(a.dq1 + (long)(b0 + s) * H1 + j, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else a.dq1[(long)(b0 + s) * H1 + j] = d;
        }
      }
    }
  }
  __syncthreads();
  // ---- layer 1 backward: wave w owns rows m = 16 w .. 16 w + 15 of Wq1 [128, 256]; lane holds columns 4 lane ..
  {
    float4 g[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s)
      g[s] = a.dq ? *reinterpret_cast<const float4*>(dq1s + s * H1 + lane * 4)
                  : (b0 + s < a.B ? ld4(a.dq1 + (long)(b0 + s) * H1 + lane * 4) : zero4());
    float4 w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = ld4(a.w1 + (long)(wave * 16 + r) * H1 + lane * 4);
    float v[16 * QS];
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int s = 0; s < QS; ++s) v[r * QS + s] = dot4(w[r], g[s]);
    const float sum = butterfly_reduce<16 * QS>(v, lane);            // index (r, s) = lane
    const int r = lane / QS, s = lane - r * QS, m = wave * 16 + r;
    if (b0 + s < a.B) {
      if (wt) __hip_atomic_store(a.dclick + (long)(b0 + s) * CT + m, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else a.dclick[(long)(b0 + s) * CT + m] = sum;
    }
  }
  tcar_signal_done(a.sig);
}
}  // namespace

// dq1 [B, 256] = (dq [B, 512] Wq2^T) * relu'(q1), dclick [B, 128] = dq1 Wq1^T: the input-gradient half of the click-query MLP's
// backward pass (modules.py:138-139; weight and bias gradients are x^T dy GEMMs / column sums over dq and dq1 elsewhere).  Same
// restrictions as tcar_query_mlp (d->ldh == 256, d->ldt == 64, dense row-major operands).
extern "C" int tcar_query_mlp_bwd(const tcar_dims_t* d, int B, const float* dq, const float* q1, const float* q1_w, const float* q2_w,
                                  float* dq1, float* dclick, void* stream) {
  return tcar_query_mlp_bwd_o(d, B, dq, q1, q1_w, q2_w, dq1, dclick, stream, nullptr);
}
// (flag-capable: dq1 and dclick leave write-through when the launch carries a flag)
int tcar_query_mlp_bwd_o(const tcar_dims_t* d, int B, const float* dq, const float* q1, const float* q1_w, const float* q2_w, float* dq1,
                         float* dclick, void* stream, TcarOpt* o) {
  if (!d || d->ldh != H1 || d->ldt * 2 != CT || B <= 0) return TCAR_E_ARG;
  if (!q1_w || !dq1 || !dclick || !tcar_aligned16(q1_w) || !tcar_aligned16(dq1)) return TCAR_E_ARG;
  if (dq && (!q1 || !q2_w || !tcar_aligned16(dq) || !tcar_aligned16(q2_w))) return TCAR_E_ARG;
  QBwdArgs a{};
  a.dq = dq; a.q1 = q1; a.w1 = q1_w; a.w2 = q2_w; a.dq1 = dq1; a.dclick = dclick; a.B = B;
  a.sig = tcar_sig(o);
  TCAR_LAUNCH(query_mlp_bwd_kernel, dim3((unsigned)((B + QS - 1) / QS)), dim3(512), 0, (hipStream_t)stream, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

namespace {
}  // namespace

// q1 [B, 256] = relu(click_t [B, 128] Wq1 [128, 256] + b1), q [B, 512] = tanh(q1 Wq2 [256, 512] + b2): dense row-major operands
// with exactly these leading dimensions (d->ldh == 256 and d->ldt == 64), else TCAR_E_ARG — the caller then runs the two layers
// as small GEMMs.  Carries a pending completion flag (tcar_common.h).
extern "C" int tcar_query_mlp(const tcar_dims_t* d, int B, const float* click_t, const float* q1_w, const float* q1_b,
                              const float* q2_w, const float* q2_b, float* q1, float* q, void* stream) {
  return tcar_query_mlp_o(d, B, click_t, q1_w, q1_b, q2_w, q2_b, q1, q, stream, nullptr);
}
// (flag-capable: q1 and q leave write-through when the launch carries a flag)
int tcar_query_mlp_o(const tcar_dims_t* d, int B, const float* click_t, const float* q1_w, const float* q1_b, const float* q2_w,
                     const float* q2_b, float* q1, float* q, void* stream, TcarOpt* o) {
  if (!d || d->ldh != H1 || d->ldt * 2 != CT || B <= 0) return TCAR_E_ARG;
  if (!click_t || !q1_w || !q1_b || !q2_w || !q2_b || !q1 || !q) return TCAR_E_ARG;
  if (!tcar_aligned16(click_t) || !tcar_aligned16(q1_w) || !tcar_aligned16(q1_b) || !tcar_aligned16(q2_w) || !tcar_aligned16(q2_b) ||
      !tcar_aligned16(q1) || !tcar_aligned16(q))
    return TCAR_E_ARG;
  QMlpArgs a{};
  a.click = click_t; a.w1 = q1_w; a.b1 = q1_b; a.w2 = q2_w; a.b2 = q2_b; a.q1 = q1; a.q = q; a.B = B;
  a.sig = tcar_sig(o);
  constexpr size_t lds = (size_t)(QS * CT + QS * H1 + 4 * QS * H2) * sizeof(float);     // 76 KB
  TCAR_SET_LDS_ONCE(query_mlp_kernel, lds);
  TCAR_LAUNCH(query_mlp_kernel, dim3((unsigned)((B + QS - 1) / QS)), dim3(512), lds, (hipStream_t)stream, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
