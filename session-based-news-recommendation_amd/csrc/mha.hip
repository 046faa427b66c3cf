// Optional op: the attention core of modules.py:220-304 (`multihead_attention`) for short sequences (T <= 64).
// NOT on TCAR's executed graph; built last, as an op of its own, with parity against a PyTorch fp64
// restatement (tests/test_gpu_ops.py).  The dense Q/K/V projections are ordinary linear layers (tcar_gemm_f32) and the
// residual is an add; this file is what remains: per (batch row n, head h)
//     S[tq,tk] = Q_h[tq] . K_h[tk] / sqrt(dh);   S = -2^32+1 where key_mask[n,tk] == 0 or (causal and tk > tq)   (:259-277)
//     P = softmax_tk(S);   P *= query_mask[n,tq]                                                              (:280-286)
//     O_h[tq] = sum_tk P[tq,tk] V_h[tk]                                                                        (:292)
// with key_mask = sign(|sum_c keys|), query_mask = sign(|sum_c queries|) computed by the caller.
// Sizes here are tiny (T <= 40, dh <= 64 in the reference's use): one 64-lane wave owns one (n, h, tq) row in the
// forward pass — lane = tk for the scores, lane = channel for the output — and one workgroup owns one (n, h) in the
// backward pass.  No MFMA: a 40 x 40 x 32 problem per head does not fill one 32 x 32 tile pair.
#include "tcar_common.h"

namespace {

constexpr float MHA_NEG = -4294967295.0f;        // -2**32 + 1, the reference's padding value

// grid = N * heads * Tq waves (4 per workgroup)
__global__ __launch_bounds__(256) void mha_core_fwd_kernel(int N, int Tq, int Tk, int C, int heads, int causal,
                                                           const float* __restrict__ Q, const float* __restrict__ K,
                                                           const float* __restrict__ V, const float* __restrict__ kmask,
                                                           const float* __restrict__ qmask, float* __restrict__ O,
                                                           float* __restrict__ P) {
  const int lane = threadIdx.x & 63;
  const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wv >= (long)N * heads * Tq) return;
  const int tq = (int)(wv % Tq);
  const int h = (int)((wv / Tq) % heads);
  const int n = (int)(wv / ((long)Tq * heads));
  const int dh = C / heads;
  const float* q = Q + ((long)n * Tq + tq) * C + h * dh;
  float s = -INFINITY;
  if (lane < Tk) {
    const float* k = K + ((long)n * Tk + lane) * C + h * dh;
    float acc = 0.f;
    for (int j = 0; j < dh; ++j) acc = fmaf(q[j], k[j], acc);
    s = acc * rsqrtf((float)dh);
    if (kmask[(long)n * Tk + lane] == 0.f || (causal && lane > tq)) s = MHA_NEG;
  }
  const float m = wave_max(s);
  const float e = lane < Tk ? expf(s - m) : 0.f;
  const float p = e / wave_sum(e) * qmask[(long)n * Tq + tq];
  if (lane < Tk) P[(((long)n * heads + h) * Tq + tq) * Tk + lane] = p;
  float* o = O + ((long)n * Tq + tq) * C + h * dh;
  for (int j0 = 0; j0 < dh; j0 += 64) {
    const int j = j0 + lane;
    float acc = 0.f;
    for (int tk = 0; tk < Tk; ++tk) {
      const float pk = __shfl(p, tk);
      if (j < dh) acc = fmaf(pk, V[((long)n * Tk + tk) * C + h * dh + j], acc);
    }
    if (j < dh) o[j] = acc;
  }
}

// one workgroup per (n, h); P [Tq, Tk] and dS live in LDS (Tq, Tk <= 64)
__global__ __launch_bounds__(256) void mha_core_bwd_kernel(int N, int Tq, int Tk, int C, int heads, int causal,
                                                           const float* __restrict__ kmask, const float* __restrict__ Q, const float* __restrict__ K,
                                                           const float* __restrict__ V, const float* __restrict__ P,
                                                           const float* __restrict__ qmask, const float* __restrict__ dO,
                                                           float* __restrict__ dQ, float* __restrict__ dK,
                                                           float* __restrict__ dV) {
  __shared__ float sP[64 * 64], sD[64 * 64];
  const int tid = threadIdx.x;
  const int h = blockIdx.x % heads, n = blockIdx.x / heads;
  const int dh = C / heads;
  const float scale = rsqrtf((float)dh);
  const float* Pn = P + ((long)n * heads + h) * Tq * Tk;
  // dP[tq,tk] = dO_h[tq] . V_h[tk]
  for (int i = tid; i < Tq * Tk; i += 256) {
    const int tq = i / Tk, tk = i - tq * Tk;
    const float* go = dO + ((long)n * Tq + tq) * C + h * dh;
    const float* v = V + ((long)n * Tk + tk) * C + h * dh;
    float acc = 0.f;
    for (int j = 0; j < dh; ++j) acc = fmaf(go[j], v[j], acc);
    sP[i] = Pn[i];
    sD[i] = acc;
  }
  __syncthreads();
  // softmax backward on the UNMASKED probabilities p0 = P / qmask:  dS = p0 * (qm * dP - sum_k p0 * qm * dP) * scale
  // (masked scores sit at -2^32+1: their p0 is exactly 0, so is their dS; rows with qmask == 0 have P == 0 and dS == 0)
  for (int tq = tid; tq < Tq; tq += 256) {
    const float qm = qmask[(long)n * Tq + tq];
    float dot = 0.f;
    for (int tk = 0; tk < Tk; ++tk) dot += sP[tq * Tk + tk] * sD[tq * Tk + tk];      // = sum p0*qm*dP  (P = p0*qm)
    for (int tk = 0; tk < Tk; ++tk) {
      const float pq = sP[tq * Tk + tk];                                               // p0 * qm
      // a masked score is the constant -2^32+1 (tf.where): no gradient reaches Q / K through it, even in a row whose keys
      // are ALL masked (where the softmax is uniform over them)
      const bool masked = kmask[(long)n * Tk + tk] == 0.f || (causal && tk > tq);
      sD[tq * Tk + tk] = (qm != 0.f && !masked) ? pq * (sD[tq * Tk + tk] - dot / qm) * scale : 0.f;
    }
  }
  __syncthreads();
  // dQ_h[tq] = sum_tk dS[tq,tk] K_h[tk];  dK_h[tk] = sum_tq dS[tq,tk] Q_h[tq];  dV_h[tk] = sum_tq P[tq,tk] dO_h[tq]
  for (int i = tid; i < Tq * dh; i += 256) {
    const int tq = i / dh, j = i - tq * dh;
    float acc = 0.f;
    for (int tk = 0; tk < Tk; ++tk) acc = fmaf(sD[tq * Tk + tk], K[((long)n * Tk + tk) * C + h * dh + j], acc);
    dQ[((long)n * Tq + tq) * C + h * dh + j] = acc;
  }
  for (int i = tid; i < Tk * dh; i += 256) {
    const int tk = i / dh, j = i - tk * dh;
    float ak = 0.f, av = 0.f;
    for (int tq = 0; tq < Tq; ++tq) {
      ak = fmaf(sD[tq * Tk + tk], Q[((long)n * Tq + tq) * C + h * dh + j], ak);
      av = fmaf(sP[tq * Tk + tk], dO[((long)n * Tq + tq) * C + h * dh + j], av);
    }
    dK[((long)n * Tk + tk) * C + h * dh + j] = ak;
    dV[((long)n * Tk + tk) * C + h * dh + j] = av;
  }
}

}  // namespace

extern "C" int tcar_mha_core_fwd(int N, int Tq, int Tk, int C, int heads, int causal, const float* Q, const float* K,
                                 const float* V, const float* key_mask, const float* query_mask, float* O, float* P,
                                 void* stream) {
  if (N <= 0 || Tq <= 0 || Tk <= 0) return TCAR_OK;
  if (!Q || !K || !V || !key_mask || !query_mask || !O || !P || heads <= 0 || C <= 0 || C % heads || Tq > 64 || Tk > 64)
    return TCAR_E_ARG;
  const long waves = (long)N * heads * Tq;
  TCAR_LAUNCH(mha_core_fwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, N, Tq, Tk, C, heads,
              causal, Q, K, V, key_mask, query_mask, O, P);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_mha_core_bwd(int N, int Tq, int Tk, int C, int heads, int causal, const float* Q, const float* K,
                                 const float* V, const float* P, const float* key_mask, const float* query_mask,
                                 const float* dO, float* dQ, float* dK, float* dV, void* stream) {
  if (N <= 0 || Tq <= 0 || Tk <= 0) return TCAR_OK;
  if (!Q || !K || !V || !P || !key_mask || !query_mask || !dO || !dQ || !dK || !dV || heads <= 0 || C <= 0 || C % heads || Tq > 64 ||
      Tk > 64)
    return TCAR_E_ARG;
  TCAR_LAUNCH(mha_core_bwd_kernel, dim3(N * heads), dim3(256), 0, (hipStream_t)stream, N, Tq, Tk, C, heads, causal, key_mask, Q,
              K, V, P, query_mask, dO, dQ, dK, dV);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
