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
// backward pass (the SCALAR form: any head size).  Head sizes 32 and 64 run the MFMA form further down.
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

// ---- MFMA form (head size 32 or 64) ------------------------------------------------------------------------------
// One 64-lane wave owns one (n, h): its Q_h / K_h / V_h slices (T <= 64 rows, zero padded to 64) sit in LDS and every
// contraction of the block runs on the matrix cores as 32 x 32 x 2 fp32 MFMAs (v_mfma_f32_32x32x2_f32: exact fp32, so the
// op keeps the fp32 parity of the scalar form):  S = Q K^T (2 x 2 tiles, K = dh), O = P V (2 x dh/32 tiles, K = Tk), and in
// the backward pass dP = dO V^T, dQ = dS K, dK = dS^T Q, dV = P^T dO.  The softmax between them is one row per lane on
// the LDS image of S (row stride 65: conflict free), P overlays the Q / K region once S is done.
typedef __attribute__((ext_vector_type(16))) float mf32x16;

// acc[mi][ni] += A[64, K] * B[K, 32 * NT] for one wave.  A(m, k) = TA ? As[k * lda + m] : As[m * lda + k];
// B(k, n) = TB ? Bs[n * ldb + k] : Bs[k * ldb + n].  K even.
template <int NT, bool TA, bool TB>
__device__ __forceinline__ void wave_mm(const float* __restrict__ As, int lda, const float* __restrict__ Bs, int ldb, int K,
                                        mf32x16 (&acc)[2][NT], int lane) {
  const int li = lane & 31, lh = lane >> 5;
  for (int k = 0; k < K; k += 2) {
    float a[2], b[NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a[mi] = TA ? As[(k + lh) * lda + mi * 32 + li] : As[(mi * 32 + li) * lda + k + lh];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) b[ni] = TB ? Bs[(ni * 32 + li) * ldb + k + lh] : Bs[(k + lh) * ldb + ni * 32 + li];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
  }
}
// Lanes of ONE wave exchange data through LDS without a workgroup barrier (the waves of a workgroup own disjoint heads):
// the wave's LDS operations execute in order, so all that is needed is that the COMPILER keeps every store in front of
// the other lanes' loads — a workgroup-scope fence pair around a wave barrier.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <int NT>
__device__ __forceinline__ void zero_acc(mf32x16 (&acc)[2][NT]) {
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
}
// head slice X_h [T, DH] (row stride C) -> LDS [64][DH + 1], rows >= T zero
template <int DH>
__device__ __forceinline__ void stage_head(const float* __restrict__ X, int T, int C, float* __restrict__ S, int lane) {
  constexpr int LD = DH + 1;
  for (int i = lane; i < 64 * DH; i += 64) {
    const int r = i / DH, c = i - r * DH;
    S[r * LD + c] = r < T ? X[(long)r * C + c] : 0.f;
  }
}
// accumulator tiles -> global rows of a head slice (rows < T)
template <int NT>
__device__ __forceinline__ void store_head(const mf32x16 (&acc)[2][NT], float* __restrict__ X, int T, int C, int lane) {
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (r < T) X[(long)r * C + ni * 32 + li] = acc[mi][ni][e];
      }
}
template <int NT>
__device__ __forceinline__ void acc_to_lds(const mf32x16 (&acc)[2][NT], float* __restrict__ S, int ld, float scale, int lane) {
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) S[(mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * ld + ni * 32 + li] = acc[mi][ni][e] * scale;
}

template <int DH, int WPB>   // WPB waves (heads) per workgroup
__global__ __launch_bounds__(64 * WPB) void mha_mfma_fwd_kernel(int N, int Tq, int Tk, int C, int heads, int causal,
                                                                const float* __restrict__ Q, const float* __restrict__ K,
                                                                const float* __restrict__ V, const float* __restrict__ kmask,
                                                                const float* __restrict__ qmask, float* __restrict__ O,
                                                                float* __restrict__ P) {
  extern __shared__ __attribute__((aligned(16))) float mha_lds[];
  constexpr int LD = DH + 1, HEAD = 64 * LD, PLD = 65;
  static_assert(2 * HEAD >= 64 * PLD, "P overlays the Q / K region");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nh = (long)blockIdx.x * WPB + wave;
  if (nh >= (long)N * heads) return;
  const int h = (int)(nh % heads), n = (int)(nh / heads);
  float* Qs = mha_lds + (long)wave * 3 * HEAD;
  float* Ks = Qs + HEAD;
  float* Vs = Ks + HEAD;
  float* Ps = Qs;                                    // [64][65], valid once S has been computed
  stage_head<DH>(Q + (long)n * Tq * C + h * DH, Tq, C, Qs, lane);
  stage_head<DH>(K + (long)n * Tk * C + h * DH, Tk, C, Ks, lane);
  stage_head<DH>(V + (long)n * Tk * C + h * DH, Tk, C, Vs, lane);
  wave_lds_sync();
  mf32x16 s[2][2];
  zero_acc<2>(s);
  wave_mm<2, false, true>(Qs, LD, Ks, LD, DH, s, lane);                   // S = Q K^T          modules.py:258
  wave_lds_sync();                                                         // every lane has read Q / K: P may overlay them
  acc_to_lds<2>(s, Ps, PLD, rsqrtf((float)DH), lane);                      // / sqrt(dh)         :261
  wave_lds_sync();
  // masks + softmax + query mask: lane = query row (modules.py:263-286)
  {
    const int tq = lane;
    if (tq < Tq) {
      float* row = Ps + tq * PLD;
      float m = -INFINITY;
      for (int tk = 0; tk < Tk; ++tk) {
        float v = row[tk];
        if (kmask[(long)n * Tk + tk] == 0.f || (causal && tk > tq)) v = MHA_NEG;
        row[tk] = v;
        m = fmaxf(m, v);
      }
      float sum = 0.f;
      for (int tk = 0; tk < Tk; ++tk) { const float e = expf(row[tk] - m); row[tk] = e; sum += e; }
      const float f = qmask[(long)n * Tq + tq] / sum;
      for (int tk = 0; tk < Tk; ++tk) row[tk] *= f;
      for (int tk = Tk; tk < 64; ++tk) row[tk] = 0.f;
    } else {
      for (int tk = 0; tk < 64; ++tk) Ps[tq * PLD + tk] = 0.f;
    }
  }
  wave_lds_sync();
  float* Pg = P + nh * Tq * Tk;
  for (int i = lane; i < Tq * Tk; i += 64) Pg[i] = Ps[(i / Tk) * PLD + (i % Tk)];
  mf32x16 o[2][DH / 32];
  zero_acc<DH / 32>(o);
  wave_mm<DH / 32, false, false>(Ps, PLD, Vs, LD, (Tk + 1) & ~1, o, lane);   // O = P V           modules.py:292
  store_head<DH / 32>(o, O + (long)n * Tq * C + h * DH, Tq, C, lane);
}

template <int DH, int WPB>
__global__ __launch_bounds__(64 * WPB) void mha_mfma_bwd_kernel(int N, int Tq, int Tk, int C, int heads, int causal,
                                                                const float* __restrict__ kmask, const float* __restrict__ Q,
                                                                const float* __restrict__ K, const float* __restrict__ V,
                                                                const float* __restrict__ P, const float* __restrict__ qmask,
                                                                const float* __restrict__ dO, float* __restrict__ dQ,
                                                                float* __restrict__ dK, float* __restrict__ dV) {
  extern __shared__ __attribute__((aligned(16))) float mha_lds[];
  constexpr int LD = DH + 1, HEAD = 64 * LD, PLD = 65, PT = 64 * PLD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nh = (long)blockIdx.x * WPB + wave;
  if (nh >= (long)N * heads) return;
  const int h = (int)(nh % heads), n = (int)(nh / heads);
  float* Qs = mha_lds + (long)wave * (4 * HEAD + 2 * PT);
  float* Ks = Qs + HEAD;
  float* Vs = Ks + HEAD;
  float* Gs = Vs + HEAD;                             // dO_h
  float* Ps = Gs + HEAD;                             // P, then unchanged
  float* Ds = Ps + PT;                               // dP, then dS
  const long qo = (long)n * Tq * C + h * DH, ko = (long)n * Tk * C + h * DH;
  stage_head<DH>(Q + qo, Tq, C, Qs, lane);
  stage_head<DH>(K + ko, Tk, C, Ks, lane);
  stage_head<DH>(V + ko, Tk, C, Vs, lane);
  stage_head<DH>(dO + qo, Tq, C, Gs, lane);
  const float* Pg = P + nh * Tq * Tk;
  for (int i = lane; i < 64 * 64; i += 64) {
    const int r = i >> 6, c = i & 63;
    Ps[r * PLD + c] = (r < Tq && c < Tk) ? Pg[r * Tk + c] : 0.f;
  }
  wave_lds_sync();
  mf32x16 t[2][2];
  zero_acc<2>(t);
  wave_mm<2, false, true>(Gs, LD, Vs, LD, DH, t, lane);                    // dP = dO V^T
  acc_to_lds<2>(t, Ds, PLD, 1.f, lane);
  wave_lds_sync();
  {
    // softmax backward on P = p0 * qm (see the scalar kernel): dS = p * (dP - sum_k p dP / qm) * scale, zero where the score
    // was a masked constant or the query row is masked
    const int tq = lane;
    const float scale = rsqrtf((float)DH);
    if (tq < Tq) {
      const float qm = qmask[(long)n * Tq + tq];
      float dot = 0.f;
      for (int tk = 0; tk < Tk; ++tk) dot += Ps[tq * PLD + tk] * Ds[tq * PLD + tk];
      for (int tk = 0; tk < 64; ++tk) {
        const bool masked = tk >= Tk || kmask[(long)n * Tk + tk] == 0.f || (causal && tk > tq);
        Ds[tq * PLD + tk] = (qm != 0.f && !masked) ? Ps[tq * PLD + tk] * (Ds[tq * PLD + tk] - dot / qm) * scale : 0.f;
      }
    } else {
      for (int tk = 0; tk < 64; ++tk) Ds[tq * PLD + tk] = 0.f;
    }
  }
  wave_lds_sync();
  mf32x16 g[2][DH / 32];
  zero_acc<DH / 32>(g);
  wave_mm<DH / 32, false, false>(Ds, PLD, Ks, LD, (Tk + 1) & ~1, g, lane);   // dQ = dS K
  store_head<DH / 32>(g, dQ + qo, Tq, C, lane);
  zero_acc<DH / 32>(g);
  wave_mm<DH / 32, true, false>(Ds, PLD, Qs, LD, (Tq + 1) & ~1, g, lane);    // dK = dS^T Q
  store_head<DH / 32>(g, dK + ko, Tk, C, lane);
  zero_acc<DH / 32>(g);
  wave_mm<DH / 32, true, false>(Ps, PLD, Gs, LD, (Tq + 1) & ~1, g, lane);    // dV = P^T dO
  store_head<DH / 32>(g, dV + ko, Tk, C, lane);
}

template <int DH, int WPB>
int launch_mha_fwd(int N, int Tq, int Tk, int C, int heads, int causal, const float* Q, const float* K, const float* V,
                   const float* km, const float* qm, float* O, float* P, hipStream_t st) {
  const size_t lds = (size_t)WPB * 3 * 64 * (DH + 1) * sizeof(float);
  TCAR_SET_LDS_ONCE((mha_mfma_fwd_kernel<DH, WPB>), lds);
  const long nh = (long)N * heads;
  TCAR_LAUNCH((mha_mfma_fwd_kernel<DH, WPB>), dim3((unsigned)((nh + WPB - 1) / WPB)), dim3(64 * WPB), lds, st, N, Tq, Tk, C, heads,
              causal, Q, K, V, km, qm, O, P);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
template <int DH, int WPB>
int launch_mha_bwd(int N, int Tq, int Tk, int C, int heads, int causal, const float* km, const float* Q, const float* K,
                   const float* V, const float* P, const float* qm, const float* dO, float* dQ, float* dK, float* dV,
                   hipStream_t st) {
  const size_t lds = (size_t)WPB * (4 * 64 * (DH + 1) + 2 * 64 * 65) * sizeof(float);
  TCAR_SET_LDS_ONCE((mha_mfma_bwd_kernel<DH, WPB>), lds);
  const long nh = (long)N * heads;
  TCAR_LAUNCH((mha_mfma_bwd_kernel<DH, WPB>), dim3((unsigned)((nh + WPB - 1) / WPB)), dim3(64 * WPB), lds, st, N, Tq, Tk, C, heads,
              causal, km, Q, K, V, P, qm, dO, dQ, dK, dV);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

}  // namespace

extern "C" int tcar_mha_core_fwd(int N, int Tq, int Tk, int C, int heads, int causal, const float* Q, const float* K,
                                 const float* V, const float* key_mask, const float* query_mask, float* O, float* P,
                                 void* stream) {
  return tcar_mha_core_fwd_tuned(nullptr, N, Tq, Tk, C, heads, causal, Q, K, V, key_mask, query_mask, O, P, stream);
}
extern "C" int tcar_mha_core_fwd_tuned(const tcar_tuning_t* tune, int N, int Tq, int Tk, int C, int heads, int causal,
                                       const float* Q, const float* K, const float* V, const float* key_mask,
                                       const float* query_mask, float* O, float* P, void* stream) {
  const TcarTuning& tn = tune ? *tune : tcar_tuning();
  if (N <= 0 || Tq <= 0 || Tk <= 0) return TCAR_OK;
  if (!Q || !K || !V || !key_mask || !query_mask || !O || !P || heads <= 0 || C <= 0 || C % heads || Tq > 64 || Tk > 64)
    return TCAR_E_ARG;
  const int dh = C / heads;
  if (tn.mha_mfma && (dh == 32 || dh == 64)) {     // matrix-core form: one wave per (n, h)
    if (dh == 32) return launch_mha_fwd<32, 4>(N, Tq, Tk, C, heads, causal, Q, K, V, key_mask, query_mask, O, P, (hipStream_t)stream);
    return launch_mha_fwd<64, 2>(N, Tq, Tk, C, heads, causal, Q, K, V, key_mask, query_mask, O, P, (hipStream_t)stream);
  }
  const long waves = (long)N * heads * Tq;
  TCAR_LAUNCH(mha_core_fwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, N, Tq, Tk, C, heads,
              causal, Q, K, V, key_mask, query_mask, O, P);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_mha_core_bwd(int N, int Tq, int Tk, int C, int heads, int causal, const float* Q, const float* K,
                                 const float* V, const float* P, const float* key_mask, const float* query_mask,
                                 const float* dO, float* dQ, float* dK, float* dV, void* stream) {
  return tcar_mha_core_bwd_tuned(nullptr, N, Tq, Tk, C, heads, causal, Q, K, V, P, key_mask, query_mask, dO, dQ, dK, dV, stream);
}
extern "C" int tcar_mha_core_bwd_tuned(const tcar_tuning_t* tune, int N, int Tq, int Tk, int C, int heads, int causal,
                                       const float* Q, const float* K, const float* V, const float* P, const float* key_mask,
                                       const float* query_mask, const float* dO, float* dQ, float* dK, float* dV, void* stream) {
  const TcarTuning& tn = tune ? *tune : tcar_tuning();
  if (N <= 0 || Tq <= 0 || Tk <= 0) return TCAR_OK;
  if (!Q || !K || !V || !P || !key_mask || !query_mask || !dO || !dQ || !dK || !dV || heads <= 0 || C <= 0 || C % heads || Tq > 64 ||
      Tk > 64)
    return TCAR_E_ARG;
  const int dh = C / heads;
  if (tn.mha_mfma && (dh == 32 || dh == 64)) {
    if (dh == 32)
      return launch_mha_bwd<32, 2>(N, Tq, Tk, C, heads, causal, key_mask, Q, K, V, P, query_mask, dO, dQ, dK, dV, (hipStream_t)stream);
    return launch_mha_bwd<64, 1>(N, Tq, Tk, C, heads, causal, key_mask, Q, K, V, P, query_mask, dO, dQ, dK, dV, (hipStream_t)stream);
  }
  TCAR_LAUNCH(mha_core_bwd_kernel, dim3(N * heads), dim3(256), 0, (hipStream_t)stream, N, Tq, Tk, C, heads, causal, key_mask, Q,
              K, V, P, query_mask, dO, dQ, dK, dV);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
