// Split-bf16 MFMA GEMM for gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the full-catalog scoring
// contractions of model_combine.py:138 and their two gradients at ~16x the fp32 matrix rate.
//
// gfx950 has no TF32/xf32 path, so fp32 operands are carried as TWO bf16 planes, x = hi + lo with
// hi = bf16(x), lo = bf16(x - hi) (relative residual <= 2^-17), and a product is three MFMAs
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi                  (NSPLIT = 3, dropped term <= 2^-16 |a||b|)
// accumulated in fp32: logits and gradients stay within ~1e-5 of the fp32 path (tests keep the 1e-3 gate) at
// 3/16 of its MFMA time.  NSPLIT = 1 uses the hi planes only (plain bf16, ~4e-3 relative).
//
// Operand layouts (both planes share one layout), tile 128 x 128 x 32, 256 threads = 4 waves (2 x 2), each wave
// 64 x 64 = 2 x 2 MFMA tiles of 32 x 32:
//   LAY 0  k-contiguous   X[row*ld + k]   LDS image [128 rows][32 k] (64 B + 16 B pad; stride 80 B = 20 dwords,
//          20/4 odd => conflict-free ds_read_b128); a lane's fragment (row = lane&31, k = 8*(lane>>5)+0..7) is one
//          16-byte read.
//   LAY 1  m/n-contiguous X[k*ld + col]   LDS image [32 k][128 cols] exactly as in memory (256 B + 64 B pad;
//          stride 320 B makes the 4 k-rows x 2 column groups x 4 pieces of one 32-lane half hit 32 distinct
//          8-byte bank pairs); fragments come from ds_read_b64_tr_b16, the gfx950 transposing LDS read: per 16-lane
//          group it returns, to lane i, column i of a 4 (k) x 16 (col) block => 4 consecutive k of the lane's own
//          row/column; two reads give the 8 k of the fragment.  No transposed copy of E or dlogits ever exists.
// Staging is global -> registers -> LDS, double buffered (next tile's loads in flight during the MFMAs); the
// XCD-aware tile remap is the one of gemm_f32.hip.  Split-K (slabs) serves the catalog-long contraction of dX.
#include "tcar_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int TB = 128, KB = 32;
constexpr int KC_STRIDE = 80;    // bytes per row of a k-contiguous tile
constexpr int MC_STRIDE = 320;   // bytes per k-row of an m/n-contiguous tile
constexpr int PLANE = 10240;     // bytes of one plane of one operand tile (128*80 == 32*320)

struct BArgs {
  const __bf16* A[2];
  const __bf16* B[2];
  long lda, ldb;
  float* C; long ldc;
  float* C2; long ldc2; int csplit;     // columns >= csplit go to C2 (column index rebased); csplit >= N: unused
  int M, N, K, kchunk, mode, mt, nt;
};

// global -> registers: two 16-byte pieces per thread and plane
template <int LAY>
__device__ __forceinline__ void gload(const __bf16* __restrict__ P, long ld, int r0, int rmax, int k0, int kend, int tid,
                                      uint4 (&reg)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = tid + 256 * i;
    if (LAY == 0) {
      const int row = f >> 2, c = f & 3;
      const int gr = r0 + row, gk = k0 + c * 8;
      reg[i] = (gr < rmax && gk < kend) ? *reinterpret_cast<const uint4*>(P + (long)gr * ld + gk) : make_uint4(0, 0, 0, 0);
    } else {
      const int krow = f >> 4, c = f & 15;
      const int gk = k0 + krow, gr = r0 + c * 8;
      reg[i] = (gk < kend && gr + 7 < rmax) ? *reinterpret_cast<const uint4*>(P + (long)gk * ld + gr) : make_uint4(0, 0, 0, 0);
    }
  }
}
template <int LAY>
__device__ __forceinline__ void lstore(char* __restrict__ S, int tid, const uint4 (&reg)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = tid + 256 * i;
    if (LAY == 0) {
      const int row = f >> 2, c = f & 3;
      *reinterpret_cast<uint4*>(S + row * KC_STRIDE + c * 16) = reg[i];
    } else {
      const int krow = f >> 4, c = f & 15;
      *reinterpret_cast<uint4*>(S + krow * MC_STRIDE + c * 16) = reg[i];
    }
  }
}
// fragment of the 32-row block starting at tile row/col `base`, k16 sub-step s (0/1) of the 32-deep stage
template <int LAY>
__device__ __forceinline__ bf16x8 frag(const char* __restrict__ S, int base, int s, int lane) {
  if (LAY == 0) {
    return *reinterpret_cast<const bf16x8*>(S + (base + (lane & 31)) * KC_STRIDE + s * 32 + (lane >> 5) * 16);
  } else {
    const int g = lane >> 4, i = lane & 15;
    const int mbase = 16 * (g & 1), kbase = 8 * (g >> 1), q = i >> 2, p = i & 3;
    const char* a0 = S + (s * 16 + kbase + q) * MC_STRIDE + (base + mbase + 4 * p) * 2;
    typedef __attribute__((address_space(3))) bf16x4* lds_p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a0 + 4 * MC_STRIDE));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

template <int LA, int LB, int NSPLIT>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const BArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = (NSPLIT == 1) ? 1 : 2;        // planes per operand
  constexpr int STAGE = 2 * NP * PLANE;            // A planes then B planes
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  int tm, tn;
  if (g.mt <= g.nt) { tn = id / g.mt; tm = id - tn * g.mt; } else { tm = id / g.nt; tn = id - tm * g.nt; }
  const int m0 = tm * TB, n0 = tn * TB;
  const int ks = blockIdx.z * g.kchunk;
  const int ke = min(g.K, ks + g.kchunk);
  const int nit = (ke - ks + KB - 1) / KB;
  const int a_rmax = (LA == 0) ? g.M : min((int)g.lda, (g.M + 7) & ~7);
  const int b_rmax = (LB == 0) ? g.N : min((int)g.ldb, (g.N + 7) & ~7);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  uint4 ra[NP][2], rb[NP][2];
  if (nit > 0) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      gload<LA>(g.A[p], g.lda, m0, a_rmax, ks, ke, tid, ra[p]);
      gload<LB>(g.B[p], g.ldb, n0, b_rmax, ks, ke, tid, rb[p]);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      lstore<LA>(smem + p * PLANE, tid, ra[p]);
      lstore<LB>(smem + (NP + p) * PLANE, tid, rb[p]);
    }
  }
  __syncthreads();
  for (int it = 0; it < nit; ++it) {
    const char* St = smem + (it & 1) * STAGE;
    const bool more = (it + 1 < nit);
    if (more) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        gload<LA>(g.A[p], g.lda, m0, a_rmax, ks + (it + 1) * KB, ke, tid, ra[p]);
        gload<LB>(g.B[p], g.ldb, n0, b_rmax, ks + (it + 1) * KB, ke, tid, rb[p]);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[NP][2], b[NP][2];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a[p][t] = frag<LA>(St + p * PLANE, wm * 64 + t * 32, s, lane);
          b[p][t] = frag<LB>(St + (NP + p) * PLANE, wn * 64 + t * 32, s, lane);
        }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if constexpr (NSPLIT == 3) {
            acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NP - 1][u], b[0][t], acc[u][t], 0, 0, 0);
            acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][u], b[NP - 1][t], acc[u][t], 0, 0, 0);
          }
          acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][u], b[0][t], acc[u][t], 0, 0, 0);
        }
    }
    if (more) {
      char* Sn = smem + ((it + 1) & 1) * STAGE;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        lstore<LA>(Sn + p * PLANE, tid, ra[p]);
        lstore<LB>(Sn + (NP + p) * PLANE, tid, rb[p]);
      }
    }
    __syncthreads();
  }

  float* C1 = g.C + (g.mode == 1 ? (long)blockIdx.z * g.M * g.ldc : 0L);
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = n0 + wn * 64 + t * 32 + li;
      if (col >= g.N) continue;
      float* base = (col < g.csplit) ? C1 + col : g.C2 + (col - g.csplit);
      const long ld = (col < g.csplit) ? g.ldc : g.ldc2;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + u * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row < g.M) base[(long)row * ld] = acc[u][t][e];
      }
    }
}

// fp32 [rows, cols] (ld) -> bf16 hi / lo planes [rows, ld16]; columns [cols, ld16) are zero filled.
// Optional second, PACKED output pair taking columns [0, c0) U [c1, cols) (the dE operand: item | time blocks).
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, long ld, int rows, int cols,
                                                         __bf16* __restrict__ hi, __bf16* __restrict__ lo, long ld16,
                                                         __bf16* __restrict__ phi, __bf16* __restrict__ plo, long pld,
                                                         int c0, int c1) {
  const long c4n = ld16 >> 2;
  const long total = (long)rows * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (c + j < cols) ? x[r * ld + c + j] : 0.f;
    bf16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = (__bf16)v[j];
      l[j] = (__bf16)(v[j] - (float)h[j]);
    }
    *reinterpret_cast<bf16x4*>(hi + r * ld16 + c) = h;
    if (lo) *reinterpret_cast<bf16x4*>(lo + r * ld16 + c) = l;
    if (phi && (c < c0 || c >= c1)) {
      const int pc = c < c0 ? c : c - (c1 - c0);
      *reinterpret_cast<bf16x4*>(phi + r * pld + pc) = h;
      if (plo) *reinterpret_cast<bf16x4*>(plo + r * pld + pc) = l;
    }
  }
}

template <int LA, int LB>
int launch_b(const BArgs& g, int nsplit, int splitk, hipStream_t st) {
  const size_t lds1 = 2 * 2 * 1 * PLANE, lds3 = 2 * 2 * 2 * PLANE;
  dim3 grid(g.mt * g.nt, 1, splitk), block(256);
  if (nsplit == 3) {
    static bool done = false;
    if (!done) {
      (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<LA, LB, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      done = true;
    }
    TCAR_LAUNCH((gemm_bf16_kernel<LA, LB, 3>), grid, block, lds3, st, g);
  } else {
    TCAR_LAUNCH((gemm_bf16_kernel<LA, LB, 1>), grid, block, lds1, st, g);
  }
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

}  // namespace

extern "C" int tcar_gemm_bf16(int layout, int M, int N, int K, const void* A_hi, const void* A_lo, int64_t lda,
                              const void* B_hi, const void* B_lo, int64_t ldb, float* C, int64_t ldc, float* C2,
                              int64_t ldc2, int csplit, int nsplit, int splitk, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return TCAR_OK;
  if (layout < 0 || layout > 2 || !A_hi || !B_hi || !C || (nsplit != 1 && nsplit != 3)) return TCAR_E_ARG;
  if (nsplit == 3 && (!A_lo || !B_lo)) return TCAR_E_ARG;
  if ((lda & 7) || (ldb & 7) || !tcar_aligned16(A_hi) || !tcar_aligned16(B_hi)) return TCAR_E_ARG;
  const bool a_kc = (layout != 2), b_kc = (layout == 1);
  if ((a_kc || b_kc) && (K & 7)) return TCAR_E_ARG;
  BArgs g;
  g.A[0] = (const __bf16*)A_hi; g.A[1] = (const __bf16*)A_lo; g.B[0] = (const __bf16*)B_hi; g.B[1] = (const __bf16*)B_lo;
  g.lda = lda; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.C2 = C2 ? C2 : C; g.ldc2 = C2 ? ldc2 : ldc; g.csplit = C2 ? csplit : N;
  g.M = M; g.N = N; g.K = K;
  if (splitk < 1) splitk = 1;
  if (splitk > 1 && C2) return TCAR_E_ARG;
  int kchunk = (K + splitk - 1) / splitk;
  kchunk = ((kchunk + KB - 1) / KB) * KB;
  g.kchunk = kchunk;
  splitk = (K + kchunk - 1) / kchunk;
  g.mode = splitk > 1 ? 1 : 0;
  g.mt = (M + TB - 1) / TB; g.nt = (N + TB - 1) / TB;
  hipStream_t st = (hipStream_t)stream;
  if (layout == 0) return launch_b<0, 1>(g, nsplit, splitk, st);
  if (layout == 1) return launch_b<0, 0>(g, nsplit, splitk, st);
  return launch_b<1, 1>(g, nsplit, splitk, st);
}

extern "C" int tcar_split_bf16(const float* x, int64_t ld, int rows, int cols, void* hi, void* lo, int64_t ld16,
                               void* packed_hi, void* packed_lo, int64_t packed_ld, int c0, int c1, void* stream) {
  if (rows <= 0 || cols <= 0) return TCAR_OK;
  if (!x || !hi || (ld16 & 3) || ld16 < cols || (c0 & 3) || (c1 & 3) || (packed_ld & 3)) return TCAR_E_ARG;
  long total = (long)rows * (ld16 >> 2);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  TCAR_LAUNCH(split_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long)ld, rows, cols, (__bf16*)hi,
              (__bf16*)lo, (long)ld16, (__bf16*)packed_hi, (__bf16*)packed_lo, (long)packed_ld, c0, c1);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
