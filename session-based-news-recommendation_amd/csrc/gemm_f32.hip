// fp32-input / fp32-accumulate MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, 256 FLOP/clk/CU).
//
// One kernel template covers the three operand layouts of the TCAR step (include/tcar_hip.h, tcar_gemm_f32):
//   LA = 0: A is k-contiguous  A[m*lda + k]      LA = 1: A is m-contiguous  A[k*lda + m]
//   LB = 0: B is k-contiguous  B[n*ldb + k]      LB = 1: B is n-contiguous  B[k*ldb + n]
// Tile 128(M) x 128(N) x 32(K), 256 threads = 4 waves in a 2x2 grid, each wave owns 64x64 = 2x2 MFMA tiles
// (64 accumulator VGPRs).  Operands are staged global -> registers -> LDS with the LDS image mirroring the
// global layout (so every global access is a 16-byte, fully coalesced load):
//   k-contiguous tiles  [128 rows][32 k + 4 pad]  read back with ds_read_b128 (row stride 36 dwords: 36/4 odd
//                       => the 16-lane b128 groups hit 16 distinct 4-bank slots, conflict free)
//   m/n-contiguous tiles [32 k][128 + 4 pad]      read back with ds_read_b32 (lanes = consecutive columns)
// MFMA operand pairing: the instruction contracts k in {0,1} = lane>>5.  For an 8-deep k chunk, lane half h
// holds k = 8c + 4h + j (j = 0..3) of BOTH operands, so MFMA j sums k = 8c+j and 8c+4+j: any pairing is valid
// as long as A and B agree, and it lets a k-contiguous operand be fetched with one b128 read per 4 MFMAs.
// LDS is double buffered; the next tile's global loads are in flight while the current one is multiplied.
// Workgroup ids are remapped so that each XCD (private L2) receives a contiguous run of tile ids, with the
// shorter tile dimension fastest: the workgroups that share the large operand's tile run on one XCD.
#include "tcar_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K;
  long lda, ldb, ldc;
  int act, beta, kchunk;
  int mt, nt;  // tile counts
};

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int KC_LD = BK + 4;    // row stride (floats) of a k-contiguous tile
constexpr int MC_LD = BM + 4;    // row stride (floats) of an m/n-contiguous tile
constexpr int TILE_FLOATS = (BM * KC_LD > BK * MC_LD) ? BM * KC_LD : BK * MC_LD;  // 4608

template <int LAY>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int r0, int rmax, int k0, int kend,
                                          int tid, float4 (&reg)[4]) {
  // LAY 0: rows r (0..127) x k (32) ; LAY 1: k rows (32) x cols r (128)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int f = tid + 256 * i;
    if (LAY == 0) {
      int row = f >> 3, c4 = f & 7;
      int gr = r0 + row, gk = k0 + c4 * 4;
      reg[i] = (gr < rmax && gk < kend) ? ld4(P + (long)gr * ld + gk) : zero4();
    } else {
      int krow = f >> 5, c4 = f & 31;
      int gk = k0 + krow, gr = r0 + c4 * 4;
      reg[i] = (gk < kend && gr + 3 < rmax) ? ld4(P + (long)gk * ld + gr) : zero4();
    }
  }
}
template <int LAY>
__device__ __forceinline__ void store_tile(float* __restrict__ S, int tid, const float4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int f = tid + 256 * i;
    if (LAY == 0) {
      int row = f >> 3, c4 = f & 7;
      st4(S + row * KC_LD + c4 * 4, reg[i]);
    } else {
      int krow = f >> 5, c4 = f & 31;
      st4(S + krow * MC_LD + c4 * 4, reg[i]);
    }
  }
}
// fragment for k chunk c (8 deep): out[j] = element k = 8c + 4h + j of tile row/col `idx`
template <int LAY>
__device__ __forceinline__ void read_frag(const float* __restrict__ S, int idx, int c, int h, float (&out)[4]) {
  if (LAY == 0) {
    float4 v = ld4(S + idx * KC_LD + c * 8 + 4 * h);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = S[(c * 8 + 4 * h + j) * MC_LD + idx];
  }
}

template <int LA, int LB>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware, bijective tile id remap (blocks b and b+8 share an XCD)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  int tm, tn;
  if (g.mt <= g.nt) { tn = id / g.mt; tm = id - tn * g.mt; } else { tm = id / g.nt; tn = id - tm * g.nt; }
  const int m0 = tm * BM, n0 = tn * BN;

  const int ks = blockIdx.z * g.kchunk;
  const int ke = min(g.K, ks + g.kchunk);
  const int nit = (ke - ks + BK - 1) / BK;
  float* Cs = g.C + (long)blockIdx.z * g.M * g.ldc;

  // rmax for loads: k-contiguous operands guard rows by M/N; m/n-contiguous guard columns by the leading dim
  // (values beyond M / N never reach a stored element; the bound only keeps 16-byte loads inside the rows)
  const int a_rmax = (LA == 0) ? g.M : min((int)g.lda, (g.M + 3) & ~3);
  const int b_rmax = (LB == 0) ? g.N : min((int)g.ldb, (g.N + 3) & ~3);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  float4 ra[4], rb[4];
  if (nit > 0) {
    load_tile<LA>(g.A, g.lda, m0, a_rmax, ks, ke, tid, ra);
    load_tile<LB>(g.B, g.ldb, n0, b_rmax, ks, ke, tid, rb);
    store_tile<LA>(smem, tid, ra);
    store_tile<LB>(smem + TILE_FLOATS, tid, rb);
  }
  __syncthreads();
  for (int it = 0; it < nit; ++it) {
    const float* As = smem + (it & 1) * 2 * TILE_FLOATS;
    const float* Bs = As + TILE_FLOATS;
    const bool more = (it + 1 < nit);
    if (more) {
      load_tile<LA>(g.A, g.lda, m0, a_rmax, ks + (it + 1) * BK, ke, tid, ra);
      load_tile<LB>(g.B, g.ldb, n0, b_rmax, ks + (it + 1) * BK, ke, tid, rb);
    }
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      float a[2][4], b[2][4];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        read_frag<LA>(As, wm * 64 + s * 32 + li, c, lh, a[s]);
        read_frag<LB>(Bs, wn * 64 + s * 32 + li, c, lh, b[s]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][j], b[t][j], acc[s][t], 0, 0, 0);
    }
    if (more) {
      float* An = smem + ((it + 1) & 1) * 2 * TILE_FLOATS;
      store_tile<LA>(An, tid, ra);
      store_tile<LB>(An + TILE_FLOATS, tid, rb);
    }
    __syncthreads();
  }

  // epilogue: D row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = n0 + wn * 64 + t * 32 + li;
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + s * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row < g.M) {
          float v = acc[s][t][e] + bv;
          if (g.act == 1) v = fmaxf(v, 0.f);
          else if (g.act == 2) v = tanhf(v);
          float* p = Cs + (long)row * g.ldc + col;
          if (g.beta) v += *p;
          *p = v;
        }
      }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int splitk, int M, int N,
                                                             long ld, float* __restrict__ out) {
  const long n4 = N >> 2;
  const long total = (long)M * n4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / n4, c = (i - row * n4) * 4;
    float4 s = zero4();
    for (int k = 0; k < splitk; ++k) s = add4(s, ld4(slabs + ((long)k * M + row) * ld + c));
    st4(out + row * ld + c, s);
  }
}

extern "C" int tcar_gemm_f32(int layout, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                             float* C, int64_t ldc, const float* bias, int act, int beta, int splitk, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return TCAR_OK;
  if (layout < 0 || layout > 2 || !A || !B || !C) return TCAR_E_ARG;
  if (!tcar_aligned16(A) || !tcar_aligned16(B) || (lda & 3) || (ldb & 3)) return TCAR_E_ARG;
  if (splitk < 1) splitk = 1;
  if (splitk > 1 && (bias || act || beta)) return TCAR_E_ARG;
  const bool a_kc = (layout != 2), b_kc = (layout == 1);
  if ((a_kc || b_kc) && (K & 3)) return TCAR_E_ARG;
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.act = act; g.beta = beta;
  int kchunk = (K + splitk - 1) / splitk;
  kchunk = ((kchunk + BK - 1) / BK) * BK;
  g.kchunk = kchunk;
  splitk = (K + kchunk - 1) / kchunk;
  g.mt = (M + BM - 1) / BM; g.nt = (N + BN - 1) / BN;
  dim3 grid(g.mt * g.nt, 1, splitk), block(256);
  const size_t lds = 4 * TILE_FLOATS * sizeof(float);  // 73,728 B
  hipStream_t st = (hipStream_t)stream;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_f32_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_f32_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_f32_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  if (layout == 0) TCAR_LAUNCH((gemm_f32_kernel<0, 1>), grid, block, lds, st, g);
  else if (layout == 1) TCAR_LAUNCH((gemm_f32_kernel<0, 0>), grid, block, lds, st, g);
  else TCAR_LAUNCH((gemm_f32_kernel<1, 1>), grid, block, lds, st, g);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// NOTE: with splitk > 1 the caller must size C as [splitk_eff, M, ldc] where splitk_eff <= splitk; slabs of
// unused splits are simply never written, so tcar_splitk_reduce must be given the effective count:
extern "C" int tcar_gemm_splitk_effective(int K, int splitk) {
  if (splitk < 1) splitk = 1;
  int kchunk = (K + splitk - 1) / splitk;
  kchunk = ((kchunk + BK - 1) / BK) * BK;
  return (K + kchunk - 1) / kchunk;
}

extern "C" int tcar_splitk_reduce(const float* slabs, int splitk, int M, int N, int64_t ld, float* out, void* stream) {
  if (M <= 0 || N <= 0) return TCAR_OK;
  if ((N & 3) || (ld & 3) || !tcar_aligned16(slabs) || !tcar_aligned16(out)) return TCAR_E_ARG;
  long total = (long)M * (N >> 2);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  TCAR_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, splitk, M, N, (long)ld, out);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
