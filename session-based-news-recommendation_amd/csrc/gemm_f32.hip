// fp32-input / fp32-accumulate MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, 256 FLOP/clk/CU).
//
// One kernel template covers the three operand layouts of the TCAR step (include/tcar_hip.h, tcar_gemm_f32):
//   LA = 0: A is k-contiguous  A[m*lda + k]      LA = 1: A is m-contiguous  A[k*lda + m]
//   LB = 0: B is k-contiguous  B[n*ldb + k]      LB = 1: B is n-contiguous  B[k*ldb + n]
// and a GROUP of independent problems per launch (tcar_gemm_f32_grouped): the step has ~25 tiny GEMMs whose
// cost is launch latency and a 16-workgroup grid, not FLOPs; grouping the independent ones (all weight
// gradients; all input projections) fills the chip with one launch.  A problem may also K-concatenate up to
// three (A, B) operand pairs into one accumulator ("segments": pre1 = X_ic W_in + X_c W_c + X_act W_int,
// modules.py:126-131) and may split K over workgroups (slabs for a deterministic reduce, or fp32 atomics
// into a pre-zeroed C for the weight gradients whose K is the batch).
//
// Tile TM x TN x 32 (128x128 for the full-catalog scoring GEMMs, 64x64 for the small ones), 256 threads =
// 4 waves in a 2x2 grid, each wave owns (TM/2)x(TN/2) as 32x32 MFMA tiles.  Operands are staged
// global -> registers -> LDS with the LDS image mirroring the global layout (every global access is a 16-byte,
// fully coalesced load):
//   k-contiguous tiles  [rows][32 k + 4 pad]  read back with ds_read_b128 (row stride 36 dwords: 36/4 odd
//                       => the 16-lane b128 groups hit 16 distinct 4-bank slots, conflict free)
//   m/n-contiguous tiles [32 k][cols + 4 pad] read back with ds_read_b32 (lanes = consecutive columns)
// MFMA operand pairing: the instruction contracts k in {0,1} = lane>>5.  For an 8-deep k chunk, lane half h
// holds k = 8c + 4h + j (j = 0..3) of BOTH operands, so MFMA j sums k = 8c+j and 8c+4+j: any pairing is valid
// as long as A and B agree, and it lets a k-contiguous operand be fetched with one b128 read per 4 MFMAs.
// LDS is double buffered; the next tile's global loads are in flight while the current one is multiplied.
// Workgroup ids are remapped so that each XCD (private L2) receives a contiguous run of tile ids, with the
// shorter tile dimension fastest: the workgroups that share the large operand's tile run on one XCD.
#include "tcar_common.h"
#include "tcar_bf16_layout.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BK = 32;
constexpr int MAXP = 10;      // problems per grouped launch
constexpr int MAXSEG = 3;     // K-concatenated operand pairs per problem

struct GemmProb {
  const float* A[MAXSEG];
  const float* B[MAXSEG];
  long lda[MAXSEG], ldb[MAXSEG];
  int K[MAXSEG];
  int nseg;
  float* C;
  long ldc;
  const float* bias;
  int M, N, act, beta;
  int ksplit, kchunk, mode;   // mode 0: plain, 1: slabs C[split][M][ldc], 2: atomicAdd into C
  int mt, nt, wg_begin;       // tiles and first workgroup id of this problem inside the launch
  // fused epilogues of the split-bf16 kernel (tcar_gemm_desc_t): activation backward + column sums; bf16 plane outputs
  int dact; const float* dact_y; long ld_dact_y; float* colsum;
  __bf16* plane_hi; __bf16* plane_lo; int plane_in32, plane_col0;
  __bf16* pack_hi; __bf16* pack_lo; int pack_in32, pack_c0, pack_c1;
};
struct GroupArgs {
  int nprob;
  int wg0[MAXP];            // first workgroup id of every problem (INT_MAX past nprob): ONE contiguous scalar load finds a
                            // workgroup's problem — a loop over p[i].wg_begin costs a dependent scalar-load round trip per problem
  GemmProb p[MAXP];
  TcarSignal sig;           // completion flag (gemm_x3_kernel only; zero = none)
};

template <int LAY, int ROWS>   // ROWS = tile extent along the m/n dimension (64 or 128)
struct TileGeo {
  static constexpr int KC_LD = BK + 4;
  static constexpr int MC_LD = ROWS + 4;
  static constexpr int FLOATS = (LAY == 0) ? ROWS * KC_LD : BK * MC_LD;
  static constexpr int NV = ROWS * BK / 4 / 256;   // float4 loads per thread (2 or 4)
};

template <int LAY, int ROWS>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int r0, int rmax, int k0, int kend,
                                          int tid, float4 (&reg)[TileGeo<LAY, ROWS>::NV]) {
#pragma unroll
  for (int i = 0; i < TileGeo<LAY, ROWS>::NV; ++i) {
    const int f = tid + 256 * i;
    if (LAY == 0) {
      const int row = f >> 3, c4 = f & 7;
      const int gr = r0 + row, gk = k0 + c4 * 4;
      reg[i] = (gr < rmax && gk < kend) ? ld4(P + (long)gr * ld + gk) : zero4();
    } else {
      constexpr int C4 = ROWS / 4;
      const int krow = f / C4, c4 = f - krow * C4;
      const int gk = k0 + krow, gr = r0 + c4 * 4;
      reg[i] = (gk < kend && gr + 3 < rmax) ? ld4(P + (long)gk * ld + gr) : zero4();
    }
  }
}
template <int LAY, int ROWS>
__device__ __forceinline__ void store_tile(float* __restrict__ S, int tid, const float4 (&reg)[TileGeo<LAY, ROWS>::NV]) {
#pragma unroll
  for (int i = 0; i < TileGeo<LAY, ROWS>::NV; ++i) {
    const int f = tid + 256 * i;
    if (LAY == 0) {
      const int row = f >> 3, c4 = f & 7;
      st4(S + row * TileGeo<LAY, ROWS>::KC_LD + c4 * 4, reg[i]);
    } else {
      constexpr int C4 = ROWS / 4;
      const int krow = f / C4, c4 = f - krow * C4;
      st4(S + krow * TileGeo<LAY, ROWS>::MC_LD + c4 * 4, reg[i]);
    }
  }
}
// fragment for k chunk c (8 deep): out[j] = element k = 8c + 4h + j of tile row/col `idx`
template <int LAY, int ROWS>
__device__ __forceinline__ void read_frag(const float* __restrict__ S, int idx, int c, int h, float (&out)[4]) {
  if (LAY == 0) {
    const float4 v = ld4(S + idx * TileGeo<LAY, ROWS>::KC_LD + c * 8 + 4 * h);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = S[(c * 8 + 4 * h + j) * TileGeo<LAY, ROWS>::MC_LD + idx];
  }
}

template <int LA, int LB, int TM, int TN>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GroupArgs ga) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int FM = TM / 64, FN = TN / 64;              // 32x32 MFMA tiles per wave along m / n
  constexpr int A_FL = TileGeo<LA, TM>::FLOATS, B_FL = TileGeo<LB, TN>::FLOATS;
  constexpr int STAGE = A_FL + B_FL;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware, bijective workgroup id remap (blocks b and b+8 share an XCD)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAXP; ++i) pi += (id >= ga.wg0[i]) ? 1 : 0;
  const GemmProb& g = ga.p[pi];
  id -= g.wg_begin;
  const int tiles = g.mt * g.nt;
  const int split = id / tiles;
  id -= split * tiles;
  int tm, tn;
  if (g.mt <= g.nt) { tn = id / g.mt; tm = id - tn * g.mt; } else { tm = id / g.nt; tn = id - tm * g.nt; }
  const int m0 = tm * TM, n0 = tn * TN;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  float4 ra[TileGeo<LA, TM>::NV], rb[TileGeo<LB, TN>::NV];
#pragma unroll 1
  for (int sg = 0; sg < g.nseg; ++sg) {
    const float* __restrict__ Ap = g.A[sg];
    const float* __restrict__ Bp = g.B[sg];
    const long lda = g.lda[sg], ldb = g.ldb[sg];
    const int Kseg = g.K[sg];
    const int ks = split * g.kchunk;
    const int ke = min(Kseg, ks + g.kchunk);
    const int nit = (ke - ks + BK - 1) / BK;
    // load bounds: k-contiguous operands guard rows by M/N; m/n-contiguous guard columns (values beyond M / N
    // never reach a stored element; the bound only keeps 16-byte loads inside the rows)
    const int a_rmax = (LA == 0) ? g.M : min((int)lda, (g.M + 3) & ~3);
    const int b_rmax = (LB == 0) ? g.N : min((int)ldb, (g.N + 3) & ~3);
    if (nit > 0) {
      load_tile<LA, TM>(Ap, lda, m0, a_rmax, ks, ke, tid, ra);
      load_tile<LB, TN>(Bp, ldb, n0, b_rmax, ks, ke, tid, rb);
      __syncthreads();                                   // previous segment finished reading stage 0
      store_tile<LA, TM>(smem, tid, ra);
      store_tile<LB, TN>(smem + A_FL, tid, rb);
    }
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
      const float* As = smem + (it & 1) * STAGE;
      const float* Bs = As + A_FL;
      const bool more = (it + 1 < nit);
      if (more) {
        load_tile<LA, TM>(Ap, lda, m0, a_rmax, ks + (it + 1) * BK, ke, tid, ra);
        load_tile<LB, TN>(Bp, ldb, n0, b_rmax, ks + (it + 1) * BK, ke, tid, rb);
      }
#pragma unroll
      for (int c = 0; c < BK / 8; ++c) {
        float a[FM][4], b[FN][4];
#pragma unroll
        for (int s = 0; s < FM; ++s) read_frag<LA, TM>(As, wm * (TM / 2) + s * 32 + li, c, lh, a[s]);
#pragma unroll
        for (int s = 0; s < FN; ++s) read_frag<LB, TN>(Bs, wn * (TN / 2) + s * 32 + li, c, lh, b[s]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int s = 0; s < FM; ++s)
#pragma unroll
            for (int t = 0; t < FN; ++t)
              acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][j], b[t][j], acc[s][t], 0, 0, 0);
      }
      if (more) {
        float* An = smem + ((it + 1) & 1) * STAGE;
        store_tile<LA, TM>(An, tid, ra);
        store_tile<LB, TN>(An + A_FL, tid, rb);
      }
      __syncthreads();
    }
  }

  // epilogue: D row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31
  float* Cs = g.C + (g.mode == 1 ? (long)split * g.M * g.ldc : 0L);
#pragma unroll
  for (int s = 0; s < FM; ++s)
#pragma unroll
    for (int t = 0; t < FN; ++t) {
      const int col = n0 + wn * (TN / 2) + t * 32 + li;
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * (TM / 2) + s * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row < g.M) {
          float* p = Cs + (long)row * g.ldc + col;
          if (g.mode == 2) {
            atomicAdd(p, acc[s][t][e]);
          } else {
            float v = acc[s][t][e] + bv;
            if (g.act == 1) v = fmaxf(v, 0.f);
            else if (g.act == 2) v = tanhf(v);
            if (g.beta) v += *p;
            *p = v;
          }
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

// ----------------------------------------------------------------------------------------------------------------
// Split-bf16 variant of the grouped kernel for the SMALL contractions (projections, output transforms, their
// gradients): same problems / segments / epilogues, fp32 operands in memory, but each staged tile is split on the fly
// into bf16 hi / lo planes (x = hi + lo) and multiplied with three v_mfma_f32_32x32x16_bf16 per product
// (hi*hi + hi*lo + lo*hi, fp32 accumulate: ~1e-5 relative, see gemm_bf16.hip).  These GEMMs are latency bound: a
// 64x64x832 tile takes 26 stages x 16 fp32 MFMAs x 64 cycles on one CU no matter how idle the chip is; the bf16 form
// needs 6 MFMAs x 32 cycles per stage.  Tile 64 x 64 x 32, 4 waves, one 32x32 MFMA tile per wave.
typedef __attribute__((ext_vector_type(8))) __bf16 xbf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 xbf16x4;
// The K loop of these launches is a latency chain (one workgroup per tile, <= 1 workgroup per CU): with 32-deep stages
// a K = 832 tile paid 26 dependent global-load round trips (~0.75 us each).  Stages are therefore 128 deep: a thread
// prefetches the whole next 64 x 128 (+ 128 x 64) fp32 slab into registers while the current one is multiplied.
// Launches with more workgroups than CUs use 64-deep stages instead: 48 KB of LDS per workgroup lets three of them share
// a CU and hide each other's load latency (a 128-deep stage set is 96 KB: one workgroup per CU).
constexpr int X3_MC = 192;           // bytes per k-row of an m/n-contiguous bf16 tile (128 + 64 pad: conflict-free tr reads)
template <int XK> struct X3 {
  static constexpr int KC = 2 * XK + 16;      // bytes per row of a k-contiguous bf16 tile (KC/16 odd: conflict-free b128)
  static constexpr int PLANE = XK * X3_MC;    // >= 64 * KC
  static constexpr int NV = 64 * XK / 4 / 256;   // float4 per thread, operand and stage
};

// The ring's loads are hidden from hipcc (inline asm, guide §5.7 form (ii)): with compiler-visible loads the wait in front
// of each stage's LDS store came out as s_waitcnt vmcnt(0) — the loop back-edge makes hipcc's counter bookkeeping
// conservative — which drains the whole ring once per round.  Every lane ALWAYS issues its loads (out-of-range lanes read
// the operand's first 16 bytes; the zero mask is applied at x3_store), so the per-stage count is exact and the waits are
// hand-counted: x3_wait<N>() = s_waitcnt vmcnt(N) naming every register of the slot as "+v", so that no consumer is
// scheduled above it.
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f gload16_hidden(const float* p) {
  v4f v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int N, int NV>
__device__ __forceinline__ void x3_wait(v4f (&a)[NV], v4f (&b)[NV]) {
  static_assert(NV == 2 || NV == 4 || NV == 8, "stage depth 32 / 64 / 128");
  if constexpr (NV == 2)
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
  else if constexpr (NV == 4)
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 : "n"(N) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]),
                   "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
                 : "n"(N) : "memory");
}

template <int LAY, int XK>
__device__ __forceinline__ unsigned x3_load(const float* __restrict__ P, long ld, int r0, int rmax, int k0, int kend, int tid,
                                            v4f (&reg)[X3<XK>::NV]) {
  unsigned mask = 0;
#pragma unroll
  for (int i = 0; i < X3<XK>::NV; ++i) {
    const int f = tid + 256 * i;
    bool ok;
    long off;
    if (LAY == 0) {
      const int row = f / (XK / 4), c4 = f % (XK / 4);
      const int gr = r0 + row, gk = k0 + c4 * 4;
      ok = gr < rmax && gk < kend;
      off = (long)gr * ld + gk;
    } else {
      const int krow = f >> 4, c4 = f & 15;
      const int gk = k0 + krow, gr = r0 + c4 * 4;
      ok = gk < kend && gr + 3 < rmax;
      off = (long)gk * ld + gr;
    }
    reg[i] = gload16_hidden(P + (ok ? off : 0L));
    mask |= ok ? (1u << i) : 0u;
  }
  return mask;
}
template <int LAY, int XK, bool HI = false>
__device__ __forceinline__ void x3_store(char* __restrict__ hi, char* __restrict__ lo, int tid, const v4f (&reg)[X3<XK>::NV],
                                         unsigned mask) {
#pragma unroll
  for (int i = 0; i < X3<XK>::NV; ++i) {
    const int f = tid + 256 * i;
    const bool ok = (mask >> i) & 1u;
    const float v[4] = {ok ? reg[i].x : 0.f, ok ? reg[i].y : 0.f, ok ? reg[i].z : 0.f, ok ? reg[i].w : 0.f};
    xbf16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = (__bf16)v[j]; l[j] = HI ? (__bf16)0.f : (__bf16)(v[j] - (float)h[j]); }
    int off;
    if (LAY == 0) { const int row = f / (XK / 4), c4 = f % (XK / 4); off = row * X3<XK>::KC + c4 * 8; }
    else { const int krow = f >> 4, c4 = f & 15; off = krow * X3_MC + c4 * 8; }
    *reinterpret_cast<xbf16x4*>(hi + off) = h;
    if (!HI) *reinterpret_cast<xbf16x4*>(lo + off) = l;
  }
}
// fragment of the 32-row block at `base`, k16 sub-step s of the XK-deep stage
template <int LAY, int XK>
__device__ __forceinline__ xbf16x8 x3_frag(const char* __restrict__ S, int base, int s, int lane) {
  if (LAY == 0) {
    return *reinterpret_cast<const xbf16x8*>(S + (base + (lane & 31)) * X3<XK>::KC + s * 32 + (lane >> 5) * 16);
  } else {
    const int g = lane >> 4, i = lane & 15;
    const int mbase = 16 * (g & 1), kbase = 8 * (g >> 1), q = i >> 2, p = i & 3;
    const char* a0 = S + (s * 16 + kbase + q) * X3_MC + (base + mbase + 4 * p) * 2;
    typedef __attribute__((address_space(3))) xbf16x4* lds_p;
    const xbf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a0));
    const xbf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a0 + 4 * X3_MC));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// Register prefetch ring: the K loop of these launches is a latency chain (<= 1 workgroup per CU, a global-load round
// trip of ~1-2 us per stage under load), so each thread keeps XD stages of both operands in flight in registers and the
// (segment, k) stages of a K-concatenated problem form ONE flattened sequence — the ring runs across segment boundaries.
// HI: plain bf16 operands (the hi planes only), ONE MFMA per product — the session-side backward GEMMs of the hi-only backward precision
template <int LA, int LB, int XK, int XD, bool HI = false>
__global__ __launch_bounds__(256) void gemm_x3_kernel(const GroupArgs ga) {
  constexpr int X3_PLANE = X3<XK>::PLANE, X3_NV = X3<XK>::NV;
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  char* smem = reinterpret_cast<char*>(smem_f);     // A hi, A lo, B hi, B lo (one stage)
  constexpr int T = 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAXP; ++i) pi += (id >= ga.wg0[i]) ? 1 : 0;
  const GemmProb& g = ga.p[pi];
  id -= g.wg_begin;
  const int tiles = g.mt * g.nt;
  const int split = id / tiles;
  id -= split * tiles;
  int tm, tn;
  if (g.mt <= g.nt) { tn = id / g.mt; tm = id - tn * g.mt; } else { tm = id / g.nt; tn = id - tm * g.nt; }
  const int m0 = tm * T, n0 = tn * T;

  // flattened stage sequence: stage s of segment sg covers k in [ks + (s - c[sg]) * XK, ...) clipped to ke[sg]
  const int ks = split * g.kchunk;
  int cst[MAXSEG + 1];
  cst[0] = 0;
#pragma unroll
  for (int sg = 0; sg < MAXSEG; ++sg) {
    int n = 0;
    if (sg < g.nseg) {
      const int ke = min(g.K[sg], ks + g.kchunk);
      n = ke > ks ? (ke - ks + XK - 1) / XK : 0;
    }
    cst[sg + 1] = cst[sg] + n;
  }
  const int total = cst[MAXSEG];
  const int a_rmax0 = (LA == 0) ? g.M : (g.M + 3) & ~3;
  const int b_rmax0 = (LB == 0) ? g.N : (g.N + 3) & ~3;

  // segment of stage s; stages >= total map to the last segment with k0 >= its end (every lane masked)
  auto stage_seg = [&](int s) { return min((s >= cst[1] ? 1 : 0) + (s >= cst[2] ? 1 : 0), g.nseg - 1); };
  auto load_stage = [&](int s, v4f (&ra)[X3_NV], v4f (&rb)[X3_NV], unsigned& ma, unsigned& mb) {
    const int sg = stage_seg(s);
    const int k0 = ks + (s - cst[sg]) * XK;
    const int ke = min(g.K[sg], ks + g.kchunk);
    const long lda = g.lda[sg], ldb = g.ldb[sg];
    const int a_rmax = (LA == 0) ? a_rmax0 : min((int)lda, a_rmax0);
    const int b_rmax = (LB == 0) ? b_rmax0 : min((int)ldb, b_rmax0);
    ma = x3_load<LA, XK>(g.A[sg], lda, m0, a_rmax, k0, ke, tid, ra);
    mb = x3_load<LB, XK>(g.B[sg], ldb, n0, b_rmax, k0, ke, tid, rb);
  };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // The ring body is STRAIGHT-LINE code: every slot is refilled unconditionally, in the same order, every round (stages
  // past the end load the operand's first bytes with an all-zero mask and are never multiplied).  Any branch around a load
  // makes hipcc fall back to s_waitcnt vmcnt(0), which would drain the ring once per round.
  v4f ra[XD][X3_NV], rb[XD][X3_NV];
  unsigned ma[XD], mb[XD];
#pragma unroll
  for (int d = 0; d < XD; ++d) load_stage(d, ra[d], rb[d], ma[d], mb[d]);
#pragma unroll 1
  for (int base = 0; base < total; base += XD) {
#pragma unroll
    for (int d = 0; d < XD; ++d) {
      const int s = base + d;
      __syncthreads();                                        // everyone is done reading the previous stage
      x3_wait<(XD - 1) * 2 * X3_NV, X3_NV>(ra[d], rb[d]);      // this slot has landed; XD - 1 younger stages stay in flight
      x3_store<LA, XK, HI>(smem, smem + X3_PLANE, tid, ra[d], ma[d]);
      x3_store<LB, XK, HI>(smem + 2 * X3_PLANE, smem + 3 * X3_PLANE, tid, rb[d], mb[d]);
      __syncthreads();
      load_stage(s + XD, ra[d], rb[d], ma[d], mb[d]);         // refill this ring slot: XD - 1 stages stay in flight
      const int sg = stage_seg(s);
      const int k0 = ks + (s - cst[sg]) * XK;
      const int left = min(g.K[sg], ks + g.kchunk) - k0;
      const int nsub = s < total ? (min(XK, left + 15) >> 4) : 0;     // k16 sub-steps that hold data
#pragma unroll 2
      for (int u = 0; u < nsub; ++u) {
        const xbf16x8 ah = x3_frag<LA, XK>(smem, wm * 32, u, lane), bh = x3_frag<LB, XK>(smem + 2 * X3_PLANE, wn * 32, u, lane);
        if constexpr (!HI) {
          const xbf16x8 al = x3_frag<LA, XK>(smem + X3_PLANE, wm * 32, u, lane), bl = x3_frag<LB, XK>(smem + 3 * X3_PLANE, wn * 32, u, lane);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the ring's trailing (masked) loads, before compiler-counted accesses
  float* Cs = g.C + (g.mode == 1 ? (long)split * g.M * g.ldc : 0L);
  const bool wt = ga.sig.cnt != nullptr;
  const int col = n0 + wn * 32 + (lane & 31);
  float csum = 0.f;
  if (col < g.N) {
    const float bv = g.bias ? g.bias[col] : 0.f;
    // packed-plane column of this lane: columns [pack_c0, pack_c1) are left out, later columns move down
    const int gcol = g.plane_col0 + col;               // column in plane space
    const bool packed = g.pack_hi && (gcol < g.pack_c0 || gcol >= g.pack_c1);
    const int pcol = gcol < g.pack_c0 ? gcol : gcol - (g.pack_c1 - g.pack_c0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (row < g.M) {
        float* p = Cs + (long)row * g.ldc + col;
        if (g.mode == 2) {
          atomicAdd(p, acc[e]);
        } else {
          float v = acc[e] + bv;
          if (g.act == 1) v = fmaxf(v, 0.f);
          else if (g.act == 2) v = tanhf(v);
          if (g.dact) {                                   // activation backward: times act'(y)
            const float y = g.dact_y[(long)row * g.ld_dact_y + col];
            v = g.dact == 1 ? (y > 0.f ? v : 0.f) : v * (1.f - y * y);
            csum += v;
          }
          if (g.beta) v += *p;
          // a launch that carries a completion flag stores its result write-through (sc1): the consumer behind the flag needs no
          // L2 write-back then (tcar_common.h)
          if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else *p = v;
          if (g.plane_hi) {
            const __bf16 h = (__bf16)v, l = (__bf16)(v - (float)h);
            const long o = kb32_off(row, gcol, g.plane_in32);
            g.plane_hi[o] = h;
            g.plane_lo[o] = l;
            if (packed) {
              const long po = kb32_off(row, pcol, g.pack_in32);
              g.pack_hi[po] = h;
              g.pack_lo[po] = l;
            }
          }
        }
      }
    }
  }
  if (g.dact && g.colsum) {      // column sums (bias gradient): the two lane halves hold the same column
    csum += __shfl_xor(csum, 32);
    if (lane < 32 && col < g.N && csum != 0.f) atomicAdd(g.colsum + col, csum);
  }
  tcar_signal_done(ga.sig);
}

namespace {

template <int LA, int LB, int TM, int TN>
int launch_variant(const GroupArgs& ga, int nwg, hipStream_t st) {
  constexpr size_t lds = 2 * (TileGeo<LA, TM>::FLOATS + TileGeo<LB, TN>::FLOATS) * sizeof(float);
  TCAR_SET_LDS_ONCE((gemm_f32_kernel<LA, LB, TM, TN>), lds);
  TCAR_LAUNCH((gemm_f32_kernel<LA, LB, TM, TN>), dim3(nwg), dim3(256), lds, st, ga);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

template <int LA, int LB>
int launch_layout(GroupArgs& ga, hipStream_t st) {
  // tile choice: 128x128 when that alone fills the chip, 64x64 otherwise
  long big = 0;
  for (int i = 0; i < ga.nprob; ++i)
    big += (long)((ga.p[i].M + 127) / 128) * ((ga.p[i].N + 127) / 128) * ga.p[i].ksplit;
  const int T = big >= 192 ? 128 : 64;
  int wg = 0;
  for (int i = 0; i < ga.nprob; ++i) {
    GemmProb& p = ga.p[i];
    p.mt = (p.M + T - 1) / T;
    p.nt = (p.N + T - 1) / T;
    p.wg_begin = wg;
    ga.wg0[i] = wg;
    wg += p.mt * p.nt * p.ksplit;
  }
  for (int i = ga.nprob; i < MAXP; ++i) ga.wg0[i] = 0x7fffffff;
  if (T == 128) return launch_variant<LA, LB, 128, 128>(ga, wg, st);
  return launch_variant<LA, LB, 64, 64>(ga, wg, st);
}

template <int LA, int LB, int XK, int XD, bool HI = false>
int launch_x3_v(GroupArgs& ga, int wg, hipStream_t st, TcarOpt* o) {
  constexpr size_t lds = 4 * X3<XK>::PLANE;
  // A launch that carries a completion flag stores C write-through (agent-scope stores); its bf16 plane / pack outputs would
  // still be plain stores, which a consumer on another XCD could read stale behind the flag: refused, not silently allowed.
  if (o && o->sig.cnt)
    for (int i = 0; i < ga.nprob; ++i)
      if (ga.p[i].plane_hi || ga.p[i].pack_hi) return TCAR_E_ARG;
  ga.sig = tcar_sig(o);
  TCAR_SET_LDS_ONCE((gemm_x3_kernel<LA, LB, XK, XD, HI>), lds);
  TCAR_LAUNCH((gemm_x3_kernel<LA, LB, XK, XD, HI>), dim3(wg), dim3(256), lds, st, ga);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

template <int LA, int LB>
int launch_x3(GroupArgs& ga, hipStream_t st, TcarOpt* o) {
  int wg = 0;
  for (int i = 0; i < ga.nprob; ++i) {
    GemmProb& p = ga.p[i];
    p.mt = (p.M + 63) / 64;
    p.nt = (p.N + 63) / 64;
    p.wg_begin = wg;
    ga.wg0[i] = wg;
    wg += p.mt * p.nt * p.ksplit;
  }
  for (int i = ga.nprob; i < MAXP; ++i) ga.wg0[i] = 0x7fffffff;
  // 64-deep stages (48 KB of LDS): these launches run beside the dE GEMM, whose workgroups hold 96 KB of a CU's 160 KB — a
  // 96-KB (128-deep) workgroup would have to wait for one of them to retire (measured: 0.704 / 0.690 / 0.680 ms per step with
  // 128-deep / mixed / 64-deep stages).  A register ring of 2-3 stages does not pay for long K (profiles/r02_x3_small_gemm_bench.txt:
  // a stage costs 1.3 us of in-workgroup work), so the default keeps ONE stage in flight.
  // Short-K launches (every workgroup has at most two 64-deep stages: the split-K "one-shot" problems of the step driver,
  // K <= 128): both stages are requested up front (ring of 2) — a workgroup pays ONE global-memory round trip.
  int max_stages = 0;
  for (int i = 0; i < ga.nprob; ++i) {
    const GemmProb& p = ga.p[i];
    int n = 0;
    for (int sg = 0; sg < p.nseg; ++sg) n += ((p.K[sg] < p.kchunk ? p.K[sg] : p.kchunk) + 63) / 64;
    max_stages = n > max_stages ? n : max_stages;
  }
  // (the ring of 2 for every launch of at most tcar_fixed::x3_oneshot stages per workgroup)
  if constexpr (LA == LB)        // (the backward layouts 1 and 2: plain bf16 operands where the caller asks — TcarOpt::hi_only)
    if (o && o->hi_only)
      return max_stages <= tcar_fixed::x3_oneshot ? launch_x3_v<LA, LB, 64, 2, true>(ga, wg, st, o) : launch_x3_v<LA, LB, 64, 1, true>(ga, wg, st, o);
  if (max_stages <= tcar_fixed::x3_oneshot) return launch_x3_v<LA, LB, 64, 2>(ga, wg, st, o);
  // (diagnostic builds, tcar_fixed::x3_deep: a long-K launch that is at most one workgroup per CU walks HALF as many, 128-deep
  //  stages — 96 KB of LDS per workgroup: only where no dE workgroup holds the CU's LDS, i.e. the step's tail)
  if constexpr (tcar_fixed::x3_deep != 0)
    if (wg <= 256) return launch_x3_v<LA, LB, 128, 1>(ga, wg, st, o);
  return launch_x3_v<LA, LB, 64, 1>(ga, wg, st, o);
}

int fill_prob(GemmProb& p, int layout, const tcar_gemm_desc_t& d) {
  if (d.M <= 0 || d.N <= 0 || d.nseg < 1 || d.nseg > MAXSEG || !d.C) return TCAR_E_ARG;
  const bool a_kc = (layout != 2), b_kc = (layout == 1);
  p.nseg = d.nseg;
  for (int s = 0; s < d.nseg; ++s) {
    if (!d.A[s] || !d.B[s] || d.K[s] <= 0 || !tcar_aligned16(d.A[s]) || !tcar_aligned16(d.B[s]) || (d.lda[s] & 3) ||
        (d.ldb[s] & 3))
      return TCAR_E_ARG;
    if ((a_kc || b_kc) && (d.K[s] & 3)) return TCAR_E_ARG;
    p.A[s] = d.A[s]; p.B[s] = d.B[s]; p.lda[s] = d.lda[s]; p.ldb[s] = d.ldb[s]; p.K[s] = d.K[s];
  }
  p.C = d.C; p.ldc = d.ldc; p.bias = d.bias; p.M = d.M; p.N = d.N; p.act = d.act; p.beta = d.beta;
  p.dact = d.dact; p.dact_y = d.dact_y; p.ld_dact_y = d.ld_dact_y; p.colsum = d.colsum;
  if (d.dact && (d.dact < 1 || d.dact > 2 || !d.dact_y)) return TCAR_E_ARG;
  p.plane_hi = (__bf16*)d.plane_hi; p.plane_lo = (__bf16*)d.plane_lo; p.plane_in32 = d.plane_inner >> 5; p.plane_col0 = d.plane_col0;
  p.pack_hi = (__bf16*)d.pack_hi; p.pack_lo = (__bf16*)d.pack_lo; p.pack_in32 = d.pack_inner >> 5;
  p.pack_c0 = d.pack_c0; p.pack_c1 = d.pack_c1;
  if (d.plane_hi && (!d.plane_lo || (d.plane_inner & 31) || d.plane_col0 < 0 || d.plane_col0 + d.N > d.plane_inner)) return TCAR_E_ARG;
  if (d.pack_hi && (!d.plane_hi || !d.pack_lo || (d.pack_inner & 31) || d.pack_c0 > d.pack_c1)) return TCAR_E_ARG;
  if ((d.dact || d.plane_hi) && d.splitk > 1) return TCAR_E_ARG;
  int splitk = d.splitk < 1 ? 1 : d.splitk;
  p.mode = splitk > 1 ? (d.atomic ? 2 : 1) : 0;
  if (splitk > 1 && (d.nseg != 1 || d.bias || d.act || d.beta)) return TCAR_E_ARG;
  int kmax = 0;
  for (int s = 0; s < d.nseg; ++s) kmax = d.K[s] > kmax ? d.K[s] : kmax;
  int kchunk = (kmax + splitk - 1) / splitk;
  kchunk = ((kchunk + BK - 1) / BK) * BK;
  p.kchunk = kchunk;
  p.ksplit = (kmax + kchunk - 1) / kchunk;
  if (p.ksplit == 1) p.mode = (splitk > 1 && d.atomic) ? 2 : 0;
  return TCAR_OK;
}

}  // namespace

extern "C" int tcar_gemm_f32_grouped(int layout, int nprob, const tcar_gemm_desc_t* descs, void* stream) {
  if (nprob <= 0) return TCAR_OK;
  if (layout < 0 || layout > 2 || nprob > MAXP || !descs) return TCAR_E_ARG;
  GroupArgs ga;
  ga.nprob = 0;
  for (int i = 0; i < nprob; ++i) {
    if (descs[i].M <= 0 || descs[i].N <= 0) continue;        // empty problems are no-ops
    const int rc = fill_prob(ga.p[ga.nprob], layout, descs[i]);
    if (rc) return rc;
    ga.nprob++;
  }
  if (ga.nprob == 0) return TCAR_OK;
  hipStream_t st = (hipStream_t)stream;
  if (layout == 0) return launch_layout<0, 1>(ga, st);
  if (layout == 1) return launch_layout<0, 0>(ga, st);
  return launch_layout<1, 1>(ga, st);
}

extern "C" int tcar_gemm_x3_grouped(int layout, int nprob, const tcar_gemm_desc_t* descs, void* stream) {
  return tcar_gemm_x3_grouped_o(layout, nprob, descs, stream, nullptr);
}
int tcar_gemm_x3_grouped_o(int layout, int nprob, const tcar_gemm_desc_t* descs, void* stream, TcarOpt* o) {
  if (nprob <= 0) return TCAR_OK;
  if (layout < 0 || layout > 2 || nprob > MAXP || !descs) return TCAR_E_ARG;
  GroupArgs ga;
  ga.nprob = 0;
  for (int i = 0; i < nprob; ++i) {
    if (descs[i].M <= 0 || descs[i].N <= 0) continue;
    const int rc = fill_prob(ga.p[ga.nprob], layout, descs[i]);
    if (rc) return rc;
    ga.nprob++;
  }
  if (ga.nprob == 0) return TCAR_OK;
  hipStream_t st = (hipStream_t)stream;
  if (layout == 0) return launch_x3<0, 1>(ga, st, o);
  if (layout == 1) return launch_x3<0, 0>(ga, st, o);
  return launch_x3<1, 1>(ga, st, o);
}

extern "C" int tcar_gemm_f32(int layout, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                             float* C, int64_t ldc, const float* bias, int act, int beta, int splitk, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return TCAR_OK;
  tcar_gemm_desc_t d = {};
  d.nseg = 1;
  d.A[0] = A; d.B[0] = B; d.lda[0] = lda; d.ldb[0] = ldb; d.K[0] = K;
  d.C = C; d.ldc = ldc; d.bias = bias; d.M = M; d.N = N; d.act = act; d.beta = beta; d.splitk = splitk; d.atomic = 0;
  return tcar_gemm_f32_grouped(layout, 1, &d, stream);
}

// with splitk > 1 (slab mode) C is [splitk_eff, M, ldc]; K is cut in multiples of 32, so fewer slabs than
// requested may be written — tcar_splitk_reduce must be given this effective count:
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
