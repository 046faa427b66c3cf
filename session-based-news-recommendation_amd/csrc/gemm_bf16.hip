// Split-bf16 MFMA GEMM for gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the full-catalog scoring
// contractions of model_combine.py:138 and their two gradients at ~16x the fp32 matrix rate.
//
// gfx950 has no TF32/xf32 path, so fp32 operands are carried as TWO bf16 planes, x = hi + lo with
// hi = bf16(x), lo = bf16(x - hi) (relative residual <= 2^-17), and a product is three MFMAs
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi                  (NSPLIT = 3, dropped term <= 2^-16 |a||b|)
// accumulated in fp32: logits and gradients stay within ~1e-5 of the fp32 path (tests keep the 1e-3 gate) at
// 3/16 of its MFMA time.  NSPLIT = 1 uses the hi planes only (plain bf16, ~4e-3 relative).
//
// Planes live in the KB32 blocked layout (tcar_bf16_layout.h): every stage of every operand tile is a handful of
// contiguous 8-KB / 2-KB chunks.  Staging is DIRECT global -> LDS DMA (global_load_lds_dwordx4: 1 KB per wave
// instruction, no VGPR round trip, no ds_write; measured on the register-staged predecessor: the data path alone
// took 118 us of a 165 us launch and its fragment-shaped 64-byte reads ran at ~1/3 of the L2 rate), double
// buffered: the next stage's DMA is in flight while the current one is multiplied; __syncthreads() drains it
// (s_waitcnt vmcnt(0) + s_barrier), which is exactly the visibility rule for LDS-DMA data.
// Workgroup tile (32*TMW*WMW) x (32*TNW*WNW) x 32 with WMW x WNW waves of TMW x TNW MFMA tiles (32 x 32).  The kernel is
// bound by the L2 -> LDS fill rate (PMC: MFMA busy 41-44 % of the SIMD cycles), so each GEMM takes the tile with the
// fewest fill bytes per flop that LDS (160 KB, two stages) and the register file allow and that fills the chip evenly:
// logits 256 x 384 (8 waves of 4 x 3 tiles), dX 512 x 128 (16 waves), dE 192 x 192 (12 waves of 1 x 3 tiles);
// 256 x 256 / 256 x 192 / 256 x 128 / 128 x 128 for other shapes (launch_b).  Operand modes:
//   MODE 0  k-contiguous: fragment (row = lane&31, k = 8*(lane>>5)+0..7) is one swizzled ds_read_b128;
//   MODE 1  m/n-contiguous: fragments come from ds_read_b64_tr_b16, the gfx950 transposing LDS read (per 16-lane group
//           it returns, to lane i, column i of a 4 (k) x 16 (col) block): no transposed copy of E or dlogits exists.
// XCD-aware tile remap as in gemm_f32.hip; split-K (slabs) serves the catalog-long contraction of dX.
#include <type_traits>
#include <cstdlib>
#include "tcar_common.h"
#include "tcar_bf16_layout.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int KB = 32;

struct BArgs {
  const __bf16* A[2];
  const __bf16* B[2];
  int a_in32, b_in32;          // inner dimension / 32 of the A / B planes
  int a_rb, b_rb;              // allocated 128-row blocks of the A / B planes
  float* C; long ldc;
  float* C2; long ldc2; int csplit;     // columns >= csplit go to C2 (column index rebased); csplit >= N: unused
  int M, N, K, kchunk, mode, mt, nt, nsk;
  int n_fastest;               // tile order inside the XCD-contiguous id run: 1 = consecutive ids walk the N tiles
  const int32_t* perm;         // optional grouped row permutation of the C2 destination (see tcar_gemm_bf16), else NULL
  int pgroup;                  // columns per permutation group
  // softmax epilogue (EPI = 1, tcar_gemm_bf16_ce): instead of C the kernel writes, per row and per column GROUP (the 32 * TNW
  // columns one wave owns), the group maximum and the sum of exp(x - max), the exponentials themselves as a bf16 KB32 plane,
  // and the label's score
  // second K segment (SEG2 = 1, logits layout only): the contraction continues over [K1, K) with OTHER planes — A2 hi / lo
  // (inner a2_in32 * 32) and ONE B2 plane (inner b2_in32 * 32; its elements are exact in bf16: the one-hot time-index matrix),
  // so that block costs two MFMAs per product and half the B fill bytes
  const __bf16* A2[2]; const __bf16* B2; int a2_in32, b2_in32, a2_rb, b2_rb, K1;
  TcarSignal sig;      // completion flag (softmax-epilogue form only)
  __bf16* p_hi; int p_in32;    // plane [ceil128(M), 32 * p_in32]
  float* stats; int ngroups;   // [M, ngroups, 2]
  const int32_t* label; float* lab_logit;
  int lab_off, lab_window;     // TcarOpt: column of the label = label[m] - lab_off; window: outside [0, N) = not in this shard
  int anchored;                // TcarOpt: anchored epilogue — the accumulators are x - anchor[m] already: exp(acc), no group maximum;
                               // statistics (sum of the plane's ROUNDED entries, sum of the exponentials) per group
  // dX of the one-hot form (layout 0, hi planes only): N tiles at or beyond column n_b2 (a multiple of the tile width) read their B
  // operand from the plane B2 (inner b2_in32 * 32, rows as B) at column n0 - n_b2 — the static one-hot matrix of
  // publish_time_MWDHM, so that those output columns are dP = dlogits OH instead of dlogits E_time.  0: unused.
  int n_b2;
  // dE with the (q, z) epilogue (EPI = 2, tcar_gemm_bf16_de_qz): output columns >= csplit are the candidate-side time block; it is
  // not stored — per catalog row n and table k the kernel leaves q = ||gy||^2 and z = clip(table row) . gy of the 64-column
  // gradient gy = dE[n, csplit + 64 k ...] at qz[perm[k * M + n]] (perm: position in the inverted index of publish_time_MWDHM)
  const int32_t* mwdhm; const float* tclip; float2* qz;
  // START flag (TcarOpt::start): workgroup 0 publishes start_epoch as its first action
  unsigned* start_flag; unsigned start_epoch;
};
// rows of the month | day | week | hour | minute tables (model_combine.py:73-81) inside their concatenation
__device__ __forceinline__ int cand_row(const int32_t* __restrict__ mwdhm, long n, int k) {
  const int voc = k == 0 ? 13 : k == 1 ? 32 : k == 2 ? 8 : k == 3 ? 25 : 61;
  const int off = k == 0 ? 0 : k == 1 ? 13 : k == 2 ? 45 : k == 3 ? 53 : 78;
  return off + clampi(mwdhm[n * 5 + k], 0, voc - 1);
}

typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(1))) const void* glb_vp;

// fragment of the 32-row block starting at tile row/col `base`, k16 sub-step s (0/1) of the 32-deep stage.
// PERM (k-contiguous operands only): MFMA row i = 8 a + 4 h + j takes tile row 16 h + 4 a + j instead of row i.  Fed as the
// FIRST operand of a 32x32 MFMA this makes a lane's 16 accumulator registers 16 CONSECUTIVE tile rows (16 (lane >> 5) + e) instead
// of four runs of four — the softmax epilogue stores 16 bytes per lane.  The swizzled image stays conflict-free: each 16-lane
// group of ds_read_b128 ({0-3, 12-15, 20-27}, ...) still meets all four values of (row >> 2) & 3.
template <int MODE, bool PERM = false>
__device__ __forceinline__ bf16x8 frag(const char* __restrict__ S, int base, int s, int lane) {
  if (MODE == 0) {
    // (128-row blocks of 8 KB, 64-byte rows: (rr >> 7) * 8192 + (rr & 127) * 64 == rr * 64 — one linear address per lane, the tile
    //  and plane offsets fold into the instruction's immediate)
    const int l5 = lane & 31;
    const int rr = base + (PERM ? ((((l5 >> 2) & 1) << 4) | ((l5 >> 3) << 2) | (l5 & 3)) : l5);
    const int piece = (s * 2 + (lane >> 5)) ^ ((rr >> 2) & 3);
    return *reinterpret_cast<const bf16x8*>(S + rr * 64 + piece * 16);
  } else {
    const int g = lane >> 4, i = lane & 15;
    const int q = i >> 2, p = i & 3;
    const int col = base + 16 * (g & 1) + 4 * p, row = s * 16 + 8 * (g >> 1) + q;
    const int cc = col & 31, pc = cc >> 3;
    const char* a0 = S + (col >> 5) * 2048 + row * 64 + (cc & 7) * 2;
    typedef __attribute__((address_space(3))) bf16x4* lds_p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a0 + ((pc ^ ((row >> 2) & 3)) << 4)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a0 + 256 + ((pc ^ (((row + 4) >> 2) & 3)) << 4)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// TMW x TNW = MFMA tiles (32 x 32) per wave along M / N: 2 x 2 for the 4-waves-per-SIMD shapes, 4 x 3 (12 accumulator
// tiles = 192 registers, 2 waves per SIMD) for the 256 x 384 workgroup tile of the logits GEMM
// KS = 32-deep k blocks per LDS stage: the hi-only (NSPLIT = 1) form has a third of the MFMA work between two barriers
// and half the bytes per stage, so it takes 64-deep stages (same LDS as the two-plane form, half the barriers).
// VAR: 0 = the product kernel.  Diagnostic forms of the softmax-epilogue kernel (built only with -DTCAR_GEMM_DIAG, selected by
// TCAR_BF16_TILE = 387 / 388 / 393 / 395; tools/gemm_variants.sh): 2 = no fills after stage 1 (compute side alone), 3 = fills and
// barriers only, 6 = epilogue only, 8 = K loop without the epilogue.  Round 4: 97.5 / 49 / 57 / 74 us of a 102-us launch — the K
// loop runs at ~1.29 PF executed, the practical bf16 rate of the chip on random data; the epilogue was 30 us of the launch.
// NST = LDS stages of the ring (round 6).  2: double buffer — the copies of stage t + 1 are issued at the top of iteration t and must
// have landed at its end (s_waitcnt vmcnt(0) + s_barrier), so an iteration cannot be shorter than one L2 -> LDS round trip: a short
// K loop of small stages (dE: K = 512 sessions = 16 stages of 24 KB) is paced by that latency, not by the MFMAs or the fill rate.
// 3: the copies of stage t + 2 are issued at the top of iteration t and the wave waits only for ITS copies of stage t + 1 (counted
// vmcnt: the copies of one stage are a wave-uniform count, the same for every stage of a K segment) — two iterations of slack per copy.
template <int MA, int MB, int NSPLIT, int WMW, int WNW, int TMW = 2, int TNW = 2, int KS = 1, int EPI = 0, int SEG2 = 0, int VAR = 0, int NST = 2>
// (EPI = 2, the dE form with the (q, z) epilogue: two 9-wave workgroups per CU need five waves per SIMD, i.e. <= 96 registers per
//  lane; without the bound the compiler takes 102 — four waves per SIMD, ONE workgroup per CU, 64 us alone.  At 80 registers, which
//  two 12-wave workgroups would need, it spills)
__global__ __launch_bounds__(64 * WMW * WNW, (EPI == 2 ? (WMW * WNW == 9 ? 5 : 4) : 1)) void gemm_bf16_kernel(const BArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = WMW * WNW, TM = 32 * TMW * WMW, TN = 32 * TNW * WNW;
  constexpr int NP = (NSPLIT == 1) ? 1 : 2;                 // planes per operand
  constexpr int A_BYTES = TM * 64, B_BYTES = TN * 64;       // one plane of one operand, one stage
  constexpr int PL = A_BYTES + B_BYTES;                     // LDS stage = [plane][A | B]
  constexpr int SUB = NP * PL;                              // one 32-deep k block of both operands
  constexpr int STAGE = KS * SUB;
  constexpr int A_CP = A_BYTES / 1024, B_CP = B_BYTES / 1024;   // 1-KB wave copies per plane
  constexpr int NCOPY = NP * (A_CP + B_CP);
  constexpr int CPW = (NCOPY + NW - 1) / NW;                // copies per wave and stage (the last round may be partial)
  static_assert((MA == 1 || TM % 128 == 0) && (MB == 1 || TN % 128 == 0), "k-contiguous operand tiles are whole 128-row blocks");
  // Every LDS-DMA destination of this kernel lies inside its own allocation, for EVERY instantiation: a copy writes 1 KB at
  // stage * STAGE + substage * SUB + plane * PL + (A: ci < A_CP | B: A_BYTES + ci < B_CP) * 1024, i.e. below 2 * STAGE — exactly the
  // dynamic LDS every launcher requests (launch_k / tcar_gemm_bf16_de_qz_o: 2 * KS * NP * (TM + TN) * 64).  (VERDICT r04 item 2:
  // hypothesis "a destination beyond the stage" — ruled out by construction.)
  static_assert(A_BYTES % 1024 == 0 && B_BYTES % 1024 == 0 && NCOPY * 1024 == SUB, "1-KB copies tile a sub-stage exactly");
  static_assert(NST * STAGE == NST * KS * NP * (TM + TN) * 64 && NST * STAGE <= 160 * 1024, "NST stages = the launch's dynamic LDS <= 160 KB");
  static_assert(NST == 2 || (NST == 3 && SEG2 == 0 && KS == 1), "the three-stage ring: one K segment, one k block per stage");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave - wm * WNW;
  if (EPI == 0 && g.start_flag && blockIdx.x == 0 && tid == 0)
    __hip_atomic_store(g.start_flag, g.start_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // 1-D grid over (split, tile): the bijective XCD remap hands each XCD a contiguous run of logical ids, and the
  // tile order inside the run is chosen by the caller so that the workgroups which re-read the SAME large operand
  // tile are neighbours on one XCD (one HBM/MALL fetch, the rest L2 hits): logits -> M fastest (share the E tile),
  // dX (split-K) and dE -> N fastest (share the dlogits tile)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tiles = g.mt * g.nt;
  const int split = id / tiles;
  id -= split * tiles;
  int tm, tn;
  if (g.n_fastest) { tm = id / g.nt; tn = id - tm * g.nt; } else { tn = id / g.mt; tm = id - tn * g.mt; }
  const int m0 = tm * TM, n0 = tn * TN;
  // (workgroup-uniform) this N tile reads the second B plane: dX columns of the one-hot time block
  const bool b2any = (SEG2 == 0 && NSPLIT == 1 && MB == 1) && g.n_b2 > 0 && n0 + TN > g.n_b2;
  // (wave-uniform, EPI = 2) this wave's 64 output columns are one table's block of the candidate-side time gradient: the two
  // operand roles are SWAPPED for it — fragments of the B tile go in as the MFMA's first operand — so that its accumulators come
  // out transposed (a lane owns one catalog row, its registers run over the 64 columns) and the row reductions of the epilogue
  // are in-register sums.  Both operands of this layout are read with the same transposing fragment loader: the swap is two
  // base pointers, the K loop is unchanged.
  const bool tw = (EPI == 2) && (n0 + wn * (32 * TNW) >= g.csplit);
  static_assert(EPI != 2 || (MA == 1 && MB == 1 && TMW == TNW && TNW == 2 && NSPLIT == 1), "EPI = 2: dE layout, 64 x 64 per wave, hi planes");
  const int a_off = tw ? A_BYTES : 0, b_off = tw ? 0 : A_BYTES;
  const int a_base = tw ? wn * (32 * TNW) : wm * (32 * TMW), b_base = tw ? wm * (32 * TMW) : wn * (32 * TNW);
  const int ks = split * g.kchunk;
  const int ke = min(g.K, ks + g.kchunk);
  const int nkb = (ke - ks) / KB;
  const int nit = (VAR == 6) ? 0 : (nkb + KS - 1) / KS;

  // (EPI = 2, time-block waves) the epilogue's two dependent index loads — the table row each of this lane's catalog rows looks up and
  // its position in the inverted index — are issued HERE, ahead of the K loop, so that only the clipped-row loads follow the loop
  int qz_row[TNW], qz_pos[TNW];
  if constexpr (EPI == 2) {
#pragma unroll
    for (int t2 = 0; t2 < TNW; ++t2) {
      qz_row[t2] = 0; qz_pos[t2] = -1;
      if (tw) {
        const int k = (n0 + wn * (32 * TNW) - g.csplit) >> 6;
        const long n = m0 + wm * (32 * TMW) + t2 * 32 + (lane & 31);
        if (n < g.M) { qz_row[t2] = cand_row(g.mwdhm, n, k); qz_pos[t2] = g.perm[(long)k * g.M + n]; }
      }
    }
  }
  f32x16 acc[TMW][TNW];
#pragma unroll
  for (int a = 0; a < TMW; ++a)
#pragma unroll
    for (int b = 0; b < TNW; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // ---- this wave's share of the DMA copies.  Everything about a copy except its position along K is fixed for a K segment, so it is
  // worked out ONCE (copy_setup: plane, row / column block, validity, LDS slot) and kept as a wave-uniform source pointer that
  // advances by one k-block per use.  (Round 4: the previous form recomputed plane selection, block indices, a 64-bit product and
  // the validity test for each of the ~10 copies in EVERY stage — ~400 scalar instructions, 30 kernel-argument loads and 77 spilled
  // SGPR reads per 72 MFMAs in the logits loop, on the one scalar unit the CU's waves share.)
  //   k-contiguous operand (MODE 0): the next k-block is the next 8-KB block: + 4096 elements
  //   transposed-read operand (MODE 1): + 1024 elements inside a 128-row block, + in32 * 4096 - 3072 when the block is used up
  const __bf16* cp_src[CPW];
  int cp_dst[CPW];                 // byte offset of the copy inside a sub-stage, -1: nothing to copy
  int cp_wrap[CPW];                // increment when the k-block index crosses a multiple of 4 (MODE 1), else = the plain one
  int kq_next = 0;                 // k-block index (k0 / 32, segment relative) of the next sub-stage to issue
  auto copy_setup = [&](bool seg2, int k0) {
    kq_next = k0 >> 5;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int c = wave + NW * i;                     // copy index in [0, NCOPY)   (wave-uniform)
      cp_dst[i] = -1; cp_src[i] = g.A[0]; cp_wrap[i] = 0;
      if (NCOPY % NW != 0 && c >= NCOPY) continue;
      const int p = c / (A_CP + B_CP), rem = c - p * (A_CP + B_CP);
      const bool isA = rem < A_CP;
      const int ci = isA ? rem : rem - A_CP;
      if (SEG2 && seg2 && !isA && p == 1) continue;    // the second segment's B operand has ONE plane
      // (one-hot dX: the B operand of output columns >= n_b2 is the plane B2.  Decided per COPY — a copy of this layout is one
      //  32-column block — so a tile may straddle n_b2: the 384-column tiles of the large-catalog form, n_b2 = 512)
      const bool b2copy = b2any && !isA && (n0 + (ci >> 1) * 32 >= g.n_b2);
      const __bf16* P = (SEG2 && seg2) ? (isA ? g.A2[p] : g.B2) : (isA ? g.A[p] : (b2copy ? g.B2 : g.B[p]));
      const int in32 = (SEG2 && seg2) ? (isA ? g.a2_in32 : g.b2_in32) : (isA ? g.a_in32 : (b2copy ? g.b2_in32 : g.b_in32));
      const int nrb = (SEG2 && seg2) ? (isA ? g.a2_rb : g.b2_rb) : (isA ? g.a_rb : g.b_rb);
      const int mode = isA ? MA : MB, t0 = isA ? m0 : (b2copy ? n0 - g.n_b2 : n0);
      long src;
      bool ok;
      if (mode == 0) {          // k-contiguous: 8 copies per 8-KB block (rows t0.., inner block k0/32)
        const int rb = (t0 >> 7) + (ci >> 3);
        ok = rb < nrb;
        src = ((long)rb * in32 + (k0 >> 5)) * 4096 + (ci & 7) * 512;
        cp_wrap[i] = 4096;
      } else {                  // transposed-read: 2 copies per 2-KB chunk (32 rows k0.. of inner block t0/32 + j)
        const int cb = (t0 >> 5) + (ci >> 1);
        ok = cb < in32;
        src = ((long)(k0 >> 7) * in32 + cb) * 4096 + (k0 & 127) * 32 + (ci & 1) * 512;
        cp_wrap[i] = in32 * 4096 - 3072;
      }
      if (ok) {
        cp_src[i] = P + src;
        cp_dst[i] = p * PL + (isA ? 0 : A_BYTES) + ci * 1024;
      }
    }
  };
  // one sub-stage (32-deep k-block) into the LDS image at St; advances every source pointer to the next k-block
  auto copy_substage = [&](char* St) {
    const bool wrap = (kq_next & 3) == 3;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      if (cp_dst[i] >= 0) __builtin_amdgcn_global_load_lds((glb_vp)(cp_src[i] + lane * 8), (lds_vp)(St + cp_dst[i]), 16, 0, 0);
      const int mode_inc = (MA == 0 && MB == 0) ? 4096 : ((cp_wrap[i] == 4096) ? 4096 : (wrap ? cp_wrap[i] : 1024));
      cp_src[i] += mode_inc;
    }
    ++kq_next;
  };
  // issue this wave's share of the DMA copies of k-stage `t` into LDS stage buffer t & 1
  auto issue = [&](int t) {
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      if (KS > 1 && t * KS + j >= nkb) break;
      copy_substage(smem + (t & 1) * STAGE + j * SUB);
    }
  };
  // the MFMAs of one 32-column k-block; S2: the block lies in the second K segment (B has no lo plane: two MFMAs per product, and
  // its lo fragments are not read).  Two fully unrolled bodies behind ONE uniform branch per k-block: a test around single MFMAs
  // would break the interleaving of LDS reads and MFMAs.
  auto block = [&](const char* St, auto s2) __attribute__((always_inline)) {
    constexpr bool S2 = decltype(s2)::value;
    constexpr int NPB = S2 ? 1 : NP;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[NP][TMW];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int u = 0; u < TMW; ++u) a[p][u] = frag<MA>(St + p * PL + (EPI == 2 ? a_off : 0), (EPI == 2 ? a_base : wm * (32 * TMW)) + u * 32, s, lane);
#pragma unroll
      for (int t2 = 0; t2 < TNW; ++t2) {          // one B tile at a time: its fragments die after TMW * NSPLIT MFMAs
        bf16x8 b[NPB];
#pragma unroll
        for (int p = 0; p < NPB; ++p) b[p] = frag<MB, EPI == 1>(St + p * PL + (EPI == 2 ? b_off : A_BYTES), (EPI == 2 ? b_base : wn * (32 * TNW)) + t2 * 32, s, lane);
#pragma unroll
        for (int u = 0; u < TMW; ++u) {
          if constexpr (EPI == 1) {     // transposed accumulator tile (rows = catalog columns, lane = session): see the epilogue
            if constexpr (NSPLIT == 3) {
              acc[u][t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], a[NP - 1][u], acc[u][t2], 0, 0, 0);
              if constexpr (!S2) acc[u][t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[NPB - 1], a[0][u], acc[u][t2], 0, 0, 0);
            }
            acc[u][t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], a[0][u], acc[u][t2], 0, 0, 0);
          } else {
            if constexpr (NSPLIT == 3) {
              acc[u][t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NP - 1][u], b[0], acc[u][t2], 0, 0, 0);
              if constexpr (!S2) acc[u][t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][u], b[NPB - 1], acc[u][t2], 0, 0, 0);
            }
            acc[u][t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][u], b[0], acc[u][t2], 0, 0, 0);
          }
        }
      }
    }
  };
  auto compute = [&](int t, auto s2) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      if (KS > 1 && t * KS + j >= nkb) break;
      block(smem + (t & 1) * STAGE + j * SUB, s2);
    }
  };

  // two loops, one per K segment (the second is empty without SEG2; with it KS = 1 and there is no split-K): each has ONE body
  const int nit1 = SEG2 ? min(nit, g.K1 / KB) : nit;
  copy_setup(SEG2 && nit1 == 0, (SEG2 && nit1 == 0) ? 0 : ks);
  if constexpr (NST == 3) {
    // ---- three-stage ring.  ncp = this wave's copies per stage (wave-uniform; KS = 1: one sub-stage per stage).  At the end of
    // iteration t the wave lets the ncp copies of stage t + 2 stay in flight and waits for its older ones (stage t + 1), then the
    // barrier makes every wave's share of stage t + 1 visible AND retires buffer t % 3, which iteration t + 1 refills with stage t + 3.
    int ncp = 0;
#pragma unroll
    for (int i = 0; i < CPW; ++i) ncp += (cp_dst[i] >= 0) ? 1 : 0;
    ncp = __builtin_amdgcn_readfirstlane(ncp);
    auto wait_keep = [&](int keep) __attribute__((always_inline)) {      // s_waitcnt vmcnt(keep), keep wave-uniform in [0, CPW]
      static_assert(CPW <= 6, "extend wait_keep");
      if (keep <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (CPW < 2 || keep == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if (CPW < 3 || keep == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (CPW < 4 || keep == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (CPW < 5 || keep == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (CPW < 6 || keep == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    };
    if (nit > 0) copy_substage(smem);
    if (nit > 1) copy_substage(smem + STAGE);
    wait_keep(nit > 1 ? ncp : 0);
    __builtin_amdgcn_s_barrier();
    int cur = 0, fill = 2 * STAGE;         // byte offsets of the buffer being multiplied / the one refilled next (wave-uniform)
    for (int it = 0; it < nit; ++it) {
      const bool more = it + 2 < nit;
      if (more) copy_substage(smem + fill);      // this buffer was last read in iteration it - 1, which ended with a barrier
      block(smem + cur, std::false_type{});
      wait_keep(more ? ncp : 0);           // stage it + 1 has landed for this wave (stage it + 2 may still be in flight)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS reads of the current buffer have returned
      __builtin_amdgcn_s_barrier();
      fill = cur;
      cur = (cur == 2 * STAGE) ? 0 : cur + STAGE;
    }
  } else {
  if (nit > 0) issue(0);
  __syncthreads();                       // s_waitcnt vmcnt(0) + s_barrier: stage 0 has landed for every wave
  for (int it = 0; it < nit1; ++it) {
    if (it + 1 < nit && (VAR != 2 || it < 1)) {      // buffer (it+1)&1 was last read in iteration it-1, which ended with a barrier
      if (SEG2 && it + 1 == nit1) copy_setup(true, 0);      // the next stage is the first of the second K segment
      issue(it + 1);
    }
    if (VAR != 3) compute(it, std::false_type{});
    __syncthreads();
  }
  if constexpr (SEG2 != 0) {
    for (int it = nit1; it < nit; ++it) {
      if (it + 1 < nit && VAR != 2) issue(it + 1);
      if (VAR != 3) compute(it, std::true_type{});
      __syncthreads();
    }
  }
  }      // NST == 2

  const int li = lane & 31, lh = lane >> 5;
  if constexpr (VAR == 8) {      // DIAGNOSTIC: K loop without the epilogue (the accumulators must stay live)
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < TMW; ++u)
#pragma unroll
      for (int b2 = 0; b2 < TNW; ++b2)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[u][b2][e];
    if (t == 12345.678f) g.C[0] = t;
    if constexpr (EPI == 1) tcar_signal_done(g.sig);
    return;
  }
  if constexpr (EPI == 1) {
#if defined(TCAR_DIAG_DOT2) && TCAR_DIAG_DOT2 >= 2      // (diagnostic: no wave of the workgroup issues an MFMA once any wave is in its epilogue)
    __builtin_amdgcn_s_barrier();
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
#endif
    // Softmax epilogue (model_combine.py:145 without materialised logits).  The accumulators of this form are TRANSPOSED
    // (compute() swaps the MFMA operands) and the B fragments are read with the row permutation of frag<., PERM>: a lane owns ONE
    // session (column lane & 31 of the tile = row m of the logits) and its 16 registers of catalog tile t are the 16 CONSECUTIVE
    // columns n = 32 t + 16 (lane >> 5) + e of the wave's GW = 32 * TNW.  Group maximum and sum are in-register loops plus ONE
    // v_permlane32_swap between the two half waves, and the exponentials leave as 16-byte stores of eight consecutive catalog columns
    // (a wave instruction fills 32 bytes of 32 plane rows; round 4: the 8-byte form took 15 us of a 102-us launch).
    // exp(x - group max) <= 1 goes out as bf16; tcar_ce_finish combines the (max, sum) pairs of a row into its log-sum-exp and rescales
    // the plane to softmax - onehot.  Interior workgroups (every row and column of the tile inside M x N: all but the last column of
    // tiles) run a body without bounds tests; the label's score is picked by a scan only in waves where a lane's label falls into
    // its columns (a 7 % event at the Globo shape).
    constexpr int GW = 32 * TNW;
    constexpr float LOG2E = 1.4426950408889634f;
    const int gidx = (n0 + wn * GW) / GW;
    const int nb = n0 + wn * GW + 16 * lh;
    const int pcols = g.p_in32 << 5;
    const bool interior = (m0 + TM <= g.M) && (n0 + TN <= g.N) && (n0 + TN <= pcols);      // workgroup-uniform
    auto xhalf = [](float x, auto op) __attribute__((always_inline)) {      // op(x of this lane, x of lane ^ 32): the same bits in both
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
      return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    };
    auto epi = [&](auto fast_) __attribute__((always_inline)) {
      constexpr bool FAST = decltype(fast_)::value;
#pragma unroll
      for (int u = 0; u < TMW; ++u) {
        const int row = m0 + wm * (32 * TMW) + u * 32 + li;
        const bool live = FAST || row < g.M;
        float mx = -INFINITY;
        if (g.anchored) {      // anchored form (workgroup-uniform): the reference was subtracted inside the contraction
          mx = 0.f;
        } else {
#pragma unroll
          for (int t = 0; t < TNW; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              if (FAST || nb + 32 * t + e < g.N) mx = fmaxf(mx, acc[u][t][e]);
          mx = xhalf(mx, [](float a, float b) { return fmaxf(a, b); });
        }
        // (clamped like tcar_ce_finish: lab_logit is always written; with a label window — catalog shard — a label outside it matches
        //  no column and lab_logit[row] is left alone)
        const int lraw = live ? g.label[row] - g.lab_off : -1;
        const int lab = !live ? -1 : g.lab_window ? ((lraw >= 0 && lraw < g.N) ? lraw : -1) : clampi(lraw, 0, g.N - 1);
        const int d = lab - nb;                        // the label among this lane's columns: tile d >> 5, register d & 31 (< 16)
        const bool has_lab = lab >= 0 && d >= 0 && d < GW && (d & 16) == 0;
        if (__any(has_lab)) {
          float labv = 0.f;
#pragma unroll
          for (int t = 0; t < TNW; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) labv = (32 * t + e == d) ? acc[u][t][e] : labv;
          if (has_lab) g.lab_logit[row] = labv;
        }
        const float c = -mx * LOG2E;
        // anchored form: TWO sums per group — of the exponentials (the loss: lse = anchor + log S) and of their bf16 ROUNDINGS as the
        // plane holds them (the gradient: plane / S_r sums to one exactly, so that the label's e_l - S_r is the true -(sum of the others).
        // With group maxima the dominant element is exp(0) = 1, exact in bf16; an anchored e_l = exp(x_l - a) is not, and near
        // convergence (p_l -> 1) its 2^-9 rounding against an unrounded S was the whole of the label's gradient entry: two engines'
        // runs of one overfit toy drifted 500x further apart than with group maxima until the sums were made consistent)
        float sum = 0.f, sumr = 0.f;
#pragma unroll
        for (int t = 0; t < TNW; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int n8 = nb + 32 * t + 8 * h;
            bf16x8 pk;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              // (the exponent is CLAMPED at 2^100: a no-op with group maxima, where it is <= 0; in the anchored form a logit more
              //  than 69 nats above its row's anchor — a per-session loss > 69 — saturates instead of overflowing, so the plane, the
              //  sums and every gradient stay finite; such a row's largest probabilities are shared out evenly among the clamped)
              const float pe = (FAST || n8 + j < g.N) ? __builtin_amdgcn_exp2f(fminf(fmaf(acc[u][t][8 * h + j], LOG2E, c), 100.f)) : 0.f;
              sum += pe;
              pk[j] = (__bf16)pe;
            }
            {      // the eight ROUNDED values back out of the four packed dwords (the exponentials themselves are dead by now: converting
              //      inside the loop above kept them alive and spilled 80 registers)
              // (v_dot2c_f32_bf16 against a pair of ones: with the pairs BIT-CAST out of the packed dwords hipcc 7.2 feeds all four
              //  instructions of a piece the same register — fast, spill-free and wrong (profiles/r06_dot2c_in_step.txt); with the
              //  pairs taken from the vector's elements it is right, spills 76 registers and is 4 us slower.  Shifts, masks and adds.)
              typedef unsigned u4_t __attribute__((ext_vector_type(4)));
              const u4_t w = __builtin_bit_cast(u4_t, pk);
#ifdef TCAR_DIAG_DOT2      // (diagnostic builds only — tools/micro/build_x3ring.sh dot2 | dot2b | dot2c)
              typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
              const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
#if TCAR_DIAG_DOT2 >= 3    // pairs taken from the vector's ELEMENTS
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const bf16x2_t pr = {pk[2 * q], pk[2 * q + 1]};
                sumr = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, sumr, false);
              }
#else                      // pairs bit-cast out of the packed dwords: hipcc 7.2 feeds all four v_dot2c of a piece the SAME register
#pragma unroll
              for (int q = 0; q < 4; ++q) sumr = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w[q]), ones, sumr, false);
#endif
#else
#pragma unroll
              for (int q = 0; q < 4; ++q) sumr += __uint_as_float(w[q] << 16) + __uint_as_float(w[q] & 0xffff0000u);
#endif
            }
            if (VAR == 7 ? (sum == 12345.678f) : (live && (FAST || n8 < pcols)))
              *reinterpret_cast<bf16x8*>(g.p_hi + kb32_off(row, n8, g.p_in32)) = pk;
          }
        sum = xhalf(sum, [](float a, float b) { return a + b; });
        if (g.anchored) mx = xhalf(sumr, [](float a, float b) { return a + b; });      // (the reference slot carries the rounded sum)
        if (live && lh == 0) *reinterpret_cast<float2*>(g.stats + ((long)row * g.ngroups + gidx) * 2) = make_float2(mx, sum);
        __builtin_amdgcn_sched_barrier(0);     // one session tile at a time: the accumulators leave no room for hoisted addresses
      }
    };
    if (interior) epi(std::true_type{}); else epi(std::false_type{});
    tcar_signal_done(g.sig);
    return;
  }
  if constexpr (EPI == 2) {
    if (tw) {
      // (q, z) epilogue of a time-block wave.  acc[u][t2][e]: lane li owns catalog row n of row block t2, its registers run over the
      // table's columns c = 32 u + 4 lh + (e & 3) + 8 (e >> 2).  The Jacobian of max_norm = 1 needs, per (n, k), only
      // q = ||gy||^2 and z = x . gy against the clipped table row x the candidate looked up (embed.hip: cand_time_bwd_onehot).
#pragma unroll
      for (int t2 = 0; t2 < TNW; ++t2) {
        const bool live = qz_pos[t2] >= 0;
        const float* xr = g.tclip + (long)qz_row[t2] * 64 + 4 * lh;
        float q = 0.f, z = 0.f;
#pragma unroll
        for (int u = 0; u < TMW; ++u)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float4 xv = ld4(xr + 32 * u + 8 * q4);
            const float v0 = acc[u][t2][4 * q4], v1 = acc[u][t2][4 * q4 + 1], v2 = acc[u][t2][4 * q4 + 2], v3 = acc[u][t2][4 * q4 + 3];
            q = fmaf(v0, v0, q); q = fmaf(v1, v1, q); q = fmaf(v2, v2, q); q = fmaf(v3, v3, q);
            z = fmaf(v0, xv.x, z); z = fmaf(v1, xv.y, z); z = fmaf(v2, xv.z, z); z = fmaf(v3, xv.w, z);
          }
        q += __shfl_xor(q, 32);
        z += __shfl_xor(z, 32);
        if (live && lh == 0) g.qz[qz_pos[t2]] = make_float2(q, z);
      }
      return;
    }
  }
  float* C1 = g.C + (g.mode == 1 ? (long)split * g.M * g.ldc : 0L);
#pragma unroll
  for (int u = 0; u < TMW; ++u)
#pragma unroll
    for (int t = 0; t < TNW; ++t) {
      const int col = n0 + wn * (32 * TNW) + t * 32 + li;
      if (col >= g.N) continue;
      const int row0 = m0 + wm * (32 * TMW) + u * 32 + 4 * lh;
      if (g.perm && col >= g.csplit) {
        // C2 element (m, cc) lives at C2[perm[(cc / pgroup) * M + m] * pgroup + cc % pgroup]: the 32 lanes of a half wave
        // share m, so the permutation load is a broadcast
        const int cc = col - g.csplit;
        const int32_t* pp = g.perm + (long)(cc / g.pgroup) * g.M;
        float* pb = g.C2 + cc % g.pgroup;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + (e & 3) + 8 * (e >> 2);
          if (row < g.M) pb[(long)pp[row] * g.pgroup] = acc[u][t][e];
        }
        continue;
      }
      float* base = (col < g.csplit) ? C1 + col : g.C2 + (col - g.csplit);
      const long ld = (col < g.csplit) ? g.ldc : g.ldc2;
      if (m0 + TM <= g.M) {            // interior tile (workgroup-uniform): 16 unguarded stores, no per-element branch
        float* pr = base + (long)row0 * ld;
#pragma unroll
        for (int e = 0; e < 16; ++e) pr[(long)((e & 3) + 8 * (e >> 2)) * ld] = acc[u][t][e];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + (e & 3) + 8 * (e >> 2);
          if (row < g.M) base[(long)row * ld] = acc[u][t][e];
        }
      }
    }
}

// (Round 5 built this GEMM on v_mfma_f32_16x16x32_bf16 as well — same tile, pair-permuted catalog fragments, 16-byte plane stores —
// and measured it 2-4 us SLOWER in the step: 40 instead of 28 fragment reads per k block.  Removed in round 6; the kernel and its
// measurements are in git history (2c5f4a0) and profiles/r05_ab_experiments.txt.)

// fp32 [rows, cols] (ld) -> bf16 hi / lo planes in the KB32 layout with inner dimension in16 (>= cols, % 32 == 0);
// rows [rows, ceil128(rows)) and columns [cols, in16) are zero filled.  Optional second, PACKED output pair taking
// columns [0, c0) U [c1, cols) (inner dimension pin16) — the dE operand: item | time blocks of attout.
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, long ld, int rows, int cols,
                                                         __bf16* __restrict__ hi, __bf16* __restrict__ lo, int in16,
                                                         __bf16* __restrict__ phi, __bf16* __restrict__ plo, int pin16,
                                                         int c0, int c1) {
  const long c4n = in16 >> 2;
  const long rpad = ((long)rows + 127) & ~127L;
  const long total = rpad * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (r < rows && c + j < cols) ? x[r * ld + c + j] : 0.f;
    bf16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = (__bf16)v[j];
      l[j] = (__bf16)(v[j] - (float)h[j]);
    }
    const long o = kb32_off(r, c, in16 >> 5);
    *reinterpret_cast<bf16x4*>(hi + o) = h;
    if (lo) *reinterpret_cast<bf16x4*>(lo + o) = l;
    if (phi && (c < c0 || c >= c1) && c < cols) {
      const long po = kb32_off(r, c < c0 ? c : c - (c1 - c0), pin16 >> 5);
      *reinterpret_cast<bf16x4*>(phi + po) = h;
      if (plo) *reinterpret_cast<bf16x4*>(plo + po) = l;
    }
  }
}

// host-side state of ONE launch call (on the caller's stack): launch options (tuning copy, completion flag), the dry-run sink of
// tcar_gemm_bf16_variant (when set the chosen instantiation is named instead of launched) and the group geometry a
// softmax-epilogue launch reports back to tcar_gemm_bf16_ce
struct LaunchCall {
  TcarOpt* o = nullptr;
  char* variant_out = nullptr;
  int variant_len = 0;
  int ce_gw = 0, ce_ngroups = 0;
};

template <int MA, int MB, int NSPLIT, int WMW, int WNW, int TMW, int TNW, int KS, int VAR = 0, int NST = 2>
int launch_k(BArgs& g, int splitk, hipStream_t st, LaunchCall& lc) {
  constexpr int NT = 64 * WMW * WNW, TM = 32 * TMW * WMW, TN = 32 * TNW * WNW, NP = (NSPLIT == 1) ? 1 : 2;
  constexpr size_t lds = NST * KS * NP * (TM + TN) * 64;
  g.mt = (g.M + TM - 1) / TM;
  g.nt = (g.N + TN - 1) / TN;
  if (lc.variant_out) {
    snprintf(lc.variant_out, lc.variant_len, "gemm_bf16_kernel<%d, %d, %d, %d, %d, %d, %d> tile %dx%dx%d grid %d", MA, MB, NSPLIT,
             WMW, WNW, TMW, TNW, TM, TN, 32 * KS, g.mt * g.nt * splitk);
    return TCAR_OK;
  }
  if (g.p_hi) {      // softmax epilogue: logits layout only
    if constexpr (MA == 0 && MB == 0) {
      if (splitk != 1 || g.C2 != g.C) return TCAR_E_ARG;
      g.ngroups = g.nt * WNW;
      lc.ce_gw = 32 * TNW;
      lc.ce_ngroups = g.ngroups;
      g.sig = tcar_sig(lc.o);
      g.lab_off = lc.o ? lc.o->lab_off : 0;
      g.lab_window = lc.o ? lc.o->lab_window : 0;
      g.anchored = (lc.o && lc.o->anchored) ? 1 : 0;
      if (g.B2) {
        if constexpr (NSPLIT == 3 && KS == 1) {
          TCAR_SET_LDS_ONCE((gemm_bf16_kernel<MA, MB, NSPLIT, WMW, WNW, TMW, TNW, KS, 1, 1, VAR>), lds);
          TCAR_LAUNCH((gemm_bf16_kernel<MA, MB, NSPLIT, WMW, WNW, TMW, TNW, KS, 1, 1, VAR>), dim3(g.mt * g.nt), dim3(NT), lds, st, g);
          TCAR_CHECK_LAUNCH();
          return TCAR_OK;
        } else {
          return TCAR_E_ARG;
        }
      }
      TCAR_SET_LDS_ONCE((gemm_bf16_kernel<MA, MB, NSPLIT, WMW, WNW, TMW, TNW, KS, 1>), lds);
      TCAR_LAUNCH((gemm_bf16_kernel<MA, MB, NSPLIT, WMW, WNW, TMW, TNW, KS, 1>), dim3(g.mt * g.nt), dim3(NT), lds, st, g);
      TCAR_CHECK_LAUNCH();
      return TCAR_OK;
    } else {
      return TCAR_E_ARG;
    }
  }
  TCAR_SET_LDS_ONCE((gemm_bf16_kernel<MA, MB, NSPLIT, WMW, WNW, TMW, TNW, KS, 0, 0, VAR, NST>), lds);
  TCAR_LAUNCH((gemm_bf16_kernel<MA, MB, NSPLIT, WMW, WNW, TMW, TNW, KS, 0, 0, VAR, NST>), dim3(g.mt * g.nt * splitk), dim3(NT), lds, st, g);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

template <int MA, int MB, int NSPLIT, int WMW, int WNW, int TMW = 2, int TNW = 2, int VAR = 0>
int launch_v(BArgs& g, int splitk, hipStream_t st, LaunchCall& lc) {
  // measured at the Globo shape (hi-only backward): dX 68 -> 59 us with 64-deep stages; dE (192 x 192 tiles, three
  // workgroups per CU at 48 KB) loses its occupancy with 96 KB and slows down 129 -> 136 us, so it keeps 32-deep stages
  // (TCAR_BF16_KS: 1 = never, 2 = k-contiguous A operand only, 3 = always)
  if constexpr (NSPLIT == 1) {
    const int ks = tcar_tn(lc.o).bf16_ks;
    // (TCAR_BF16_KS=4: the 256 x 128 dX tile as a three-stage ring of 32-deep stages — 72 KB instead of 96 KB of LDS, two
    //  iterations of slack per copy instead of one)
    if constexpr (MA == 0 && MB == 1 && WMW == 4 && WNW == 2 && TMW == 2 && TNW == 2 && VAR == 0) {
      if (ks == 4 && !g.p_hi) return launch_k<MA, MB, NSPLIT, WMW, WNW, TMW, TNW, 1, 0, 3>(g, splitk, st, lc);
    }
    if (ks == 3 || (ks == 2 && MA == 0)) return launch_k<MA, MB, NSPLIT, WMW, WNW, TMW, TNW, 2, VAR>(g, splitk, st, lc);
  }
  return launch_k<MA, MB, NSPLIT, WMW, WNW, TMW, TNW, 1, VAR>(g, splitk, st, lc);
}

template <int MA, int MB>
int launch_b(BArgs& g, int nsplit, int splitk, hipStream_t st, LaunchCall& lc) {
  // Tile choice.  The kernel is bound by the per-CU load path (~70 GB/s from L2): bytes per flop fall with the tile
  // area/perimeter ratio, so take the largest tile that still gives the chip about a full wave of workgroups:
  // 256 x 256 (16 waves, 64 KB per stage), then 256 x 128 (8 waves), else 128 x 128 (4 waves).
  const int f0 = tcar_tn(lc.o).bf16_tile;
  const int f = (f0 == 1922 || f0 == 1923 || f0 == 1283 || f0 == 2562) ? 0 : f0;      // (codes of the dE (q, z) launcher only: heuristic here)
  const long w256 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256) * splitk;
  const long w128 = (long)((g.M + 255) / 256) * ((g.N + 127) / 128) * splitk;
  if constexpr (MA == 0 && MB == 0) {
    // 256 x 384 (8 waves, 4 x 3 MFMA tiles each; 160 KB of LDS): the kernel is bound by the L2 -> LDS fill rate, and this
    // shape moves 17 % fewer bytes per flop than 256 x 256; at the Globo catalog it is also ONE round of 240 workgroups
    // instead of 360 workgroups in 1.4 rounds
    const long w384 = (long)((g.M + 255) / 256) * ((g.N + 383) / 384) * splitk;
#ifdef TCAR_GEMM_DIAG
    if (nsplit == 3 && f == 387) return launch_v<0, 0, 3, 2, 4, 4, 3, 2>(g, splitk, st, lc);
    if (nsplit == 3 && f == 388) return launch_v<0, 0, 3, 2, 4, 4, 3, 3>(g, splitk, st, lc);
    if (nsplit == 3 && f == 393) return launch_v<0, 0, 3, 2, 4, 4, 3, 6>(g, splitk, st, lc);
    if (nsplit == 3 && f == 394) return launch_v<0, 0, 3, 2, 4, 4, 3, 7>(g, splitk, st, lc);
    if (nsplit == 3 && f == 395) return launch_v<0, 0, 3, 2, 4, 4, 3, 8>(g, splitk, st, lc);
#endif
    if (nsplit == 3 && (f == 384 || (f == 0 && w384 >= 200))) return launch_v<0, 0, 3, 2, 4, 4, 3>(g, splitk, st, lc);
  }
  if constexpr (MA == 0 && MB == 1) {
    // 512 x 128 (16 waves): the whole session batch is ONE M tile, so every dlogits stage is fetched once per N tile and
    // the fill bytes per flop drop 16 % against two 256 x 128 tiles (dX: 170 -> 147 us at split-K 36)
    const long w512 = (long)((g.M + 511) / 512) * ((g.N + 127) / 128) * splitk;
#ifdef TCAR_GEMM_DIAG
    if (nsplit == 1 && f == 1002) return launch_v<0, 1, 1, 8, 2, 2, 2, 2>(g, splitk, st, lc);
    if (nsplit == 1 && f == 1003) return launch_v<0, 1, 1, 8, 2, 2, 2, 3>(g, splitk, st, lc);
    if (nsplit == 1 && f == 1006) return launch_v<0, 1, 1, 8, 2, 2, 2, 6>(g, splitk, st, lc);
    if (nsplit == 1 && f == 1008) return launch_v<0, 1, 1, 8, 2, 2, 2, 8>(g, splitk, st, lc);
#endif
    // 256 x 384 (8 waves of 4 x 3 MFMA tiles, the logits GEMM's shape; hi planes only): 44 % fewer fill bytes per flop than 256 x 128.
    // Its 4 output tiles need a split-K of ~64 to fill the chip, i.e. 64 slabs of [M, N] fp32 — 3.5 x the slab bytes of the default
    // form: right where the contraction is long enough for the slabs not to matter (K = catalog rows >= 2^20: the 10 M-item
    // configuration), wrong at the Globo catalog (85 MB of slabs against 25)
    const long w384 = (long)((g.M + 255) / 256) * ((g.N + 383) / 384) * splitk;
    if (nsplit == 1 && (f == 384 || (f == 0 && g.K >= (1 << 20) && w384 >= 192)))
      return launch_v<0, 1, 1, 2, 4, 4, 3>(g, splitk, st, lc);
    if (f == 512 || (f == 0 && g.M > 256 && w512 >= 192))
      return nsplit == 3 ? launch_v<0, 1, 3, 8, 2>(g, splitk, st, lc) : launch_v<0, 1, 1, 8, 2>(g, splitk, st, lc);
  }
  if constexpr (MB == 1) {
    // 256 x 192 (12 waves): an N extent such as 576 = 3 x 192 wastes no MFMA work on padding columns (256-wide tiles
    // would run a third, three-quarters-empty tile column)
    const int n192 = (g.N + 191) / 192 * 192, n256 = (g.N + 255) / 256 * 256;
    if constexpr (MA == 1) {
      // 192 x 192 (12 waves of 1 x 3 MFMA tiles): when 256-row tiles leave a ragged last round (dE at the Globo catalog:
      // 540 workgroups = 2.1 rounds of 256 CUs) three-quarter-size tiles in 2.8 rounds do 25 % less work per round
      const long w192 = (long)((g.M + 191) / 192) * (n192 / 192) * splitk;
      const long r256 = (w256 > 0 ? ((long)((g.M + 255) / 256) * (n192 / 192) * splitk + 255) / 256 : 0) * 4;   // rounds x size
      const long r192 = ((w192 + 255) / 256) * 3;
      if (f == 193 || (f == 0 && w256 >= 224 && n192 < n256 && r192 < r256))
        return nsplit == 3 ? launch_v<1, 1, 3, 6, 2, 1, 3>(g, splitk, st, lc) : launch_v<1, 1, 1, 6, 2, 1, 3>(g, splitk, st, lc);
    }
    if (f == 192 || (f == 0 && w256 >= 224 && n192 < n256))
      return nsplit == 3 ? launch_v<MA, MB, 3, 4, 3>(g, splitk, st, lc) : launch_v<MA, MB, 1, 4, 3>(g, splitk, st, lc);
  }
  if (f == 256 || (f == 0 && w256 >= 224))
    return nsplit == 3 ? launch_v<MA, MB, 3, 4, 4>(g, splitk, st, lc) : launch_v<MA, MB, 1, 4, 4>(g, splitk, st, lc);
  if (f == 128 || (f == 0 && w128 >= 192))
    return nsplit == 3 ? launch_v<MA, MB, 3, 4, 2>(g, splitk, st, lc) : launch_v<MA, MB, 1, 4, 2>(g, splitk, st, lc);
  return nsplit == 3 ? launch_v<MA, MB, 3, 2, 2>(g, splitk, st, lc) : launch_v<MA, MB, 1, 2, 2>(g, splitk, st, lc);
}

}  // namespace

extern "C" int tcar_gemm_bf16(int layout, int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner,
                              int64_t a_rows, const void* B_hi, const void* B_lo, int64_t b_inner, int64_t b_rows, float* C,
                              int64_t ldc, float* C2, int64_t ldc2, int csplit, int nsplit, int splitk, void* stream) {
  return tcar_gemm_bf16_perm(layout, M, N, K, A_hi, A_lo, a_inner, a_rows, B_hi, B_lo, b_inner, b_rows, C, ldc, C2, ldc2,
                             csplit, nullptr, 0, nsplit, splitk, stream);
}
extern "C" int tcar_gemm_bf16_tuned(const tcar_tuning_t* tune, int layout, int M, int N, int K, const void* A_hi, const void* A_lo,
                                    int64_t a_inner, int64_t a_rows, const void* B_hi, const void* B_lo, int64_t b_inner,
                                    int64_t b_rows, float* C, int64_t ldc, float* C2, int64_t ldc2, int csplit, int nsplit,
                                    int splitk, void* stream) {
  TcarOpt o;
  o.tune = tune;
  return tcar_gemm_bf16_perm_o(layout, M, N, K, A_hi, A_lo, a_inner, a_rows, B_hi, B_lo, b_inner, b_rows, C, ldc, C2, ldc2, csplit,
                               nullptr, 0, nsplit, splitk, stream, &o);
}

namespace {
struct CeOut { void* p_hi; int64_t p_inner; float* stats; const int32_t* label; float* lab_logit;
               const void* A2_hi; const void* A2_lo; const void* B2_hi; int64_t inner2, a2_rows, b2_rows; int K1; };
int gemm_bf16_impl(int layout, int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner, int64_t a_rows,
                   const void* B_hi, const void* B_lo, int64_t b_inner, int64_t b_rows, float* C, int64_t ldc, float* C2,
                   int64_t ldc2, int csplit, const int32_t* c2_perm, int c2_group, int nsplit, int splitk, const CeOut* ce,
                   void* stream, LaunchCall& lc);
}  // namespace

extern "C" int tcar_gemm_bf16_perm(int layout, int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner,
                                   int64_t a_rows, const void* B_hi, const void* B_lo, int64_t b_inner, int64_t b_rows,
                                   float* C, int64_t ldc, float* C2, int64_t ldc2, int csplit, const int32_t* c2_perm,
                                   int c2_group, int nsplit, int splitk, void* stream) {
  return tcar_gemm_bf16_perm_o(layout, M, N, K, A_hi, A_lo, a_inner, a_rows, B_hi, B_lo, b_inner, b_rows, C, ldc, C2, ldc2, csplit,
                               c2_perm, c2_group, nsplit, splitk, stream, nullptr);
}
int tcar_gemm_bf16_perm_o(int layout, int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner, int64_t a_rows,
                          const void* B_hi, const void* B_lo, int64_t b_inner, int64_t b_rows, float* C, int64_t ldc, float* C2,
                          int64_t ldc2, int csplit, const int32_t* c2_perm, int c2_group, int nsplit, int splitk, void* stream,
                          TcarOpt* o) {
  if (!C) return (M <= 0 || N <= 0 || K <= 0) ? TCAR_OK : TCAR_E_ARG;
  LaunchCall lc;
  lc.o = o;
  return gemm_bf16_impl(layout, M, N, K, A_hi, A_lo, a_inner, a_rows, B_hi, B_lo, b_inner, b_rows, C, ldc, C2, ldc2, csplit,
                        c2_perm, c2_group, nsplit, splitk, nullptr, stream, lc);
}

extern "C" int tcar_gemm_bf16_ce(int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner, int64_t a_rows,
                                 const void* B_hi, const void* B_lo, int64_t b_inner, int64_t b_rows, int K1, const void* A2_hi,
                                 const void* A2_lo, const void* B2_hi, int64_t inner2, void* p_hi, int64_t p_inner, int64_t p_rows,
                                 float* stats, int64_t stats_floats, const int32_t* label, float* lab_logit, int nsplit,
                                 int32_t* group_width, int32_t* ngroups, void* stream) {
  return tcar_gemm_bf16_ce_o(M, N, K, A_hi, A_lo, a_inner, a_rows, B_hi, B_lo, b_inner, b_rows, K1, A2_hi, A2_lo, B2_hi, inner2, p_hi,
                             p_inner, p_rows, stats, stats_floats, label, lab_logit, nsplit, group_width, ngroups, stream, nullptr);
}
// anchored form (TcarOpt::anchored): the caller has put the row references INTO the contraction (e.g. minus their partial sums in spare
// columns of A2 against columns of ones in B2), so the plane is exp(accumulator), the statistics (0, group sum), lab_logit relative
extern "C" int tcar_gemm_bf16_ce_anchor(int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner, int64_t a_rows,
                                        const void* B_hi, const void* B_lo, int64_t b_inner, int64_t b_rows, int K1, const void* A2_hi,
                                        const void* A2_lo, const void* B2_hi, int64_t inner2, void* p_hi, int64_t p_inner,
                                        int64_t p_rows, float* stats, int64_t stats_floats, const int32_t* label, float* lab_logit,
                                        int nsplit, int32_t* group_width, int32_t* ngroups, void* stream) {
  TcarOpt o;
  o.anchored = true;
  return tcar_gemm_bf16_ce_o(M, N, K, A_hi, A_lo, a_inner, a_rows, B_hi, B_lo, b_inner, b_rows, K1, A2_hi, A2_lo, B2_hi, inner2, p_hi,
                             p_inner, p_rows, stats, stats_floats, label, lab_logit, nsplit, group_width, ngroups, stream, &o);
}
// (flag-capable: consumers behind its flag read nothing this launch writes — the flag only times the arena zero)
int tcar_gemm_bf16_ce_o(int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner, int64_t a_rows, const void* B_hi,
                        const void* B_lo, int64_t b_inner, int64_t b_rows, int K1, const void* A2_hi, const void* A2_lo,
                        const void* B2_hi, int64_t inner2, void* p_hi, int64_t p_inner, int64_t p_rows, float* stats,
                        int64_t stats_floats, const int32_t* label, float* lab_logit, int nsplit, int32_t* group_width,
                        int32_t* ngroups, void* stream, TcarOpt* o) {
  if (M <= 0 || N <= 0 || K <= 0) return TCAR_OK;
  if (!p_hi || !stats || !label || !lab_logit || !group_width || !ngroups || (p_inner & 31) || p_inner < N || p_rows < M ||
      !tcar_aligned16(p_hi) || ((uintptr_t)stats & 7))
    return TCAR_E_ARG;
  // every tile of the logits layout has groups of at least 64 columns: [M, ceil(N / 64), 2] floats always suffice
  if (stats_floats < (int64_t)M * ((N + 63) / 64 + 8) * 2) return TCAR_E_ARG;
  CeOut ce = {p_hi, p_inner, stats, label, lab_logit, nullptr, nullptr, nullptr, 0, 0, 0, K};
  int k_first = K;
  if (B2_hi) {      // second K segment: [K1, K) from (A2 hi / lo, B2 hi), planes of inner dimension inner2 >= K - K1
    if (nsplit != 3 || !A2_hi || !A2_lo || K1 <= 0 || K1 >= K || (K1 & 31) || ((K - K1) & 31) || (inner2 & 31) || inner2 < K - K1 ||
        !tcar_aligned16(A2_hi) || !tcar_aligned16(A2_lo) || !tcar_aligned16(B2_hi))
      return TCAR_E_ARG;
    ce.A2_hi = A2_hi; ce.A2_lo = A2_lo; ce.B2_hi = B2_hi; ce.inner2 = inner2; ce.a2_rows = a_rows; ce.b2_rows = b_rows; ce.K1 = K1;
    k_first = K1;
  }
  float dummy;
  (void)k_first;
  LaunchCall lc;
  lc.o = o;
  const int rc = gemm_bf16_impl(1, M, N, K, A_hi, A_lo, a_inner, a_rows, B_hi, B_lo, b_inner, b_rows, &dummy, N, nullptr, 0, 0,
                                nullptr, 0, nsplit, 1, &ce, stream, lc);
  if (rc) return rc;
  *group_width = lc.ce_gw;
  *ngroups = lc.ce_ngroups;
  return TCAR_OK;
}

namespace {
int gemm_bf16_impl(int layout, int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner, int64_t a_rows,
                   const void* B_hi, const void* B_lo, int64_t b_inner, int64_t b_rows, float* C, int64_t ldc, float* C2,
                   int64_t ldc2, int csplit, const int32_t* c2_perm, int c2_group, int nsplit, int splitk, const CeOut* ce,
                   void* stream, LaunchCall& lc) {
  if (c2_perm && (!C2 || c2_group <= 0 || splitk > 1)) return TCAR_E_ARG;
  if (M <= 0 || N <= 0 || K <= 0) return TCAR_OK;
  if (layout < 0 || layout > 2 || !A_hi || !B_hi || !C || (nsplit != 1 && nsplit != 3)) return TCAR_E_ARG;
  if (nsplit == 3 && (!A_lo || !B_lo)) return TCAR_E_ARG;
  if ((a_inner & 31) || (b_inner & 31) || (K & 31) || !tcar_aligned16(A_hi) || !tcar_aligned16(B_hi)) return TCAR_E_ARG;
  // the planes must cover what the tiles touch: a k-contiguous operand has inner >= K and rows >= its M/N extent,
  // a transposed-read operand has inner >= its M/N extent and rows >= K
  const bool a_kc = (layout != 2), b_kc = (layout == 1);
  const int Kp = (ce && ce->B2_hi) ? ce->K1 : K;      // the first K segment is what these planes must cover
  if (a_kc ? (a_inner < Kp || a_rows < M) : (a_inner < M || a_rows < Kp)) return TCAR_E_ARG;
  if (b_kc ? (b_inner < Kp || b_rows < N) : (b_inner < N || b_rows < Kp)) return TCAR_E_ARG;
  BArgs g{};
  g.A[0] = (const __bf16*)A_hi; g.A[1] = (const __bf16*)A_lo; g.B[0] = (const __bf16*)B_hi; g.B[1] = (const __bf16*)B_lo;
  g.a_in32 = (int)(a_inner >> 5); g.b_in32 = (int)(b_inner >> 5);
  g.a_rb = (int)((a_rows + 127) >> 7); g.b_rb = (int)((b_rows + 127) >> 7);
  g.C = C; g.ldc = ldc;
  g.C2 = C2 ? C2 : C; g.ldc2 = C2 ? ldc2 : ldc; g.csplit = C2 ? csplit : N;
  g.perm = c2_perm; g.pgroup = c2_group;
  g.p_hi = nullptr; g.p_in32 = 0; g.stats = nullptr; g.ngroups = 0; g.label = nullptr; g.lab_logit = nullptr;
  g.A2[0] = g.A2[1] = nullptr; g.B2 = nullptr; g.a2_in32 = g.b2_in32 = g.a2_rb = g.b2_rb = 0; g.K1 = K;
  if (ce) {
    g.p_hi = (__bf16*)ce->p_hi; g.p_in32 = (int)(ce->p_inner >> 5); g.stats = ce->stats; g.label = ce->label;
    g.lab_logit = ce->lab_logit;
    if (ce->B2_hi) {
      g.A2[0] = (const __bf16*)ce->A2_hi; g.A2[1] = (const __bf16*)ce->A2_lo; g.B2 = (const __bf16*)ce->B2_hi;
      g.a2_in32 = g.b2_in32 = (int)(ce->inner2 >> 5);
      g.a2_rb = (int)((ce->a2_rows + 127) >> 7); g.b2_rb = (int)((ce->b2_rows + 127) >> 7);
      g.K1 = ce->K1;
    }
  }
  g.M = M; g.N = N; g.K = K;
  if (splitk < 1) splitk = 1;
  if (splitk > 1 && C2) return TCAR_E_ARG;
  int kchunk = (K + splitk - 1) / splitk;
  kchunk = ((kchunk + KB - 1) / KB) * KB;
  g.kchunk = kchunk;
  splitk = (K + kchunk - 1) / kchunk;
  g.mode = splitk > 1 ? 1 : 0;
  g.nsk = splitk;
  g.n_fastest = (layout != 1);        // layouts 0 / 2 stream dlogits tiles that several N tiles re-read
  hipStream_t st = (hipStream_t)stream;
  if (layout == 0) return launch_b<0, 1>(g, nsplit, splitk, st, lc);
  if (layout == 1) return launch_b<0, 0>(g, nsplit, splitk, st, lc);
  return launch_b<1, 1>(g, nsplit, splitk, st, lc);
}
}  // namespace

// ---- one-hot form of the two scoring GRADIENT GEMMs (hi planes only) ------------------------------------------------
// dX' = dlogits [E_item | E_content | OH]: the candidate-side time columns of the candidate matrix are five clipped table rows
// per item, selected by publish_time_MWDHM (139 distinct rows), so dlogits E_time = (dlogits OH) T_clip with the static 0/1 matrix
// OH [N, 160]: the GEMM contracts N against 2 ldh + 160 columns instead of 2 ldh + 5 ldt, and its last 160 columns are
// dP [B, 160] (tcar_reduce_dact_onehot expands them).  Layout 0 of tcar_gemm_bf16; B2 = OH plane [Npad rows, inner2 >= 160].
int tcar_gemm_bf16_dx_onehot_o(int M, int N1, int K, const void* A_hi, int64_t a_inner, int64_t a_rows, const void* B_hi,
                               int64_t b_inner, int64_t b_rows, const void* B2_hi, int64_t inner2, float* C, int64_t ldc, int splitk,
                               void* stream, TcarOpt* o) {
  if (M <= 0 || N1 <= 0 || K <= 0) return TCAR_OK;
  if (!A_hi || !B_hi || !B2_hi || !C || (a_inner & 31) || (b_inner & 31) || (inner2 & 31) || (K & 31) || (N1 & 127) || inner2 < 160 ||
      a_inner < K || a_rows < M || b_inner < N1 || b_rows < K || ldc < N1 + 160 || !tcar_aligned16(A_hi) || !tcar_aligned16(B_hi) ||
      !tcar_aligned16(B2_hi))
    return TCAR_E_ARG;
  BArgs g{};
  g.A[0] = (const __bf16*)A_hi; g.B[0] = (const __bf16*)B_hi; g.B2 = (const __bf16*)B2_hi;
  g.a_in32 = (int)(a_inner >> 5); g.b_in32 = (int)(b_inner >> 5); g.b2_in32 = (int)(inner2 >> 5);
  g.a_rb = (int)((a_rows + 127) >> 7); g.b_rb = (int)((b_rows + 127) >> 7);
  g.C = C; g.ldc = ldc; g.C2 = C; g.ldc2 = ldc; g.csplit = N1 + 160;
  g.M = M; g.N = N1 + 160; g.K = K; g.K1 = K; g.n_b2 = N1;
  if (splitk < 1) splitk = 1;
  int kchunk = (K + splitk - 1) / splitk;
  kchunk = ((kchunk + KB - 1) / KB) * KB;
  g.kchunk = kchunk;
  splitk = (K + kchunk - 1) / kchunk;
  g.mode = splitk > 1 ? 1 : 0;
  g.nsk = splitk;
  g.n_fastest = 1;
  if (o && o->start.flag) { g.start_flag = o->start.flag; g.start_epoch = o->start.epoch; o->started = true; }
  LaunchCall lc;
  lc.o = o;
  return launch_b<0, 1>(g, 1, splitk, (hipStream_t)stream, lc);
}
extern "C" int tcar_gemm_bf16_dx_onehot(int M, int N1, int K, const void* A_hi, int64_t a_inner, int64_t a_rows, const void* B_hi,
                                        int64_t b_inner, int64_t b_rows, const void* B2_hi, int64_t inner2, float* C, int64_t ldc,
                                        int splitk, void* stream) {
  return tcar_gemm_bf16_dx_onehot_o(M, N1, K, A_hi, a_inner, a_rows, B_hi, b_inner, b_rows, B2_hi, inner2, C, ldc, splitk, stream, nullptr);
}

extern "C" int tcar_gemm_bf16_dx_onehot_tuned(const tcar_tuning_t* tune, int M, int N1, int K, const void* A_hi, int64_t a_inner,
                                              int64_t a_rows, const void* B_hi, int64_t b_inner, int64_t b_rows, const void* B2_hi,
                                              int64_t inner2, float* C, int64_t ldc, int splitk, void* stream) {
  TcarOpt o;
  o.tune = tune;
  return tcar_gemm_bf16_dx_onehot_o(M, N1, K, A_hi, a_inner, a_rows, B_hi, b_inner, b_rows, B2_hi, inner2, C, ldc, splitk, stream, &o);
}

// dE' = dlogits^T [attout_item | attout_time] with the (q, z) epilogue: the item block [M, ldh] goes to C as in tcar_gemm_bf16
// (layout 2); the time block [M, 5 * 64] is NOT stored — per catalog row n and table k: qz[perm[k M + n]] = (||gy||^2, x . gy) with
// gy = the 64-column gradient block and x = tclip row the candidate looks up.  A plane = dlogits [K rows = sessions, inner >= M],
// B plane = packed attout [K rows, inner >= ldh + 320]; ldt must be 64.  tile: 0 = 192 x 192 (9 waves), 256 = 256 x 192 (12),
// 128 = 128 x 192 (6), 64 = 64 x 192 (3).
int tcar_gemm_bf16_de_qz_o(int M, int K, const void* A_hi, int64_t a_inner, int64_t a_rows, const void* B_hi, int64_t b_inner,
                           int64_t b_rows, int ldh, float* C, int64_t ldc, const int32_t* mwdhm, const int32_t* perm,
                           const float* tclip, float* qz, int tile, void* stream, TcarOpt* o) {
  if (M <= 0 || K <= 0) return TCAR_OK;
  const int N = ldh + 320;
  if (!A_hi || !B_hi || !C || !mwdhm || !perm || !tclip || !qz || ldh <= 0 || (ldh & 63) || (a_inner & 31) || (b_inner & 31) ||
      (K & 31) || a_inner < M || a_rows < K || b_inner < N || b_rows < K || ldc < ldh || !tcar_aligned16(A_hi) ||
      !tcar_aligned16(B_hi) || !tcar_aligned16(tclip) || ((uintptr_t)qz & 7))
    return TCAR_E_ARG;
  BArgs g{};
  g.A[0] = (const __bf16*)A_hi; g.B[0] = (const __bf16*)B_hi;
  g.a_in32 = (int)(a_inner >> 5); g.b_in32 = (int)(b_inner >> 5);
  g.a_rb = (int)((a_rows + 127) >> 7); g.b_rb = (int)((b_rows + 127) >> 7);
  g.C = C; g.ldc = ldc; g.C2 = C; g.ldc2 = ldc; g.csplit = ldh;
  g.perm = perm; g.pgroup = 64;
  g.M = M; g.N = N; g.K = K; g.K1 = K; g.kchunk = K; g.mode = 0; g.nsk = 1; g.n_fastest = 1;
  g.mwdhm = mwdhm; g.tclip = tclip; g.qz = (float2*)qz;
  hipStream_t st = (hipStream_t)stream;
  (void)o;
  if (tile == 256 || tile == 2562) {
    constexpr int TM = 256, TN = 192;
    constexpr size_t lds = 2 * (TM + TN) * 64;
    g.mt = (M + TM - 1) / TM; g.nt = (N + TN - 1) / TN;
    TCAR_SET_LDS_ONCE((gemm_bf16_kernel<1, 1, 1, 4, 3, 2, 2, 1, 2, 0>), lds);
    TCAR_LAUNCH((gemm_bf16_kernel<1, 1, 1, 4, 3, 2, 2, 1, 2, 0>), dim3(g.mt * g.nt), dim3(64 * 12), lds, st, g);
  } else if (tile == 64) {
    // (short catalog shards with many sessions — the sharded step at 8 ranks: 5,760 rows x 4,096 sessions — want MORE workgroups
    //  than 128-row tiles give: 3 waves, 32 KB of LDS, several per CU)
    constexpr int TM = 64, TN = 192;
    constexpr size_t lds = 2 * (TM + TN) * 64;
    g.mt = (M + TM - 1) / TM; g.nt = (N + TN - 1) / TN;
    TCAR_SET_LDS_ONCE((gemm_bf16_kernel<1, 1, 1, 1, 3, 2, 2, 1, 2, 0>), lds);
    TCAR_LAUNCH((gemm_bf16_kernel<1, 1, 1, 1, 3, 2, 2, 1, 2, 0>), dim3(g.mt * g.nt), dim3(64 * 3), lds, st, g);
  } else if (tile == 128) {
    constexpr int TM = 128, TN = 192;
    constexpr size_t lds = 2 * (TM + TN) * 64;
    g.mt = (M + TM - 1) / TM; g.nt = (N + TN - 1) / TN;
    TCAR_SET_LDS_ONCE((gemm_bf16_kernel<1, 1, 1, 2, 3, 2, 2, 1, 2, 0>), lds);
    TCAR_LAUNCH((gemm_bf16_kernel<1, 1, 1, 2, 3, 2, 2, 1, 2, 0>), dim3(g.mt * g.nt), dim3(64 * 6), lds, st, g);
  } else if (tile == 1922) {
    // (the double-buffered 192 x 192 form, kept for A/B against the ring: TCAR_BF16_TILE=1922)
    constexpr int TM = 192, TN = 192;
    constexpr size_t lds = 2 * (TM + TN) * 64;
    g.mt = (M + TM - 1) / TM; g.nt = (N + TN - 1) / TN;
    TCAR_SET_LDS_ONCE((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0>), lds);
    TCAR_LAUNCH((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0>), dim3(g.mt * g.nt), dim3(64 * 9), lds, st, g);
  } else if (tile == 1283) {
    constexpr int TM = 128, TN = 192;
    constexpr size_t lds = 3 * (TM + TN) * 64;
    g.mt = (M + TM - 1) / TM; g.nt = (N + TN - 1) / TN;
    TCAR_SET_LDS_ONCE((gemm_bf16_kernel<1, 1, 1, 2, 3, 2, 2, 1, 2, 0, 0, 3>), lds);
    TCAR_LAUNCH((gemm_bf16_kernel<1, 1, 1, 2, 3, 2, 2, 1, 2, 0, 0, 3>), dim3(g.mt * g.nt), dim3(64 * 6), lds, st, g);
  } else if (tile == 1923) {
    // 192 x 192, three-stage ring (72 KB: still two workgroups per CU)
    constexpr int TM = 192, TN = 192;
    constexpr size_t lds = 3 * (TM + TN) * 64;
    g.mt = (M + TM - 1) / TM; g.nt = (N + TN - 1) / TN;
    TCAR_SET_LDS_ONCE((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0, 0, 3>), lds);
    TCAR_LAUNCH((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0, 0, 3>), dim3(g.mt * g.nt), dim3(64 * 9), lds, st, g);
#ifdef TCAR_GEMM_DIAG
  } else if (tile >= 1002 && tile <= 1008) {
    constexpr int TM = 192, TN = 192;
    constexpr size_t lds = 2 * (TM + TN) * 64;
    g.mt = (M + TM - 1) / TM; g.nt = (N + TN - 1) / TN;
    if (tile == 1002) { TCAR_SET_LDS_ONCE((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0, 2>), lds);
      TCAR_LAUNCH((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0, 2>), dim3(g.mt * g.nt), dim3(64 * 9), lds, st, g); }
    else if (tile == 1003) { TCAR_SET_LDS_ONCE((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0, 3>), lds);
      TCAR_LAUNCH((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0, 3>), dim3(g.mt * g.nt), dim3(64 * 9), lds, st, g); }
    else if (tile == 1006) { TCAR_SET_LDS_ONCE((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0, 6>), lds);
      TCAR_LAUNCH((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0, 6>), dim3(g.mt * g.nt), dim3(64 * 9), lds, st, g); }
    else { TCAR_SET_LDS_ONCE((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0, 8>), lds);
      TCAR_LAUNCH((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0, 8>), dim3(g.mt * g.nt), dim3(64 * 9), lds, st, g); }
#endif
  } else {
    constexpr int TM = 192, TN = 192;
    constexpr size_t lds = 2 * (TM + TN) * 64;
    g.mt = (M + TM - 1) / TM; g.nt = (N + TN - 1) / TN;
    TCAR_SET_LDS_ONCE((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0>), lds);
    TCAR_LAUNCH((gemm_bf16_kernel<1, 1, 1, 3, 3, 2, 2, 1, 2, 0>), dim3(g.mt * g.nt), dim3(64 * 9), lds, st, g);
  }
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
extern "C" int tcar_gemm_bf16_de_qz(int M, int K, const void* A_hi, int64_t a_inner, int64_t a_rows, const void* B_hi, int64_t b_inner,
                                    int64_t b_rows, int ldh, float* C, int64_t ldc, const int32_t* mwdhm, const int32_t* perm,
                                    const float* tclip, float* qz, int tile, void* stream) {
  return tcar_gemm_bf16_de_qz_o(M, K, A_hi, a_inner, a_rows, B_hi, b_inner, b_rows, ldh, C, ldc, mwdhm, perm, tclip, qz, tile, stream,
                                nullptr);
}

extern "C" int tcar_gemm_bf16_variant(int layout, int M, int N, int K, int nsplit, int splitk, char* buf, int buflen) {
  if (!buf || buflen < 8 || layout < 0 || layout > 2 || M <= 0 || N <= 0 || K <= 0 || (K & 31)) return TCAR_E_ARG;
  LaunchCall lc;
  lc.variant_out = buf;
  lc.variant_len = buflen;
  buf[0] = 0;
  // plane geometry as tcar_gemm_bf16 requires it; the pointers are never dereferenced on the dry path
  const bool a_kc = (layout != 2), b_kc = (layout == 1);
  const int64_t r32 = 31;
  const int64_t a_inner = ((a_kc ? K : M) + r32) & ~r32, a_rows = a_kc ? M : K;
  const int64_t b_inner = ((b_kc ? K : N) + r32) & ~r32, b_rows = b_kc ? N : K;
  alignas(16) const char dummy[16] = {0};
  float c = 0.f;
  return gemm_bf16_impl(layout, M, N, K, dummy, dummy, a_inner, a_rows, dummy, dummy, b_inner, b_rows, &c, N, nullptr, 0, 0, nullptr,
                        0, nsplit, splitk, nullptr, nullptr, lc);
}

extern "C" int tcar_split_bf16(const float* x, int64_t ld, int rows, int cols, void* hi, void* lo, int64_t inner,
                               void* packed_hi, void* packed_lo, int64_t packed_inner, int c0, int c1, void* stream) {
  if (rows <= 0 || cols <= 0) return TCAR_OK;
  if (!x || !hi || (inner & 31) || inner < cols || (c0 & 3) || (c1 & 3) || (packed_hi && (packed_inner & 31))) return TCAR_E_ARG;
  long total = (((long)rows + 127) & ~127L) * (inner >> 2);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  TCAR_LAUNCH(split_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long)ld, rows, cols, (__bf16*)hi,
              (__bf16*)lo, (int)inner, (__bf16*)packed_hi, (__bf16*)packed_lo, (int)packed_inner, c0, c1);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
