// Embedding lookups with norm clipping (forward + backward) for gfx950.
//
// Reference ops: tf.nn.embedding_lookup(table, ids, max_norm=1) at modules.py:36 and
// model_combine.py:68,87-91,95-96 (gather, then clip every gathered row to L2 norm <= 1, gradient through the
// clip), plus the concats of model_combine.py:65,84,94,111.
//
// HBM-bound kernels.  One 64-lane wave owns one (session, position) row: an H-float row (<= 256 floats) is one
// 16-byte load per lane = one fully coalesced 1-KiB wave instruction; the five 64-float time rows and the
// dwell row share a wave instruction in 16-lane groups.  Row norms are wave / group shuffle reductions; nothing
// goes through LDS in the forward pass.  The backward pass re-gathers the raw rows (needed by the clip
// Jacobian), adds item-row gradients straight into the dense [N, ldh] gradient with 256-byte-contiguous float
// atomics (the shape the memory-side atomic unit runs at full rate), and privatises the tiny, heavily
// contended tables (position, 5 time tables, dwell: <= 190 rows) in LDS, flushing each workgroup's non-zero
// rows once at the end.
#include "tcar_common.h"
#include "tcar_bf16_layout.h"

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_e;

namespace {

struct EmbArgs {
  tcar_dims_t d;
  tcar_tables_t tab;
  tcar_batch_t bt;
  TcarSignal sig;      // completion flag (latency form of the forward gather only): click_t is then stored write-through
  float* x_icp; float* x_pt; float* x_act; float* click_t;
  const float* dx_icp; const float* dx_pt; const float* dx_act; const float* dclick;
  tcar_grads_t g;
  // backward only: with n_gather > 0 the workgroups [n_gather, gridDim.x) compute block partials of sum sq_g^2 (the dense
  // item norm, tcar_sqnorm_det's job) beside the row gradients instead of in a launch of their own
  const float* sq_g; long sq_len; float* sq_part; int n_gather;
};

__device__ __forceinline__ int time_vocab(int k) {
  return k == 0 ? 13 : k == 1 ? 32 : k == 2 ? 8 : k == 3 ? 25 : k == 4 ? 61 : TCAR_DUR_VOCAB;
}
// row offset of table k inside the contiguous [month|day|week|hour|minute|dur] block
__device__ __forceinline__ int time_rowoff(int k) {
  return k == 0 ? 0 : k == 1 ? 13 : k == 2 ? 45 : k == 3 ? 53 : k == 4 ? 78 : 139;
}
constexpr int SMALL_ROWS = 150;  // 13+32+8+25+61+11

// select one of five kernel-argument pointers without a runtime-indexed (scratch) copy of the argument struct
template <typename P>
__device__ __forceinline__ P pick5(P const (&arr)[5], int k) {
  return k == 0 ? arr[0] : k == 1 ? arr[1] : k == 2 ? arr[2] : k == 3 ? arr[3] : arr[4];
}

// ------------------------------------------------------------------------------------------------ forward
template <int NCH>  // NCH = ceil(ldh / 256): 16-byte chunks per lane for an H-row
__global__ __launch_bounds__(256) void gather_clip_fwd_kernel(const EmbArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const int B = a.bt.B, T = a.bt.T, BT = B * T;
  const int ldh = a.d.ldh, ldt = a.d.ldt;
  const int ic = 2 * ldh, pt = 5 * ldt, ek = ic + pt, ct = 2 * ldt;
  const int sub = ldt >> 2;          // lanes per time row (16 for ldt = 64)
  const int gpw = 64 / sub;          // time rows per wave instruction
  const int grp = lane / sub, lin = lane - grp * sub;

  // Two session rows per wave and trip: all row loads of both rows (item, content, position; then the six small
  // rows) are issued before the first reduction, doubling the bytes a wave keeps in flight (a row is ~3.5 KB).
  constexpr int R = 2;
  for (int row0 = wave_g; row0 < BT; row0 += R * nwaves) {
    float4 xi[R][NCH], xc[R][NCH], xp[R][NCH];
    bool live[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int row = row0 + u * nwaves;
      live[u] = row < BT;
      const int rw = live[u] ? row : row0;
      const int t = rw % T;
      const int n = clampi(a.bt.seq[rw], 1, a.d.n_items) - 1;
      const float* e = a.tab.E + (long)n * ek;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        const bool ok = col < ldh;
        xi[u][c] = ok ? ld4(e + col) : zero4();
        xc[u][c] = ok ? ld4(e + ldh + col) : zero4();
        xp[u][c] = ok ? ld4(a.tab.pos + (long)t * ldh + col) : zero4();
      }
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int row = row0 + u * nwaves;
      float si = 0.f, sc = 0.f, sp = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        si += dot4(xi[u][c], xi[u][c]); sc += dot4(xc[u][c], xc[u][c]); sp += dot4(xp[u][c], xp[u][c]);
      }
      si = clip_scale(wave_sum(si)); sc = clip_scale(wave_sum(sc)); sp = clip_scale(wave_sum(sp));
      if (live[u]) {
        float* o = a.x_icp + (long)row * ic;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int col = c * 256 + lane * 4;
          if (col < ldh) {
            st4(o + col, fma4(xi[u][c], si, scale4(xp[u][c], sp)));
            st4(o + ldh + col, scale4(xc[u][c], sc));
          }
        }
      }
    }
    // five publish-time rows + the dwell row of both session rows, `gpw` small rows per pass
    for (int k0 = 0; k0 < 6; k0 += gpw) {
      const int k = k0 + grp;
      const bool valid = k < 6;
      const int kk = valid ? k : 0;
      const float* tp = (kk < 5) ? pick5(a.tab.time, kk) : a.tab.dur;
      float4 x[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int rw = live[u] ? row0 + u * nwaves : row0;
        int id = (kk < 5) ? pick5(a.bt.pub, kk)[rw] : a.bt.gap[rw];
        const bool oob = (kk == 5) && (id >= TCAR_DUR_VOCAB || id < 0);   // dwell bucket 11 -> zero row (S7)
        id = clampi(id, 0, time_vocab(kk) - 1);
        x[u] = (valid && !oob) ? ld4(tp + (long)id * ldt + lin * 4) : zero4();
      }
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int row = row0 + u * nwaves;
        const float sx = clip_scale(group_sum(dot4(x[u], x[u]), sub));
        if (valid && live[u]) {
          float* dst = (kk < 5) ? a.x_pt + (long)row * pt + kk * ldt : a.x_act + (long)row * ldt;
          st4(dst + lin * 4, scale4(x[u], sx));
        }
      }
    }
  }
  // click-time query rows: week table by cw, hour table by ch (model_combine.py:94-97)
  for (int b = wave_g; b < B; b += nwaves) {
    for (int k0 = 0; k0 < 2; k0 += gpw) {
      const int j = k0 + grp;
      const bool valid = j < 2;
      const int jj = valid ? j : 0;
      const int kk = jj == 0 ? 2 : 3;
      const int id = clampi(jj == 0 ? a.bt.cw[b] : a.bt.ch[b], 0, time_vocab(kk) - 1);
      float4 x = valid ? ld4(pick5(a.tab.time, kk) + (long)id * ldt + lin * 4) : zero4();
      const float sx = clip_scale(group_sum(dot4(x, x), sub));
      if (valid) {
        // with a completion flag the click rows are what the flag's consumer reads (the click-query MLP on a side stream)
        if (a.sig.cnt) st4_sc1(a.click_t + (long)b * ct + jj * ldt + lin * 4, scale4(x, sx));
        else st4(a.click_t + (long)b * ct + jj * ldt + lin * 4, scale4(x, sx));
      }
    }
  }
  tcar_signal_done(a.sig);
}

// ------------------------------------------------------------------------------- forward, throughput form
// Large launches (>= 16 k session rows: evaluation sweeps, the 100 M-session stress configuration, tools/gather_bench.py).
// The latency form above keeps ~2 rows per wave in flight and re-reads the seven small rows of every session row from L2
// (2.5 KB of L1/L2 traffic beside 2 KB of HBM reads) with a shuffle reduction each.  Here:
//   * the <= 190 small rows (position, 5 time tables, dwell) are clipped ONCE per workgroup into LDS (78 KB for H = 250,
//     Ht = 64): a session row then costs two HBM row reads, two wave reductions and seven LDS reads;
//   * 1024-thread workgroups (16 waves), two per CU, every wave owns R = 2 CONSECUTIVE session rows per trip (round 5: 2.4 % faster
//     than 4 rows in five interleaved rounds; 8 rows and 1 row slower): 4 KB of HBM reads in flight per wave, 128 KB per CU;
//   * the ids of the next trip are fetched (scalar loads: the row index is wave uniform) before the current one is reduced;
//   * outputs leave with non-temporal stores (3.5 KB written per 2 KB read: they would only evict the table rows).
typedef float v4f_e __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  v4f_e x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<v4f_e*>(p));
}

// R: consecutive session rows per wave and trip; PLAIN: plain stores instead of non-temporal ones (round 5 measured R = 1 / 4 / 8 and
// plain stores slower than the defaults: profiles/r05_ab_experiments.txt)
template <int NCH, int R = 2, bool PLAIN = false>
__global__ __launch_bounds__(1024) void gather_clip_fwd_big_kernel(const EmbArgs a) {
  auto st_out = [](float* p, float4 v) { if (PLAIN) st4(p, v); else st4_nt(p, v); };
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int B = a.bt.B, T = a.bt.T, BT = B * T;
  const int ldh = a.d.ldh, ldt = a.d.ldt;
  const int ic = 2 * ldh, pt = 5 * ldt, ek = ic + pt, ct = 2 * ldt;
  const int sub = ldt >> 2, gpw = 64 / sub;
  const int grp = lane / sub, lin = lane - grp * sub;
  float* lpos = lds;                                  // [40][ldh]   clipped position rows
  float* lsm = lds + TCAR_POS_VOCAB * ldh;            // [150][ldt]  clipped month|day|week|hour|minute|dwell rows
  for (int r = wave; r < TCAR_POS_VOCAB; r += 16) {
    float4 x[NCH];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      x[c] = col < ldh ? ld4(a.tab.pos + (long)r * ldh + col) : zero4();
      ss += dot4(x[c], x[c]);
    }
    ss = clip_scale(wave_sum(ss));
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < ldh) st4(lpos + r * ldh + col, scale4(x[c], ss));
    }
  }
  for (int r0 = wave * gpw; r0 < SMALL_ROWS; r0 += 16 * gpw) {
    const int r = r0 + grp;
    const bool ok = r < SMALL_ROWS;
    const int rr = ok ? r : 0;
    const int k = rr < 13 ? 0 : rr < 45 ? 1 : rr < 53 ? 2 : rr < 78 ? 3 : rr < 139 ? 4 : 5;
    const float* tp = (k < 5) ? pick5(a.tab.time, k) : a.tab.dur;
    const float4 x = ok ? ld4(tp + (long)(rr - time_rowoff(k)) * ldt + lin * 4) : zero4();
    const float sx = clip_scale(group_sum(dot4(x, x), sub));
    if (ok) st4(lsm + rr * ldt + lin * 4, scale4(x, sx));
  }
  __syncthreads();

  const long wave_g = (long)blockIdx.x * 16 + wave;
  const long stride = (long)gridDim.x * 16 * R;
  int ids[R], ids_n[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const long rw = wave_g * R + u;
    ids[u] = a.bt.seq[rw < BT ? rw : 0];
  }
  for (long row0 = wave_g * R; row0 < BT; row0 += stride) {
#pragma unroll
    for (int u = 0; u < R; ++u) {                     // next trip's item ids (consumed one trip later)
      const long rn = row0 + stride + u;
      ids_n[u] = a.bt.seq[rn < BT ? rn : 0];
    }
    float4 xi[R][NCH], xc[R][NCH];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int n = clampi(ids[u], 1, a.d.n_items) - 1;
      const float* e = a.tab.E + (long)n * ek;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        const bool ok = col < ldh;
        xi[u][c] = ok ? ld4(e + col) : zero4();
        xc[u][c] = ok ? ld4(e + ldh + col) : zero4();
      }
    }
    // the small rows of the R session rows: `gpw` (row, table) pairs per pass out of 6 * R, straight from LDS
    for (int q0 = 0; q0 < 6 * R; q0 += gpw) {
      const int q = q0 + grp;
      const int u = q / 6, k = q - u * 6;
      const long row = row0 + u;
      const bool valid = q < 6 * R && row < BT;
      const long rw = valid ? row : row0;
      const int kk = valid ? k : 0;
      int id = (kk < 5) ? pick5(a.bt.pub, kk)[rw] : a.bt.gap[rw];
      const bool oob = (kk == 5) && (id >= TCAR_DUR_VOCAB || id < 0);   // dwell bucket 11 -> zero row (S7)
      id = clampi(id, 0, time_vocab(kk) - 1);
      const float4 x = oob ? zero4() : ld4(lsm + (time_rowoff(kk) + id) * ldt + lin * 4);
      if (valid) {
        float* dst = (kk < 5) ? a.x_pt + row * pt + kk * ldt : a.x_act + row * ldt;
        st_out(dst + lin * 4, x);
      }
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const long row = row0 + u;
      float si = 0.f, sc = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) { si += dot4(xi[u][c], xi[u][c]); sc += dot4(xc[u][c], xc[u][c]); }
      si = clip_scale(wave_sum(si)); sc = clip_scale(wave_sum(sc));
      if (row < BT) {
        const int t = (int)(row % T);
        float* o = a.x_icp + row * ic;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int col = c * 256 + lane * 4;
          if (col < ldh) {
            st_out(o + col, fma4(xi[u][c], si, ld4(lpos + t * ldh + col)));
            st_out(o + ldh + col, scale4(xc[u][c], sc));
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < R; ++u) ids[u] = ids_n[u];
  }
  // click-time query rows: week table by cw, hour table by ch (model_combine.py:94-97)
  for (long b0 = wave_g * gpw; b0 < 2L * B; b0 += (long)gridDim.x * 16 * gpw) {
    const long q = b0 + grp;
    const bool valid = q < 2L * B;
    const long b = valid ? (q >> 1) : 0;
    const int jj = (int)(q & 1), kk = jj == 0 ? 2 : 3;
    const int id = clampi(jj == 0 ? a.bt.cw[b] : a.bt.ch[b], 0, time_vocab(kk) - 1);
    const float4 x = ld4(lsm + (time_rowoff(kk) + id) * ldt + lin * 4);
    if (valid) st4(a.click_t + b * ct + jj * ldt + lin * 4, x);
  }
}

// ----------------------------------------------------------------------------------------------- backward
template <int NCH, int LDT>   // LDT = ldt (64 / 128 / 256): the 16-lane-group geometry of the small tables is compile time
__global__ __launch_bounds__(256) void gather_clip_bwd_kernel(const EmbArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int ngb = a.n_gather > 0 ? a.n_gather : (int)gridDim.x;
  if ((int)blockIdx.x >= ngb) {          // dense-norm role: partial j of nsq, fixed stripes (deterministic, segsum.hip folds them)
    const int j = blockIdx.x - ngb, nsq = gridDim.x - ngb;
    float s = 0.f;
    for (long base = (long)j * 8192; base < a.sq_len; base += (long)nsq * 8192) {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long e = base + (i * 256 + tid) * 4;
        v[i] = (e < a.sq_len) ? ld4(a.sq_g + e) : zero4();
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += dot4(v[i], v[i]);
    }
    s = wave_sum(s);
    float* sh = lds;
    if (lane == 0) sh[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) a.sq_part[j] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    return;
  }
  const int wave_g = blockIdx.x * 4 + (tid >> 6);
  const int nwaves = ngb * 4;
  const int B = a.bt.B, T = a.bt.T, BT = B * T;
  const int ldh = a.d.ldh;
  constexpr int ldt = LDT;
  const int ic = 2 * ldh, pt = 5 * ldt, ek = ic + pt, ct = 2 * ldt;
  constexpr int sub = LDT >> 2, gpw = 64 / sub, NIT = (6 + gpw - 1) / gpw;
  const int grp = lane / sub, lin = lane - grp * sub;
  float* pos_acc = lds;                       // [T, ldh]
  float* small_acc = lds + T * ldh;           // [150, ldt]
  float* sq_acc = small_acc + SMALL_ROWS * ldt;  // [8]
  const int lds_floats = T * ldh + SMALL_ROWS * ldt + 8;
  for (int i = tid; i < lds_floats; i += 256) lds[i] = 0.f;
  __syncthreads();

  float sq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sq[i] = 0.f;

  const bool item_only = a.g.skip_small != 0;     // the order-fixed form (tcar_small_tables_bwd_det) owns every other table
  for (int row = wave_g; row < (item_only ? BT : BT + B); row += nwaves) {
    if (row < BT) {
      const int t = row % T;
      const int n = clampi(a.bt.seq[row], 1, a.d.n_items) - 1;
      const float* e = a.tab.E + (long)n * ek;
      const float* gy = a.dx_icp + (long)row * ic;
      // every load of the row is issued before its first use: the ids of the six small tables first, then all rows
      // (a row is otherwise a chain of 5-6 dependent global round trips, ~9 us per row and wave)
      int kid[NIT];
      bool kval[NIT], kact[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int k = it * gpw + grp;
        kval[it] = k < 6;
        const int kk = kval[it] ? k : 0;
        int id = (kk < 5) ? pick5(a.bt.pub, kk)[row] : a.bt.gap[row];
        const bool oob = (kk == 5) && (id >= TCAR_DUR_VOCAB || id < 0);   // dwell bucket 11 (S7)
        kid[it] = clampi(id, 0, time_vocab(kk) - 1);
        kact[it] = kval[it] && !oob;
      }
      float4 xi[NCH], xp[NCH], gi[NCH], kx[NIT], kgy[NIT];
      float ssi = 0.f, di = 0.f, ssp = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        const bool ok = col < ldh;
        xi[c] = ok ? ld4(e + col) : zero4();
        xp[c] = ok ? ld4(a.tab.pos + (long)t * ldh + col) : zero4();
        gi[c] = ok ? ld4(gy + col) : zero4();
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int kk = kval[it] ? it * gpw + grp : 0;
        const float* tp = (kk < 5) ? pick5(a.tab.time, kk) : a.tab.dur;
        const float* gp = (kk < 5) ? a.dx_pt + (long)row * pt + kk * ldt : a.dx_act + (long)row * ldt;
        kx[it] = (kact[it] && !item_only) ? ld4(tp + (long)kid[it] * ldt + lin * 4) : zero4();
        kgy[it] = (kval[it] && !item_only) ? ld4(gp + lin * 4) : zero4();
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        ssi += dot4(xi[c], xi[c]); di += dot4(xi[c], gi[c]);
        ssp += dot4(xp[c], xp[c]); dp += dot4(xp[c], gi[c]);
      }
      ssi = wave_sum(ssi); di = wave_sum(di); ssp = wave_sum(ssp); dp = wave_sum(dp);
      float ai, bi, ap, bp;
      clip_bwd_coef(ssi, di, ai, bi);
      clip_bwd_coef(ssp, dp, ap, bp);
      float rown = 0.f;                       // this source row's squared norm (norms_out: summed later in a fixed order)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < ldh) {
          float4 gx = fma4(xi[c], -bi, scale4(gi[c], ai));
          if (a.g.norms_out) rown += dot4(gx, gx);
          else sq[0] += dot4(gx, gx);
          if (a.g.rows_out) st4(a.g.rows_out + (long)row * (a.g.rows_ld ? a.g.rows_ld : (long)ldh) + col, gx);
          else atomic_add4(a.g.g_item + (long)n * ldh + col, gx);
          if (!item_only) {
            float4 gp = fma4(xp[c], -bp, scale4(gi[c], ap));
            sq[1] += dot4(gp, gp);
            atomic_add4(pos_acc + t * ldh + col, gp);
          }
        }
      }
      if (a.g.norms_out) {
        rown = wave_sum(rown);
        if (lane == 0) a.g.norms_out[row] = rown;
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        if (item_only) break;
        const int kk = kval[it] ? it * gpw + grp : 0;
        const float4 x = kx[it], gyv = kgy[it];
        const float ss = group_sum(dot4(x, x), sub), dd = group_sum(dot4(x, gyv), sub);
        float ca, cb;
        clip_bwd_coef(ss, dd, ca, cb);
        if (kval[it]) {
          // an out-of-range dwell id gathers a zero row (S7): its gradient row still belongs to the
          // IndexedSlices values (it counts in the clip norm) but is dropped by the scatter
          float4 gx = fma4(x, -cb, scale4(gyv, ca));
          const float q = dot4(gx, gx);
#pragma unroll
          for (int s2 = 0; s2 < 6; ++s2) sq[2 + s2] += (s2 == kk) ? q : 0.f;
          if (kact[it]) atomic_add4(small_acc + (time_rowoff(kk) + kid[it]) * ldt + lin * 4, gx);
        }
      }
    } else {
      const int b = row - BT;
      for (int k0 = 0; k0 < 2; k0 += gpw) {
        const int j = k0 + grp;
        const bool valid = j < 2;
        const int jj = valid ? j : 0;
        const int kk = jj == 0 ? 2 : 3;
        const int id = clampi(jj == 0 ? a.bt.cw[b] : a.bt.ch[b], 0, time_vocab(kk) - 1);
        float4 x = valid ? ld4(pick5(a.tab.time, kk) + (long)id * ldt + lin * 4) : zero4();
        float4 gyv = valid ? ld4(a.dclick + (long)b * ct + jj * ldt + lin * 4) : zero4();
        const float ss = group_sum(dot4(x, x), sub), dd = group_sum(dot4(x, gyv), sub);
        float ca, cb;
        clip_bwd_coef(ss, dd, ca, cb);
        if (valid) {
          float4 gx = fma4(x, -cb, scale4(gyv, ca));
          const float q = dot4(gx, gx);
          sq[2 + 2] += (kk == 2) ? q : 0.f;
          sq[2 + 3] += (kk == 3) ? q : 0.f;
          atomic_add4(small_acc + (time_rowoff(kk) + id) * ldt + lin * 4, gx);
        }
      }
    }
  }
  // squared-norm pieces: wave reduce -> LDS -> one global atomic per workgroup and slot
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float s = wave_sum(sq[i]);
    if (lane == 0 && s != 0.f) atomicAdd(sq_acc + i, s);
  }
  __syncthreads();
  for (int i = tid; i < T * ldh; i += 256) {
    const float v = pos_acc[i];
    if (v != 0.f) atomicAdd(a.g.g_pos + i, v);
  }
  for (int i = tid; i < SMALL_ROWS * ldt; i += 256) {
    const float v = small_acc[i];
    if (v != 0.f) atomicAdd(a.g.g_time[0] + i, v);   // g_time[0..4], g_dur are one contiguous block
  }
  if (tid < 8) {
    const float v = sq_acc[tid];
    if (v != 0.f) {
      const int slot = tid == 0 ? a.g.slot_item : tid == 1 ? a.g.slot_pos : tid == 7 ? a.g.slot_dur : pick5(a.g.slot_time, tid - 2);
      atomicAdd(a.g.sqn + slot, v);
    }
  }
}

// ------------------------------------------------------------------------------- candidate-side time vectors
struct CandArgs {
  tcar_dims_t d;
  const float* tab[5];
  const int32_t* mwdhm;
  float* E;
  const float* d_et;
  tcar_grads_t g;
  __bf16* eh; __bf16* el;     // optional bf16 hi / lo planes of E (same [Npad, ek] layout)
};

// E[n, ic + k*ldt ...] = clip(table_k[mwdhm[n,k]])   (model_combine.py:86-92)
__global__ __launch_bounds__(256) void cand_time_fwd_kernel(const CandArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // clipped copies of the 139 time rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldt = a.d.ldt, ldh = a.d.ldh, ic = 2 * ldh, ek = ic + 5 * ldt;
  const int sub = ldt >> 2, gpw = 64 / sub;
  const int grp = lane / sub, lin = lane - grp * sub;
  // stage: each 16-lane group clips one table row into LDS
  for (int r0 = wave * gpw; r0 < 139; r0 += 4 * gpw) {
    const int r = r0 + grp;
    const bool valid = r < 139;
    const int rr = valid ? r : 0;
    const int k = rr < 13 ? 0 : rr < 45 ? 1 : rr < 53 ? 2 : rr < 78 ? 3 : 4;
    const int id = rr - time_rowoff(k);
    float4 x = valid ? ld4(pick5(a.tab, k) + (long)id * ldt + lin * 4) : zero4();
    const float s = clip_scale(group_sum(dot4(x, x), sub));
    if (valid) st4(lds + rr * ldt + lin * 4, scale4(x, s));
  }
  __syncthreads();
  // One 128-row block of E per workgroup pass; thread = (row, 16-column half of a 32-wide k-block).  In the KB32 planes
  // a wave then writes 32 rows x 64 B = 2 KB contiguous per store pair; all index arithmetic is 32-bit.
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_e;
  const int kpt = ldt >> 5, in32 = ek >> 5;
  const int r = tid >> 1, half = tid & 1;
  const int nrb = (a.d.n_items + 127) >> 7;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int n = rb * 128 + r;
    if (n >= a.d.n_items) continue;                   // padding rows of E / the planes stay zero
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int id = clampi(a.mwdhm[n * 5 + k], 0, time_vocab(k) - 1);
      const float* src = lds + (time_rowoff(k) + id) * ldt + half * 16;
      for (int q = 0; q < kpt; ++q) {
        const int col = ic + k * ldt + q * 32 + half * 16;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 t = ld4(src + q * 32 + j * 4);
          v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
          if (a.E) st4(a.E + (long)n * ek + col + j * 4, t);
        }
        if (a.eh) {
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) {
            bf16x8_e h, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              h[j] = (__bf16)v[pc * 8 + j];
              lo[j] = (__bf16)(v[pc * 8 + j] - (float)h[j]);
            }
            const long o = kb32_off(n, col + pc * 8, in32);
            *reinterpret_cast<bf16x8_e*>(a.eh + o) = h;
            *reinterpret_cast<bf16x8_e*>(a.el + o) = lo;
          }
        }
      }
    }
  }
}

// ---- candidate-side time scores through a one-hot contraction (round 3) ---------------------------------------------
// logits[b, n] = attout_ic[b] . E_ic[n] + sum_k attout_tk[b] . clip(table_k[mwdhm[n, k]])      (model_combine.py:86-92,135-138)
// The second term only takes 139 distinct values per session: P[b, r] = attout_t,k(r)[b] . clip(table row r), and
// sum_k P[b, r_k(n)] = (P OH^T)[b, n] with the STATIC one-hot matrix OH[n, r] = [r in {r_0(n) .. r_4(n)}].  The logits GEMM then
// contracts 512 + 160 columns instead of 512 + 320, and the one-hot block (exact in bf16: ONE plane, two MFMAs per product)
// costs half the B-operand fill of a two-plane block.
__global__ __launch_bounds__(256) void time_onehot_kernel(int n_items, const int32_t* __restrict__ mwdhm, __bf16* __restrict__ oh,
                                                          int in32) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= n_items) return;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int id = clampi(mwdhm[n * 5 + k], 0, time_vocab(k) - 1);
    oh[kb32_off(n, time_rowoff(k) + id, in32)] = (__bf16)1.0f;
  }
  // columns 139 .. 146: ONES — the anchor columns of the anchored softmax form.  The time-score planes P hold minus the partial sums of
  // a session's anchor there (attout_finish_kernel) and zeros otherwise (every other writer of P zeroes columns >= 139), so the logits
  // GEMM's one-hot K segment — 160 columns of which 139 carry time rows — subtracts the anchor inside the contraction, for free
#pragma unroll
  for (int j = 0; j < TCAR_ANCHOR_COLS; ++j) oh[kb32_off(n, 139 + j, in32)] = (__bf16)1.0f;
}
// anchor of the anchored softmax form where the finishing launch of the output transforms is not at hand (catalog-sharded step: the
// shard scores the GATHERED attout rows of every rank): one wave per session, a = attout[b, 0 : 2 ldh) . E[label[b], 0 : 2 ldh) in a
// fixed order — the same bits on every rank, whose copies of attout and of the fp32 table are identical —, P[b, 139] = -a (hi / lo);
// the other anchor columns stay zero (tcar_time_scores_clip has just zeroed columns >= 139).  Padding sessions (label < 0): zero.
__global__ __launch_bounds__(256) void anchor_scores_kernel(int ic, int B, const float* __restrict__ attout, long ld_att,
                                                            const int32_t* __restrict__ label, const float* __restrict__ E, long ldE,
                                                            long n_rows, __bf16* __restrict__ ph, __bf16* __restrict__ pl, int in32) {
  const int lane = threadIdx.x & 63;
  const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lab = label[b];
  float a = 0.f;
  if (lab >= 0) {
    const float* x = attout + b * ld_att;
    const float* e = E + (lab < n_rows ? (long)lab : n_rows - 1) * ldE;
    for (int c = lane * 4; c < ic; c += 256) a += dot4(ld4(x + c), ld4(e + c));
    a = -wave_sum(a);
  }
  if (lane == 0) {
    const __bf16 h = (__bf16)a;
    const long o = kb32_off(b, 139, in32);
    ph[o] = h;
    pl[o] = (__bf16)(a - (float)h);
  }
}
struct ScoreArgs {
  tcar_dims_t d;
  const float* tab[5];
  const float* attout; long ld_att;
  int B, in32;
  __bf16* ph; __bf16* pl;
  float* tclip;        // optional [160 ldt | 160 | 160]: clipped table rows, their clip scales, 1.0 where the clip is active
};
// One workgroup = 16 session rows x ONE table (grid.y = 5): its raw rows (<= 61), their clip scales and the 16 x ldt block of
// attout are staged with independent loads (one memory round trip), the dots run out of LDS — thread (b, rg) takes rows rg, rg +
// 16, ... so that a wave reads 16 distinct attout rows and 4 broadcast table rows per instruction — and every score leaves as a
// bf16 hi / lo pair.  P = (x . t) * scale(t): the clip's scale (modules.py embedding max_norm, oracle S2) on the dot.
template <int LDT>
__global__ __launch_bounds__(256) void time_scores_kernel(const ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int k = blockIdx.y;
  const int off = time_rowoff(k), nk = time_rowoff(k + 1) - off;
  constexpr int ldt = LDT, ls = LDT + 4;
  const int ic = 2 * a.d.ldh;
  float* tl = lds;                 // [61][ls]
  float* sc = tl + 61 * ls;        // [64]
  float* xl = sc + 64;             // [16][ls]
  constexpr int sub = LDT >> 2;
  const long b0 = (long)blockIdx.x * 16;
  const float* tab = pick5(a.tab, k);
  const int nf = nk * sub;
#pragma unroll 2
  for (int f0 = 0; f0 < nf; f0 += 256) {
    const int f = f0 + tid;
    const int row = f / sub, lin = f - row * sub;
    const bool valid = row < nk;
    const float4 x = valid ? ld4(tab + (long)row * ldt + lin * 4) : zero4();
    const float ss = group_sum(dot4(x, x), sub);
    if (valid) {
      st4(tl + row * ls + lin * 4, x);
      if (lin == 0) sc[row] = clip_scale(ss);
      if (a.tclip && blockIdx.x == 0) {      // the step's backward kernels read the clipped rows from here (one writer per table)
        st4(a.tclip + (long)(off + row) * ldt + lin * 4, scale4(x, clip_scale(ss)));
        if (lin == 0) {
          a.tclip[160 * ldt + off + row] = clip_scale(ss);
          a.tclip[160 * ldt + 160 + off + row] = ss > 1.0f ? 1.0f : 0.0f;
        }
      }
    }
  }
  for (int f = tid; f < 16 * sub; f += 256) {
    const int r = f / sub, c = f - r * sub;
    const float4 x = b0 + r < a.B ? ld4(a.attout + (b0 + r) * a.ld_att + ic + k * ldt + c * 4) : zero4();
    st4(xl + r * ls + c * 4, x);
  }
  __syncthreads();
  const int b = tid & 15, rg = tid >> 4;
  const float* x = xl + b * ls;
  const int nz = k == 4 ? 160 - 139 : 0;         // the last table's workgroups also zero the padding columns
#pragma unroll 2
  for (int r = rg; r < nk + nz; r += 16) {
    float v = 0.f;
    if (r < nk) {
      const float* t = tl + r * ls;
      float4 s4 = zero4();
#pragma unroll 16
      for (int i = 0; i < ldt; i += 4) {
        const float4 u = *reinterpret_cast<const float4*>(x + i), w = *reinterpret_cast<const float4*>(t + i);
        s4.x = fmaf(u.x, w.x, s4.x); s4.y = fmaf(u.y, w.y, s4.y); s4.z = fmaf(u.z, w.z, s4.z); s4.w = fmaf(u.w, w.w, s4.w);
      }
      v = ((s4.x + s4.y) + (s4.z + s4.w)) * sc[r];
    }
    const __bf16 h = (__bf16)v;
    const long o = kb32_off(b0 + b, off + r, a.in32);
    a.ph[o] = h;
    a.pl[o] = (__bf16)(v - (float)h);
  }
}

// ---- output transforms finished + time scores in ONE launch (round 4) ------------------------------------------------------------
// attout = tanh(pooled W + b) (model_combine.py:119,127,132) used to be one grouped small GEMM of 104 workgroups walking 8 / 5
// serial 64-deep stages (41 us beside the deferred Adam rest pass, 23 alone) followed by the time-score launch.  Now the GEMM runs
// as ONE 128-deep K chunk per workgroup into split-K slabs (376 workgroups, the form of the projections and of the backward's
// dpooled) and THIS kernel folds the slabs in slab order, adds the bias, applies tanh, writes attout, its hi / lo planes and the
// packed item | time planes (operand of dE), and — in the workgroups that own a time table's 64 columns — goes straight on to the
// scores of tcar_time_scores_clip with the rows still in LDS.  grid = (16-session blocks, 5 tables + ic / 64 column blocks).
struct FinishArgs {
  ScoreArgs s;                    // tab, B, ph / pl (may be NULL: no scores), in32, tclip; attout = output here
  const float* slabs; int nd_ic, nd_pt; long stride;      // slab k of the item|content problem at slabs + k * stride (columns [0, ic)),
                                                          // of the time problem at slabs + ic + k * stride (columns [ic, ek)); row stride ek
  const float* b_o; const float* b_ot;
  float* out; long ld_out;
  __bf16* a_hi; __bf16* a_lo; int a_in32;                 // planes of attout (may be NULL)
  __bf16* ap_hi; __bf16* ap_lo; int ap_in32;              // packed [item | time] planes (may be NULL)
  // anchor of the anchored softmax form (score.hip: ce_anchor_fold_kernel; label NULL: none): per row and 64-column block j of the
  // item | content columns, the partial sum of attout[b, :] . E[label[b], :] — the label's score without its time part — leaves as
  // P[b, 139 + j] = -partial (hi / lo), against the ones of the one-hot plane's anchor columns (time_onehot_kernel)
  const int32_t* label; const float* E; long ldE;
};
__device__ __forceinline__ void store_planes4(__bf16* hi, __bf16* lo, long o, float4 y) {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_f;
  const bf16x4_f h = {(__bf16)y.x, (__bf16)y.y, (__bf16)y.z, (__bf16)y.w};
  const bf16x4_f l = {(__bf16)(y.x - (float)h[0]), (__bf16)(y.y - (float)h[1]), (__bf16)(y.z - (float)h[2]), (__bf16)(y.w - (float)h[3])};
  *reinterpret_cast<bf16x4_f*>(hi + o) = h;
  *reinterpret_cast<bf16x4_f*>(lo + o) = l;
}
__global__ __launch_bounds__(256) void attout_finish_kernel(const FinishArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int ldt = 64, ls = 68, sub = 16;
  const int tid = threadIdx.x;
  const int ldh = a.s.d.ldh, ic = 2 * ldh, ek = ic + 5 * ldt;
  const long b0 = (long)blockIdx.x * 16;
  const int y = blockIdx.y;
  const bool timeblk = y < 5;
  const int col0 = timeblk ? ic + y * ldt : (y - 5) * 64;
  float* tl = lds;                 // [61][ls]
  float* sc = tl + 61 * ls;        // [64]
  float* xl = sc + 64;             // [16][ls]
  const bool scores = timeblk && a.s.ph != nullptr;
  int off = 0, nk = 0;
  if (timeblk) {
    off = time_rowoff(y); nk = time_rowoff(y + 1) - off;
    if (scores || (a.s.tclip && blockIdx.x == 0)) {
      const float* tab = pick5(a.s.tab, y);
      const int nf = nk * sub;
#pragma unroll 2
      for (int f0 = 0; f0 < nf; f0 += 256) {
        const int f = f0 + tid;
        const int row = f / sub, lin = f - row * sub;
        const bool valid = row < nk;
        const float4 x = valid ? ld4(tab + (long)row * ldt + lin * 4) : zero4();
        const float ss = group_sum(dot4(x, x), sub);
        if (valid) {
          st4(tl + row * ls + lin * 4, x);
          if (lin == 0) sc[row] = clip_scale(ss);
          if (a.s.tclip && blockIdx.x == 0) {
            st4(a.s.tclip + (long)(off + row) * ldt + lin * 4, scale4(x, clip_scale(ss)));
            if (lin == 0) {
              a.s.tclip[160 * ldt + off + row] = clip_scale(ss);
              a.s.tclip[160 * ldt + 160 + off + row] = ss > 1.0f ? 1.0f : 0.0f;
            }
          }
        }
      }
    }
  }
  // fold + bias + tanh of this workgroup's 16 x 64 block: thread (r, c) = one float4
  {
    const int r = tid >> 4, c = tid & 15;
    const long row = b0 + r;
    const int col = col0 + c * 4;
    float4 yv = zero4();
    if (row < a.s.B) {
      const float* sp = a.slabs + row * ek + col;
      const int nd = timeblk ? a.nd_pt : a.nd_ic;
      // (slab order: fixed; the first MS slabs' loads are issued together, not as a chain of dependent round trips)
      constexpr int MS = 8;
      float4 ts[MS];
#pragma unroll
      for (int k = 0; k < MS; ++k) ts[k] = (k == 0 || k < nd) ? ld4(sp + (long)k * a.stride) : zero4();
      const float4 bb = timeblk ? ld4(a.b_ot + (col - ic)) : ld4(a.b_o + col);
      float4 acc = ts[0];
#pragma unroll
      for (int k = 1; k < MS; ++k)
        if (k < nd) acc = add4(acc, ts[k]);
      for (int k = MS; k < nd; ++k) acc = add4(acc, ld4(sp + (long)k * a.stride));
      yv = make_float4(tanhf(acc.x + bb.x), tanhf(acc.y + bb.y), tanhf(acc.z + bb.z), tanhf(acc.w + bb.w));
      st4(a.out + row * a.ld_out + col, yv);
      if (a.a_hi) store_planes4(a.a_hi, a.a_lo, kb32_off(row, col, a.a_in32), yv);
      if (a.ap_hi && (col < ldh || col >= ic))
        store_planes4(a.ap_hi, a.ap_lo, kb32_off(row, col < ldh ? col : col - (ic - ldh), a.ap_in32), yv);
    }
    if (scores) st4(xl + r * ls + c * 4, yv);
    if (a.label && !timeblk) {          // (workgroup-uniform) one partial per (row, column block): a fixed 16-lane tree
      float dl = 0.f;
      if (row < a.s.B) dl = dot4(yv, ld4(a.E + (long)clampi(a.label[row], 0, a.s.d.n_items - 1) * a.ldE + col));
      dl = -group_sum(dl, 16);
      if (c == 0) {                       // (padding rows of the 16-row block: zero, like every padding row of P)
        const __bf16 h = (__bf16)dl;
        const long o = kb32_off(row, 139 + (y - 5), a.s.in32);
        a.s.ph[o] = h;
        a.s.pl[o] = (__bf16)(dl - (float)h);
      }
    }
  }
  if (!scores) return;
  __syncthreads();
  const int b = tid & 15, rg = tid >> 4;
  const float* x = xl + b * ls;
  // the last table's workgroups also zero the padding columns — but for the anchor columns the item | content workgroups fill (above)
  const int nanc = a.label ? (ic >> 6) : 0;
  const int nz = y == 4 ? 160 - 139 : 0;
#pragma unroll 2
  for (int r = rg; r < nk + nz; r += 16) {
    float v = 0.f;
    if (r < nk) {
      const float* t = tl + r * ls;
      float4 s4 = zero4();
#pragma unroll 16
      for (int i = 0; i < ldt; i += 4) {
        const float4 u = *reinterpret_cast<const float4*>(x + i), w = *reinterpret_cast<const float4*>(t + i);
        s4.x = fmaf(u.x, w.x, s4.x); s4.y = fmaf(u.y, w.y, s4.y); s4.z = fmaf(u.z, w.z, s4.z); s4.w = fmaf(u.w, w.w, s4.w);
      }
      v = ((s4.x + s4.y) + (s4.z + s4.w)) * sc[r];
    }
    if (r >= nk && r - nk < nanc) continue;       // an anchor column
    const __bf16 h = (__bf16)v;
    const long o = kb32_off(b0 + b, off + r, a.s.in32);
    a.s.ph[o] = h;
    a.s.pl[o] = (__bf16)(v - (float)h);
  }
}

// gradient of the candidate-side lookups: for every (n, k) clip-backward of d_et[n, k*ldt ...]
__global__ __launch_bounds__(256) void cand_time_bwd_kernel(const CandArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [139, ldt] accumulators + [5] norms
  const int tid = threadIdx.x, lane = tid & 63;
  const int ldt = a.d.ldt, pt = 5 * ldt;
  const int sub = ldt >> 2, gpw = 64 / sub;
  const int grp = lane / sub, lin = lane - grp * sub;
  float* acc = lds;
  float* sq_acc = lds + 139 * ldt;
  for (int i = tid; i < 139 * ldt + 8; i += 256) lds[i] = 0.f;
  __syncthreads();
  float sq[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const long npairs = (long)a.d.n_items * 5;
  const long wave_g = (long)blockIdx.x * 4 + (tid >> 6);
  const long nwaves = (long)gridDim.x * 4;
  for (long p0 = wave_g * gpw; p0 < npairs; p0 += nwaves * gpw) {
    const long p = p0 + grp;
    const bool valid = p < npairs;
    const long pp = valid ? p : 0;
    const int k = (int)(pp % 5);
    const long n = pp / 5;
    const int id = clampi(a.mwdhm[pp], 0, time_vocab(k) - 1);
    float4 x = valid ? ld4(pick5(a.tab, k) + (long)id * ldt + lin * 4) : zero4();
    float4 gy = valid ? ld4(a.d_et + n * pt + k * ldt + lin * 4) : zero4();
    const float ss = group_sum(dot4(x, x), sub), dd = group_sum(dot4(x, gy), sub);
    float ca, cb;
    clip_bwd_coef(ss, dd, ca, cb);
    if (valid) {
      float4 gx = fma4(x, -cb, scale4(gy, ca));
      const float q = dot4(gx, gx);
#pragma unroll
      for (int s = 0; s < 5; ++s) sq[s] += (s == k) ? q : 0.f;
      atomic_add4(acc + (time_rowoff(k) + id) * ldt + lin * 4, gx);
    }
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const float s = wave_sum(sq[i]);
    if (lane == 0 && s != 0.f) atomicAdd(sq_acc + i, s);
  }
  __syncthreads();
  for (int i = tid; i < 139 * ldt; i += 256) {
    const float v = acc[i];
    if (v != 0.f) atomicAdd(a.g.g_time[0] + i, v);
  }
  if (tid < 5) {
    const float v = sq_acc[tid];
    if (v != 0.f) atomicAdd(a.g.sqn + a.g.slot_time[tid], v);
  }
}

// ---- candidate-side time gradient through a STATIC inverted index (deterministic, no atomics) ---------------
// publish_time_MWDHM never changes, so the candidates that share table row (k, v) are listed once on the host:
// inv_n[inv_off[r] .. inv_off[r+1]) for r = rowoff(k) + v.  The clip Jacobian depends only on the table row x, so
//   sum_n gx_n          = a*S - x*inv^3*(x.S),   S = sum_n gy_n         (linear in the list sum)
//   sum_n ||gx_n||^2    = inv^2*Q - inv^4*D2,    Q = sum ||gy_n||^2, D2 = sum (x.gy_n)^2     (n over the list, ||x||>1)
// Pass 1: grid (139 rows, CH chunks); each 16-lane group streams its candidates' 256-byte d_et segments and keeps
// S, Q, D2; fixed-order reductions; partials to a workspace.  Pass 2: one workgroup folds the chunks in order,
// applies the Jacobian, adds into the time-table gradients and the norm pieces.
constexpr int CT_CHUNKS = 32;

__global__ __launch_bounds__(256) void cand_time_bwd_idx_kernel(const CandArgs a, const int32_t* __restrict__ inv_n,
                                                                const int32_t* __restrict__ inv_off,
                                                                float* __restrict__ ws, int permuted) {
  __shared__ __attribute__((aligned(16))) float shS[16 * 256];   // [groups][ldt]  (groups*ldt = 1024 floats)
  __shared__ float shQ[16], shD[16];
  const int tid = threadIdx.x;
  const int ldt = a.d.ldt, pt = 5 * ldt, sub = ldt >> 2, groups = 256 / sub;
  const int grp = tid / sub, lin = tid - grp * sub;
  const int r = blockIdx.x, c = blockIdx.y;
  const int k = r < 13 ? 0 : r < 45 ? 1 : r < 53 ? 2 : r < 78 ? 3 : 4;
  const int v = r - time_rowoff(k);
  const int lo = inv_off[r], hi = inv_off[r + 1];
  const int per = (hi - lo + CT_CHUNKS - 1) / CT_CHUNKS;
  const int s0 = lo + c * per, s1 = min(hi, s0 + per);
  const float4 x = ld4(pick5(a.tab, k) + (long)v * ldt + lin * 4);
  float4 S = zero4();
  float Q = 0.f, D2 = 0.f;
  // 4 candidates per group and trip: index loads, then row loads, all independent (memory-level parallelism)
  for (int i0 = s0 + grp; i0 < s1; i0 += 4 * groups) {
    long n[4];
    float4 gy[4];
    if (permuted) {         // d_et is in list order: entry i is the contiguous segment i (no index load, pure streaming)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * groups;
        gy[u] = (i < s1) ? ld4(a.d_et + (long)i * ldt + lin * 4) : zero4();
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * groups;
        n[u] = (i < s1) ? (long)inv_n[i] : -1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) gy[u] = (n[u] >= 0) ? ld4(a.d_et + n[u] * pt + k * ldt + lin * 4) : zero4();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      S = add4(S, gy[u]);
      Q += dot4(gy[u], gy[u]);
      const float d = group_sum(dot4(x, gy[u]), sub);
      D2 += (lin == 0) ? d * d : 0.f;
    }
  }
  Q = group_sum(Q, sub);
  st4(shS + grp * ldt + lin * 4, S);
  if (lin == 0) { shQ[grp] = Q; shD[grp] = D2; }
  __syncthreads();
  float* out = ws + ((long)r * CT_CHUNKS + c) * (ldt + 4);
  if (tid < sub) {
    float4 t = zero4();
    for (int g2 = 0; g2 < groups; ++g2) t = add4(t, ld4(shS + g2 * ldt + tid * 4));
    st4(out + tid * 4, t);
  }
  if (tid == 0) {
    float q = 0.f, d2 = 0.f;
    for (int g2 = 0; g2 < groups; ++g2) { q += shQ[g2]; d2 += shD[g2]; }
    out[ldt] = q;
    out[ldt + 1] = d2;
  }
}

// pass 2a: one workgroup per table row folds its CT_CHUNKS partials (each 16-lane group takes chunks g, g+16, ...;
// the groups' sums meet in LDS and are added in fixed order) and applies the clip Jacobian
__global__ __launch_bounds__(256) void cand_time_bwd_fin_kernel(const CandArgs a, float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) float shS[16 * 256];
  __shared__ float shQ[16], shD[16];
  const int tid = threadIdx.x;
  const int ldt = a.d.ldt, sub = ldt >> 2, groups = 256 / sub;
  const int grp = tid / sub, lin = tid - grp * sub;
  const int r = blockIdx.x;
  const int k = r < 13 ? 0 : r < 45 ? 1 : r < 53 ? 2 : r < 78 ? 3 : 4;
  float4 S = zero4();
  float Q = 0.f, D2 = 0.f;
  for (int c = grp; c < CT_CHUNKS; c += groups) {
    const float* p = ws + ((long)r * CT_CHUNKS + c) * (ldt + 4);
    S = add4(S, ld4(p + lin * 4));
    Q += p[ldt];
    D2 += p[ldt + 1];
  }
  st4(shS + grp * ldt + lin * 4, S);
  if (lin == 0) { shQ[grp] = Q; shD[grp] = D2; }
  __syncthreads();
  if (tid < sub) {                                            // the first 16-lane group finishes the row
    S = zero4(); Q = 0.f; D2 = 0.f;
    for (int g2 = 0; g2 < groups; ++g2) { S = add4(S, ld4(shS + g2 * ldt + lin * 4)); Q += shQ[g2]; D2 += shD[g2]; }
    const float4 x = ld4(pick5(a.tab, k) + (long)(r - time_rowoff(k)) * ldt + lin * 4);
    const float ss = group_sum(dot4(x, x), sub), xs = group_sum(dot4(x, S), sub);
    float4 gx = S;
    float pc = Q;
    if (ss > 1.0f) {
      const float inv = 1.0f / sqrtf(ss), inv2 = inv * inv;
      gx = fma4(x, -(xs * inv2 * inv), scale4(S, inv));
      pc = inv2 * Q - inv2 * inv2 * D2;
    }
    float* gp = a.g.g_time[0] + (long)r * ldt + lin * 4;       // month..minute gradients are contiguous
    atomic_add4(gp, gx);       // atomics: the session-side gather backward may be adding into the same rows concurrently
    if (lin == 0) ws[(long)139 * CT_CHUNKS * (ldt + 4) + r] = pc;
  }
}
// pass 2b: per-table norm pieces: one wave per table, fixed lane assignment + shuffle tree (deterministic)
__global__ __launch_bounds__(64) void cand_time_bwd_piece_kernel(const CandArgs a, const float* __restrict__ ws) {
  const int k = blockIdx.x, lane = threadIdx.x;
  const float* pc = ws + (long)139 * CT_CHUNKS * (a.d.ldt + 4) + time_rowoff(k);
  const float s = wave_sum(lane < time_vocab(k) ? pc[lane] : 0.f);
  if (lane == 0) atomicAdd(a.g.sqn + pick5(a.g.slot_time, k), s);
}

// ---- candidate-side time gradient of the ONE-HOT form (round 4): no [N, 5 ldt] block of dE exists ---------------------------------
// The dE GEMM's epilogue (gemm_bf16.hip, EPI = 2) left, per (candidate n, table k) in the order of the inverted index,
// (q, z) = (||gy||^2, x . gy) for gy = dE[n, time block k] and x = the CLIPPED row the candidate looks up; the dX GEMM left
// dP = dlogits OH [B, 160] (summed over its split-K slabs by tcar_reduce_dact_onehot).  With the list of row r:
//   Q = sum q,  D2c = sum z^2,  S = sum_n gy_n = sum_b dP[b, r] attout_t,k[b, :]          (the list sum is LINEAR in dlogits)
//   clipped row (x = raw * sc, sc = 1 / ||raw||):  sum gx = sc (S - x (x . S)),  sum ||gx||^2 = sc^2 (Q - D2c);  else S, Q
// One workgroup per table row, fixed summation orders (bit-for-bit repeatable).  ldt = 64.
__global__ __launch_bounds__(256) void cand_time_bwd_onehot_kernel(int B, int ek, int ic, const float2* __restrict__ qz,
                                                                  const int32_t* __restrict__ inv_off, const float* __restrict__ dP,
                                                                  const float* __restrict__ attout, const float* __restrict__ tclip,
                                                                  float* __restrict__ g_time, float* __restrict__ pc_out,
                                                                  const TcarWait wait_dp) {
  __shared__ __attribute__((aligned(16))) float shS[16 * 64];
  __shared__ float shQ[256], shD[256];
  const int tid = threadIdx.x;
  const int r = blockIdx.x;
  const int k = r < 13 ? 0 : r < 45 ? 1 : r < 53 ? 2 : r < 78 ? 3 : 4;
  const int lo = inv_off[r], hi = inv_off[r + 1];
  float Q = 0.f, D2 = 0.f;
  // (all loads of a trip are issued before the first use: the kernel is a handful of dependent memory round trips, not bytes)
  for (int i0 = lo + tid; i0 < hi; i0 += 256 * 8) {
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (i0 + 256 * u < hi) ? qz[i0 + 256 * u] : make_float2(0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) { Q += v[u].x; D2 = fmaf(v[u].y, v[u].y, D2); }
  }
  shQ[tid] = Q; shD[tid] = D2;
  tcar_wave_wait(wait_dp);       // dP comes from the main chain's slab reduce: waited for here, behind the list pass above
  const int j4 = tid & 15, bg = tid >> 4;
  float4 S = zero4();
  const float* ap = attout + ic + k * 64 + j4 * 4;
  for (int b0 = bg; b0 < B; b0 += 16 * 8) {
    float4 x[8];
    float w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + 16 * u;
      const bool ok = b < B;
      x[u] = ok ? ld4(ap + (long)b * ek) : zero4();
      w[u] = ok ? (wait_dp.flag ? ld1_sc1(dP + (long)b * 160 + r) : dP[(long)b * 160 + r]) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) S = fma4(x[u], w[u], S);
  }
  st4(shS + bg * 64 + j4 * 4, S);
  __syncthreads();
  if (tid < 16) {
    S = zero4(); Q = 0.f; D2 = 0.f;
    for (int g2 = 0; g2 < 16; ++g2) S = add4(S, ld4(shS + g2 * 64 + tid * 4));
    for (int t = tid; t < 256; t += 16) { Q += shQ[t]; D2 += shD[t]; }     // lane t of 16 takes every 16th partial, in order
    Q = group_sum(Q, 16);
    D2 = group_sum(D2, 16);
    const float4 x = ld4(tclip + (long)r * 64 + tid * 4);
    const float sc = tclip[160 * 64 + r];
    const bool clipped = tclip[160 * 64 + 160 + r] != 0.f;
    const float xs = group_sum(dot4(x, S), 16);
    float4 gx = S;
    float pc = Q;
    if (clipped) {
      gx = scale4(fma4(x, -xs, S), sc);
      pc = sc * sc * (Q - D2);
    }
    atomic_add4(g_time + (long)r * 64 + tid * 4, gx);      // (the session-side small-table backward adds into the same rows)
    if (tid == 0) pc_out[r] = pc;
  }
}

// g_item[ids[r]-1] += rows[r]: one wave per row, 256-byte-contiguous float atomics
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(int ldh, int n_items, const int32_t* __restrict__ ids,
                                                               const float* __restrict__ rows, long R,
                                                               float* __restrict__ g_item) {
  const int lane = threadIdx.x & 63;
  const long wave_g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  for (long r = wave_g; r < R; r += nwaves) {
    const int id = ids[r];
    if (id < 1 || id > n_items) continue;
    for (int col = lane * 4; col < ldh; col += 256)
      atomic_add4(g_item + (long)(id - 1) * ldh + col, ld4(rows + r * ldh + col));
  }
}

// ---- order-fixed backward of the SMALL tables (position, five time tables, dwell) on the session side -----------------
// Every source that gathers table row r sees the SAME clipped row x, so (as on the candidate side above) the sum of the
// per-source clip Jacobians needs only S = sum gy, Q = sum ||gy||^2 and D2 = sum (x . gy)^2 over those sources:
//   sum gx = S / n - x (x . S) / n^3,    sum ||gx||^2 = Q / n^2 - D2 / n^4      (n = ||x|| > 1; identity otherwise)
// One workgroup of 16 waves per destination row: wave w walks the w-th sixteenth of the source rows IN ORDER (ids 64 at a
// time, ballot, the matches' gradient slices four loads at a time), the 16 partial (S, Q, D2) are folded in wave order, the
// Jacobian is applied once and the row is added to the gradient with ONE atomic per element (the candidate side adds its one
// contribution to the same rows: two addends commute).  Nothing depends on the dispatch order.
struct SmallDetArgs {
  tcar_dims_t d;
  tcar_tables_t tab;
  tcar_batch_t bt;
  const float* dx_icp; const float* dx_pt; const float* dx_act; const float* dclick;
  float* g_pos; float* g_small;      // [40, ldh]; [150, ldt] = month | day | week | hour | minute | dwell
  float* rowq;                        // [SMALL_DET_ROWS] per-row norm pieces (folded per table by small_norm_fold_kernel)
  int CH; float* part;                // CH > 1: the sources of every row in CH chunks, each its own workgroup leaving (S, q, D) in part
};
constexpr int SMALL_DET_ROWS = TCAR_POS_VOCAB + SMALL_ROWS + 1;   // + the out-of-range dwell bucket (norm only, S7)
constexpr int SMALL_DET_CH = 32;                                  // most chunks per row (workspace: rows x chunks x (64 NC + 2) floats)
constexpr int SMALL_DET_PW = 64 * 8 + 2;                          // floats of one partial (NC <= 8)
// the ch-th of CH equal parts of [0, n), and the wv-th sixteenth of that part
__device__ __forceinline__ void det_range(int n, int CH, int ch, int wv, int& r0, int& r1) {
  const int perc = (n + CH - 1) / CH, c0 = min(n, ch * perc), c1 = min(n, c0 + perc);
  const int per = (c1 - c0 + 15) / 16;
  r0 = min(c1, c0 + wv * per);
  r1 = min(c1, r0 + per);
}
// what a destination row R is: kind 0 position row, 1 time / dwell row sr of table k (value v), 2 the out-of-range dwell bucket
struct DetRow { int kind, k, v, cols, sr; const float* xrow; };
__device__ __forceinline__ DetRow det_row(const SmallDetArgs& a, int R, int T) {
  DetRow r{0, 0, 0, 0, 0, nullptr};
  const int ldh = a.d.ldh, ldt = a.d.ldt;
  if (R < T) { r.kind = 0; r.cols = ldh; r.xrow = a.tab.pos + (long)R * ldh; }
  else if (R < T + SMALL_ROWS) {
    r.kind = 1; r.sr = R - T; r.cols = ldt;
    const int sr = r.sr;
    r.k = sr < 13 ? 0 : sr < 45 ? 1 : sr < 53 ? 2 : sr < 78 ? 3 : sr < 139 ? 4 : 5;
    r.v = sr - time_rowoff(r.k);
    r.xrow = (r.k < 5 ? pick5(a.tab.time, r.k) : a.tab.dur) + (long)r.v * ldt;
  } else { r.kind = 2; r.k = 5; r.cols = ldt; }
  return r;
}
// the end of a destination row (one wave): clip Jacobian once per row on the summed sources, the gradient row, the norm piece
template <int NC>
__device__ __forceinline__ void det_finish(const SmallDetArgs& a, const DetRow& r, int R, int lane, const float (&x)[NC],
                                           const float (&St)[NC], float Q, float D) {
  const int ldh = a.d.ldh, ldt = a.d.ldt;
  float ss = 0.f, xs = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) { ss += x[c] * x[c]; xs += x[c] * St[c]; }
  ss = wave_sum(ss);
  xs = wave_sum(xs);
  float pc = Q;
  float ca = 1.f, cb = 0.f;
  if (r.kind != 2 && ss > 1.0f) {
    const float inv = 1.0f / sqrtf(ss), inv2 = inv * inv;
    ca = inv;
    cb = xs * inv2 * inv;
    pc = inv2 * Q - inv2 * inv2 * D;
  }
  if (r.kind != 2) {
    float* dst = (r.kind == 0) ? a.g_pos + (long)R * ldh : a.g_small + (long)r.sr * ldt;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = lane + 64 * c;
      const float gx = ca * St[c] - cb * x[c];
      if (col < r.cols && gx != 0.f) atomicAdd(dst + col, gx);
    }
  }
  if (lane == 0) a.rowq[r.kind == 0 ? R : TCAR_POS_VOCAB + (r.kind == 1 ? r.sr : SMALL_ROWS)] = pc;
}

template <int NC>   // 64-column groups per lane: columns <= 64 * NC
__global__ __launch_bounds__(1024) void small_tables_bwd_det_kernel(const SmallDetArgs a) {
  __shared__ float partS[16][64 * NC];
  __shared__ float partQ[16], partD[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int B = a.bt.B, T = a.bt.T, BT = B * T;
  const int ldh = a.d.ldh, ldt = a.d.ldt, ic = 2 * ldh, pt = 5 * ldt, ct = 2 * ldt;
  // [0, T) position rows | [T, T + 150) small-table rows | T + 150: dwell out of range; CH > 1: chunk ch of the row's sources
  const int nrows = T + SMALL_ROWS + 1;
  const int R = blockIdx.x % nrows, ch = blockIdx.x / nrows, CH = a.CH;
  const DetRow dr = det_row(a, R, T);
  const int kind = dr.kind, k = dr.k, v = dr.v, cols = dr.cols;
  const float* xrow = dr.xrow;
  (void)ldh;
  float x[NC], S[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = lane + 64 * c;
    x[c] = (xrow && col < cols) ? xrow[col] : 0.f;
    S[c] = 0.f;
  }
  float ql = 0.f, D2 = 0.f;
  // acc of up to U sources whose gradient slices start at p[0..n): ALL loads first (one memory round trip for the group), then the sums
  // in source order — the order of the sums does not depend on U.  Position rows are ldh wide (NC registers per source, 8 in flight);
  // the time / dwell rows are ldt <= 64 wide: one register per source, 16 in flight (round 5: a wave of the week table's workgroups
  // walks (B*T / 7) / 16 matches — 4 at a time that was 8 dependent round trips at T = 7, 46 at T = 40: 72 / 400 us on the chain
  // that ends the step)
  auto takeP = [&](const float* const* p, int n) {
    float g[8][NC];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = lane + 64 * c;
        g[u][c] = (u < n && col < cols) ? p[u][col] : 0.f;
      }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u >= n) break;
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) { S[c] += g[u][c]; ql += g[u][c] * g[u][c]; d += x[c] * g[u][c]; }
      d = wave_sum(d);
      D2 += d * d;
    }
  };
  // (up to 16 matches of the ballot mask m, in ascending lane order; src(j) = gradient slice of the match at lane j.  cols <= 64:
  //  column group 0 only — x and S of the other groups stay 0)
  auto takeS = [&](unsigned long long& m, auto&& src) {
    float g[16], d[16];
    int n = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const bool v = m != 0;                   // (wave-uniform)
      const int j = v ? __builtin_ctzll(m) : 0;
      if (v) { m &= m - 1; n = u + 1; }
      const float* q = src(j);
      g[u] = (v && lane < cols) ? q[lane] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) d[u] = wave_sum(x[0] * g[u]);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (u < n) {
        S[0] += g[u]; ql += g[u] * g[u];
        D2 += d[u] * d[u];
      }
    }
  };
  if (kind == 0) {
    // position t = R: the sources are the rows b * T + t, no ids to match
    int b0, b1;
    det_range(B, CH, ch, wv, b0, b1);
    for (int b = b0; b < b1; b += 8) {
      const float* p[8];
      const int n = min(8, b1 - b);
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = a.dx_icp + ((long)(b + (u < n ? u : 0)) * T + R) * ic;
      takeP(p, n);
    }
  } else {
    // session rows: id of table k at every source row; matches of this wave's sixteenth, in order
    const int32_t* ids = (k < 5) ? pick5(a.bt.pub, k) : a.bt.gap;
    int r0, r1;
    det_range(BT, CH, ch, wv, r0, r1);
    for (int base = r0; base < r1; base += 64) {
      const int row = base + lane;
      bool hit = false;
      if (row < r1) {
        const int id = ids[row];
        const bool oob = (k == 5) && (id >= TCAR_DUR_VOCAB || id < 0);
        hit = (kind == 2) ? oob : (!oob && clampi(id, 0, time_vocab(k) - 1) == v);
      }
      unsigned long long m = __ballot(hit);
      while (m)
        takeS(m, [&](int j) {
          const long rr = base + j;
          return (k < 5) ? a.dx_pt + rr * pt + k * ldt : a.dx_act + rr * ldt;
        });
    }
    // click rows: the week table by cw, the hour table by ch (model_combine.py:94-97)
    if (kind == 1 && (k == 2 || k == 3)) {
      const int32_t* cid = (k == 2) ? a.bt.cw : a.bt.ch;
      const int jj = (k == 2) ? 0 : 1;
      int b0, b1;
      det_range(B, CH, ch, wv, b0, b1);
      for (int base = b0; base < b1; base += 64) {
        const int b = base + lane;
        const bool hit = (b < b1) && clampi(cid[b], 0, time_vocab(k) - 1) == v;
        unsigned long long m = __ballot(hit);
        while (m) takeS(m, [&](int j) { return a.dclick + (long)(base + j) * ct + jj * ldt; });
      }
    }
  }
  ql = wave_sum(ql);
#pragma unroll
  for (int c = 0; c < NC; ++c) partS[wv][lane + 64 * c] = S[c];
  if (lane == 0) { partQ[wv] = ql; partD[wv] = D2; }
  __syncthreads();
  if (wv != 0) return;
  float St[NC];
  float Q = 0.f, D = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) St[c] = 0.f;
  for (int w = 0; w < 16; ++w) {                // the sixteenths in order
#pragma unroll
    for (int c = 0; c < NC; ++c) St[c] += partS[w][lane + 64 * c];
    Q += partQ[w];
    D += partD[w];
  }
  if (CH > 1) {                                  // this chunk's partial; small_tables_fold_kernel adds the chunks in order
    float* pp = a.part + ((long)R * CH + ch) * SMALL_DET_PW;
#pragma unroll
    for (int c = 0; c < NC; ++c) pp[lane + 64 * c] = St[c];
    if (lane == 0) { pp[64 * NC] = Q; pp[64 * NC + 1] = D; }
    return;
  }
  det_finish<NC>(a, dr, R, lane, x, St, Q, D);
}

// CH > 1: one wave per destination row adds the CH chunk partials in chunk order and ends the row
template <int NC>
__global__ __launch_bounds__(64) void small_tables_fold_kernel(const SmallDetArgs a) {
  const int lane = threadIdx.x, R = blockIdx.x, T = a.bt.T, CH = a.CH;
  const DetRow dr = det_row(a, R, T);
  float x[NC], St[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = lane + 64 * c;
    x[c] = (dr.xrow && col < dr.cols) ? dr.xrow[col] : 0.f;
    St[c] = 0.f;
  }
  float Q = 0.f, D = 0.f;
  const float* pp = a.part + (long)R * CH * SMALL_DET_PW;
  for (int ch = 0; ch < CH; ++ch, pp += SMALL_DET_PW) {
#pragma unroll
    for (int c = 0; c < NC; ++c) St[c] += pp[lane + 64 * c];
    Q += pp[64 * NC];
    D += pp[64 * NC + 1];
  }
  det_finish<NC>(a, dr, R, lane, x, St, Q, D);
}

// per-table norm pieces from the per-row ones, rows in order; one add per slot (the candidate side adds its one)
__global__ __launch_bounds__(64) void small_norm_fold_kernel(const float* __restrict__ rowq, int T, float* __restrict__ sqn, int slot_pos,
                                                             int s0, int s1, int s2, int s3, int s4, int slot_dur,
                                                             const TcarSignal sig, const float* __restrict__ cand_pc) {
  // one 64-lane wave: table k = 0 position, 1..5 month..minute, 6 dwell (+ its out-of-range bucket); lane i takes row i of the
  // table (every table has <= 64 rows) — the session side's piece plus, in the one-hot form, the candidate side's piece of the
  // same row (cand_time_bwd_onehot_kernel) — and the wave's shuffle tree adds them up: a fixed order, one memory round trip
  // (the round-3 form walked the rows in a dependent scalar loop: 6 us, 19 us with the candidate pieces)
  const int lane = threadIdx.x;
#pragma unroll
  for (int k = 0; k <= 6; ++k) {
    int lo, n, slot;
    if (k == 0) { lo = 0; n = T; slot = slot_pos; }
    else if (k <= 5) { lo = TCAR_POS_VOCAB + time_rowoff(k - 1); n = time_vocab(k - 1); slot = k == 1 ? s0 : k == 2 ? s1 : k == 3 ? s2 : k == 4 ? s3 : s4; }
    else { lo = TCAR_POS_VOCAB + 139; n = TCAR_DUR_VOCAB + 1; slot = slot_dur; }
    float v = lane < n ? rowq[lo + lane] : 0.f;
    if (cand_pc && k >= 1 && k <= 5 && lane < n) v += cand_pc[time_rowoff(k - 1) + lane];
    const float s = wave_sum(v);
    if (lane == 0 && s != 0.f) atomicAdd(sqn + slot, s);
  }
  tcar_signal_done(sig);        // (the fused step joins the aux stream into the main one behind this launch)
}

// packed form: the id rides behind its row (row stride ld), ids shifted by id0 (catalog shards)
__global__ __launch_bounds__(256) void scatter_add_rows_packed_kernel(int ldh, int n_items, int id0, const float* __restrict__ packed,
                                                                      long ld, long R, float* __restrict__ g_item) {
  const int lane = threadIdx.x & 63;
  const long wave_g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  for (long r = wave_g; r < R; r += nwaves) {
    const float* row = packed + r * ld;
    const int id = __float_as_int(row[ldh]) - id0;
    if (id < 1 || id > n_items) continue;
    for (int col = lane * 4; col < ldh; col += 256) atomic_add4(g_item + (long)(id - 1) * ldh + col, ld4(row + col));
  }
}

int check_dims(const tcar_dims_t* d) {
  if (!d || d->n_items <= 0 || d->H <= 0 || d->Ht <= 0) return TCAR_E_ARG;
  if (d->ldh < d->H || d->ldt < d->Ht || (d->ldh & 63) || (d->ldh > 512)) return TCAR_E_ARG;
  if (d->ldt != 64 && d->ldt != 128 && d->ldt != 256) return TCAR_E_ARG;
  return TCAR_OK;
}
int grid_for_rows(long rows) {
  long g = (rows + 3) / 4;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;          // 8 workgroups x 4 waves per CU: 32 waves/CU, each with a ~3.5 KB row in flight
  return (int)g;
}

}  // namespace

extern "C" int tcar_gather_clip_fwd(const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt,
                                    float* x_icp, float* x_pt, float* x_act, float* click_t, void* stream) {
  return tcar_gather_clip_fwd_o(d, tab, bt, x_icp, x_pt, x_act, click_t, stream, nullptr);
}
extern "C" int tcar_gather_clip_fwd_tuned(const tcar_tuning_t* tune, const tcar_dims_t* d, const tcar_tables_t* tab,
                                          const tcar_batch_t* bt, float* x_icp, float* x_pt, float* x_act, float* click_t,
                                          void* stream) {
  TcarOpt o;
  o.tune = tune;
  return tcar_gather_clip_fwd_o(d, tab, bt, x_icp, x_pt, x_act, click_t, stream, &o);
}
// (flag-capable in the latency form: the click rows leave write-through when the launch carries a flag; the throughput form is not)
int tcar_gather_clip_fwd_o(const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt, float* x_icp, float* x_pt,
                           float* x_act, float* click_t, void* stream, TcarOpt* o) {
  const TcarTuning& tn = tcar_tn(o);
  if (check_dims(d) || !tab || !bt || bt->B <= 0 || bt->T <= 0 || bt->T > TCAR_POS_VOCAB) return TCAR_E_ARG;
  EmbArgs a{};
  a.d = *d; a.tab = *tab; a.bt = *bt;
  a.x_icp = x_icp; a.x_pt = x_pt; a.x_act = x_act; a.click_t = click_t;
  const long rows = (long)bt->B * bt->T;
  const size_t big_lds = ((size_t)TCAR_POS_VOCAB * d->ldh + (size_t)SMALL_ROWS * d->ldt) * sizeof(float);
  if (rows >= tn.gather_big_rows && big_lds <= 160 * 1024) {
    // throughput form: one 16-wave workgroup per CU (the clipped small tables live in its LDS)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // (TWO consecutive rows per wave and trip, non-temporal stores, two workgroups per CU — round 5, tools/gather_bench.py: 0.680-0.717 ms
    //  at 655,360 rows against 0.702-0.728 for four rows; eight rows, one row, plain stores and one workgroup per CU all measured slower)
    const int rpw = 2;
    long g = (rows + 16 * rpw - 1) / (16 * rpw);
    const long cap = (long)cus * tcar_fixed::gather_wg;
    if (g > cap) g = cap;
    if (d->ldh <= 256) {
      TCAR_SET_LDS_ONCE(gather_clip_fwd_big_kernel<1>, 160 * 1024);
      TCAR_LAUNCH(gather_clip_fwd_big_kernel<1>, dim3((int)g), dim3(1024), big_lds, (hipStream_t)stream, a);
    } else {
      TCAR_SET_LDS_ONCE(gather_clip_fwd_big_kernel<2>, 160 * 1024);
      TCAR_LAUNCH(gather_clip_fwd_big_kernel<2>, dim3((int)g), dim3(1024), big_lds, (hipStream_t)stream, a);
    }
    TCAR_CHECK_LAUNCH();
    return TCAR_OK;
  }
  const int grid = grid_for_rows((long)bt->B * bt->T + bt->B);
  a.sig = tcar_sig(o);
  if (d->ldh <= 256) TCAR_LAUNCH(gather_clip_fwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else TCAR_LAUNCH(gather_clip_fwd_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

namespace {
int gather_bwd_launch(const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt, const float* dx_icp, const float* dx_pt,
                      const float* dx_act, const float* dclick, const tcar_grads_t* g, const float* sq_g, long sq_len,
                      float* sq_part, int sq_blocks, void* stream) {
  if (check_dims(d) || !tab || !bt || !g || bt->B <= 0 || bt->T <= 0 || bt->T > TCAR_POS_VOCAB) return TCAR_E_ARG;
  EmbArgs a{};
  a.d = *d; a.tab = *tab; a.bt = *bt; a.g = *g;
  a.dx_icp = dx_icp; a.dx_pt = dx_pt; a.dx_act = dx_act; a.dclick = dclick;
  long rows = (long)bt->B * bt->T + bt->B;
  // rows are a latency chain per wave (ids -> rows -> reductions -> atomics): one row per wave until the chip is
  // full (512 workgroups), more beyond; the LDS zero / flush per workgroup costs ~2 us
  int grid = (int)((rows + 3) / 4);
  if (grid < 1) grid = 1;
  if (grid > 512) grid = 512;
  if (sq_blocks > 0) { a.sq_g = sq_g; a.sq_len = sq_len; a.sq_part = sq_part; a.n_gather = grid; }
  const int total = grid + (sq_blocks > 0 ? sq_blocks : 0);
  const size_t lds = ((size_t)bt->T * d->ldh + (size_t)SMALL_ROWS * d->ldt + 8) * sizeof(float);
  if (lds > 160 * 1024) return TCAR_E_ARG;
  hipStream_t st = (hipStream_t)stream;
#define TCAR_GBWD(NCH_, LDT_)                                                                                         \
  do {                                                                                                                \
    TCAR_SET_LDS_ONCE((gather_clip_bwd_kernel<NCH_, LDT_>), 160 * 1024);                                              \
    TCAR_LAUNCH((gather_clip_bwd_kernel<NCH_, LDT_>), dim3(total), dim3(256), lds, st, a);                            \
  } while (0)
  if (d->ldh <= 256) {
    if (d->ldt == 64) TCAR_GBWD(1, 64); else if (d->ldt == 128) TCAR_GBWD(1, 128); else TCAR_GBWD(1, 256);
  } else {
    if (d->ldt == 64) TCAR_GBWD(2, 64); else if (d->ldt == 128) TCAR_GBWD(2, 128); else TCAR_GBWD(2, 256);
  }
#undef TCAR_GBWD
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
}  // namespace

extern "C" int tcar_gather_clip_bwd(const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt,
                                    const float* dx_icp, const float* dx_pt, const float* dx_act,
                                    const float* dclick, const tcar_grads_t* g, void* stream) {
  return gather_bwd_launch(d, tab, bt, dx_icp, dx_pt, dx_act, dclick, g, nullptr, 0, nullptr, 0, stream);
}

// the same launch also computes tcar_sqnorm_det's block partials of sum sq_g[0:sq_len]^2 into the last 4096 bytes of the segsum
// workspace `ws` (512 floats: one per extra workgroup): the two passes are independent and both sit between the negative rows and
// the session rows of the sorted item-row sum
extern "C" int tcar_gather_clip_bwd_sqnorm(const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt,
                                           const float* dx_icp, const float* dx_pt, const float* dx_act, const float* dclick,
                                           const tcar_grads_t* g, const float* sq_g, int64_t sq_len, void* ws, int64_t ws_bytes,
                                           void* stream) {
  if (!sq_g || sq_len <= 0 || (sq_len & 3) || !tcar_aligned16(sq_g) || !ws || ws_bytes < 4096) return TCAR_E_ARG;
  return gather_bwd_launch(d, tab, bt, dx_icp, dx_pt, dx_act, dclick, g, sq_g, (long)sq_len, (float*)((char*)ws + ws_bytes - 4096),
                           512, stream);
}
extern "C" int tcar_cand_time_bwd_indexed(const tcar_dims_t* d, const float* const time_tab[5], const int32_t* inv_n,
                                         const int32_t* inv_off, const float* d_et, int permuted, float* ws,
                                         const tcar_grads_t* g, void* stream) {
  if (check_dims(d) || !time_tab || !inv_n || !inv_off || !d_et || !ws || !g) return TCAR_E_ARG;
  CandArgs a{};
  a.d = *d;
  for (int k = 0; k < 5; ++k) a.tab[k] = time_tab[k];
  a.d_et = d_et; a.g = *g;
  TCAR_LAUNCH(cand_time_bwd_idx_kernel, dim3(139, CT_CHUNKS), dim3(256), 0, (hipStream_t)stream, a, inv_n, inv_off, ws, permuted);
  TCAR_CHECK_LAUNCH();
  TCAR_LAUNCH(cand_time_bwd_fin_kernel, dim3(139), dim3(256), 0, (hipStream_t)stream, a, ws);
  TCAR_CHECK_LAUNCH();
  TCAR_LAUNCH(cand_time_bwd_piece_kernel, dim3(5), dim3(64), 0, (hipStream_t)stream, a, (const float*)ws);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// One-hot form of tcar_cand_time_bwd_indexed (see cand_time_bwd_onehot_kernel): qz [5 N] float2 in inverted-index order from
// tcar_gemm_bf16_de_qz, dP [B, 160] from tcar_reduce_dact_onehot, tclip from tcar_time_scores_clip; adds into g->g_time and the
// norm pieces exactly as tcar_cand_time_bwd_indexed does (same workspace ws).  ldt must be 64.
extern "C" int tcar_cand_time_bwd_onehot(const tcar_dims_t* d, int B, const int32_t* inv_off, const float* qz, const float* dP,
                                         const float* attout, int64_t ld_att, const float* tclip, float* ws, const tcar_grads_t* g,
                                         void* stream) {
  return tcar_cand_time_bwd_onehot_w(d, B, inv_off, qz, dP, attout, ld_att, tclip, ws, g, stream, TcarWait{}, 1);
}
// wait_dp: dP is produced on another stream behind a completion flag (the kernel waits itself); with_pieces = 0 leaves the
// per-table fold of the norm pieces to tcar_small_tables_bwd_det_o (cand_pc), one launch less on the chain
int tcar_cand_time_bwd_onehot_w(const tcar_dims_t* d, int B, const int32_t* inv_off, const float* qz, const float* dP,
                                const float* attout, int64_t ld_att, const float* tclip, float* ws, const tcar_grads_t* g, void* stream,
                                const TcarWait& wait_dp, int with_pieces) {
  if (check_dims(d) || d->ldt != 64 || B <= 0 || !inv_off || !qz || !dP || !attout || !tclip || !ws || !g || (ld_att & 3) ||
      !tcar_aligned16(attout) || !tcar_aligned16(tclip))
    return TCAR_E_ARG;
  CandArgs a{};
  a.d = *d;
  a.g = *g;
  float* pc = ws + (long)139 * CT_CHUNKS * (d->ldt + 4);
  hipStream_t st = (hipStream_t)stream;
  TCAR_LAUNCH(cand_time_bwd_onehot_kernel, dim3(139), dim3(256), 0, st, B, (int)ld_att, 2 * d->ldh, (const float2*)qz, inv_off, dP,
              attout, tclip, g->g_time[0], pc, wait_dp);
  TCAR_CHECK_LAUNCH();
  if (with_pieces) {
    TCAR_LAUNCH(cand_time_bwd_piece_kernel, dim3(5), dim3(64), 0, st, a, (const float*)ws);
    TCAR_CHECK_LAUNCH();
  }
  return TCAR_OK;
}

extern "C" int tcar_cand_time_ws_floats(const tcar_dims_t* d) { return d ? 139 * CT_CHUNKS * (d->ldt + 4) + 144 : 0; }

extern "C" int tcar_scatter_add_rows(const tcar_dims_t* d, const int32_t* ids, const float* rows, int64_t R,
                                     float* g_item, void* stream) {
  if (R <= 0) return TCAR_OK;
  if (check_dims(d) || !ids || !rows || !g_item) return TCAR_E_ARG;
  int grid = (int)((R + 3) / 4);
  if (grid > 2048) grid = 2048;
  TCAR_LAUNCH(scatter_add_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d->ldh, d->n_items, ids, rows,
              (long)R, g_item);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_scatter_add_rows_packed(const tcar_dims_t* d, const float* packed, int64_t ld, int64_t R, int32_t id0,
                                            float* g_item, void* stream) {
  if (R <= 0) return TCAR_OK;
  if (check_dims(d) || !packed || !g_item || ld < d->ldh + 1 || (ld & 3) || !tcar_aligned16(packed)) return TCAR_E_ARG;
  int grid = (int)((R + 3) / 4);
  if (grid > 2048) grid = 2048;
  TCAR_LAUNCH(scatter_add_rows_packed_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d->ldh, d->n_items, (int)id0, packed,
              (long)ld, (long)R, g_item);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_cand_time_fwd(const tcar_dims_t* d, const float* const time_tab[5], const int32_t* mwdhm,
                                  float* E, void* stream) {
  return tcar_cand_time_fwd_bf16(d, time_tab, mwdhm, E, nullptr, nullptr, stream);
}

extern "C" int tcar_cand_time_fwd_bf16(const tcar_dims_t* d, const float* const time_tab[5], const int32_t* mwdhm,
                                       float* E, void* e16_hi, void* e16_lo, void* stream) {
  if (check_dims(d) || !time_tab || !mwdhm || (!E && !e16_hi) || (e16_hi && !e16_lo)) return TCAR_E_ARG;
  CandArgs a{};
  a.d = *d;
  for (int k = 0; k < 5; ++k) a.tab[k] = time_tab[k];
  a.mwdhm = mwdhm; a.E = E; a.eh = (__bf16*)e16_hi; a.el = (__bf16*)e16_lo;
  const size_t lds = (size_t)139 * d->ldt * sizeof(float);
  int grid = (d->n_items + 127) / 128;      // one 128-row block per workgroup (grid-stride beyond 1024 blocks)
  if (grid > 1024) grid = 1024;
  TCAR_SET_LDS_ONCE(cand_time_fwd_kernel, 160 * 1024);
  TCAR_LAUNCH(cand_time_fwd_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_cand_time_bwd(const tcar_dims_t* d, const float* const time_tab[5], const int32_t* mwdhm,
                                  const float* d_et, const tcar_grads_t* g, void* stream) {
  if (check_dims(d) || !time_tab || !mwdhm || !d_et || !g) return TCAR_E_ARG;
  CandArgs a{};
  a.d = *d;
  for (int k = 0; k < 5; ++k) a.tab[k] = time_tab[k];
  a.mwdhm = mwdhm; a.d_et = d_et; a.g = *g;
  const size_t lds = ((size_t)139 * d->ldt + 8) * sizeof(float);
  long npairs = (long)d->n_items * 5;
  const int gpw = 64 / (d->ldt >> 2);
  int grid = (int)((npairs / gpw + 4 * 16 - 1) / (4 * 16));   // >= 16 passes per wave
  if (grid < 1) grid = 1;
  if (grid > 512) grid = 512;
  TCAR_SET_LDS_ONCE(cand_time_bwd_kernel, 160 * 1024);
  TCAR_LAUNCH(cand_time_bwd_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// Order-fixed session-side backward of the position, time and dwell tables (see small_tables_bwd_det_kernel); the item rows
// are tcar_gather_clip_bwd's with g->skip_small set.  ws: >= tcar_small_det_ws_floats() floats.
extern "C" int tcar_small_det_ws_floats(void) { return SMALL_DET_ROWS + SMALL_DET_ROWS * SMALL_DET_CH * SMALL_DET_PW; }
extern "C" int tcar_small_tables_bwd_det(const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt, const float* dx_icp,
                                         const float* dx_pt, const float* dx_act, const float* dclick, const tcar_grads_t* g,
                                         float* ws, void* stream) {
  return tcar_small_tables_bwd_det_o(d, tab, bt, dx_icp, dx_pt, dx_act, dclick, g, ws, stream, nullptr, nullptr, tcar_small_det_ws_floats());
}
// (flag-capable: the norm fold, its last launch, publishes its slots with atomics)
int tcar_small_tables_bwd_det_o(const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt, const float* dx_icp,
                                const float* dx_pt, const float* dx_act, const float* dclick, const tcar_grads_t* g, float* ws,
                                void* stream, TcarOpt* o, const float* cand_pc, int64_t ws_floats) {
  if (check_dims(d) || !tab || !bt || !g || !ws || bt->B <= 0 || bt->T <= 0 || bt->T > TCAR_POS_VOCAB || ws_floats < SMALL_DET_ROWS)
    return TCAR_E_ARG;
  if (!dx_icp || !dx_pt || !dx_act || !dclick || !g->g_pos || !g->g_time[0] || !g->sqn) return TCAR_E_ARG;
  // (the time / dwell / click rows take the one-column-group path, takeS: a lane owns one of 64 columns.  A wider time hidden size
  //  would leave columns >= 64 out of the sums — refused, not silently wrong)
  if (d->ldt != 64) return TCAR_E_ARG;
  SmallDetArgs a{};
  a.d = *d; a.tab = *tab; a.bt = *bt;
  a.dx_icp = dx_icp; a.dx_pt = dx_pt; a.dx_act = dx_act; a.dclick = dclick;
  a.g_pos = g->g_pos; a.g_small = g->g_time[0]; a.rowq = ws;
  const int rows = bt->T + SMALL_ROWS + 1;
  const int widest = d->ldh > d->ldt ? d->ldh : d->ldt;
  hipStream_t st = (hipStream_t)stream;
  // Long buckets: the sources of every destination row in CH chunks of ~1,024 rows, each chunk its own workgroup, and one wave per row
  // that adds the chunk partials in chunk order (a fixed order again).  A fold where every session shares a month — every real one —
  // sends ALL B * T sources to one row of that table: one workgroup then walks them all (71 us at T = 7, 365 us at T = 40, on the
  // chain that ends the step; tools/small_det_bench.py).  Short buckets (B * T < 2,048) keep the single pass and its bits.
  const long BT = (long)bt->B * bt->T;
  // (chunks of 512 / 256 / 128 sources measured 42 / 61 / 94 us at T = 7 against 41 us: a 1,024-thread workgroup costs ~15 us whatever it finds)
  int CH = BT >= 2048 ? (int)(BT / 1024) : 1;
  if (CH > SMALL_DET_CH) CH = SMALL_DET_CH;
  if (ws_floats < (int64_t)SMALL_DET_ROWS + (int64_t)SMALL_DET_ROWS * CH * SMALL_DET_PW) CH = 1;      // (a caller with the row pieces only)
  a.CH = CH; a.part = ws + SMALL_DET_ROWS;
  if (widest <= 256) TCAR_LAUNCH(small_tables_bwd_det_kernel<4>, dim3(rows * CH), dim3(1024), 0, st, a);
  else TCAR_LAUNCH(small_tables_bwd_det_kernel<8>, dim3(rows * CH), dim3(1024), 0, st, a);
  TCAR_CHECK_LAUNCH();
  if (CH > 1) {
    if (widest <= 256) TCAR_LAUNCH(small_tables_fold_kernel<4>, dim3(rows), dim3(64), 0, st, a);
    else TCAR_LAUNCH(small_tables_fold_kernel<8>, dim3(rows), dim3(64), 0, st, a);
    TCAR_CHECK_LAUNCH();
  }
  TCAR_LAUNCH(small_norm_fold_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, bt->T, g->sqn, g->slot_pos, g->slot_time[0],
              g->slot_time[1], g->slot_time[2], g->slot_time[3], g->slot_time[4], g->slot_dur, tcar_sig(o), cand_pc);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// OH[n, rowoff_k + mwdhm[n, k]] = 1 (k = 0..4), everything else 0: bf16 KB32 plane [ceil128(N), inner], inner >= 160.  Static: it
// depends on publish_time_MWDHM only.
extern "C" int tcar_time_onehot(const tcar_dims_t* d, const int32_t* mwdhm, void* oh_hi, int64_t inner, void* stream) {
  if (check_dims(d) || !mwdhm || !oh_hi || (inner & 31) || inner < 160) return TCAR_E_ARG;
  const long rows = ((long)d->n_items + 127) & ~127L;
  if (hipMemsetAsync(oh_hi, 0, (size_t)rows * inner * 2, (hipStream_t)stream) != hipSuccess) return TCAR_E_LAUNCH;
  TCAR_LAUNCH(time_onehot_kernel, dim3((unsigned)((d->n_items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d->n_items, mwdhm,
              (__bf16*)oh_hi, (int)(inner >> 5));
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// P[b, r] = attout[b, 2 ldh + k(r) ldt ...] . clip(time row r)  (r = 0..138 in month|day|week|hour|minute order) as bf16 hi / lo
// KB32 planes [ceil128(B), inner] (columns >= 139 and rows >= B zero): the A operand of the one-hot K segment of the logits GEMM
extern "C" int tcar_time_scores(const tcar_dims_t* d, const float* const time_tab[5], int B, const float* attout, int64_t ld_att,
                                void* p_hi, void* p_lo, int64_t inner, void* stream) {
  return tcar_time_scores_clip(d, time_tab, B, attout, ld_att, p_hi, p_lo, inner, nullptr, stream);
}
// attout [B, ek] (row stride ld_out) = tanh(sum_k slabs_k + bias): slabs of the two output transforms as tcar_gemm_x3_grouped leaves
// them with splitk = nd_ic / nd_pt (slab k at slabs + k * stride, row stride ek, the time problem's columns at + 2 ldh); optional
// bf16 hi / lo planes of attout (inner a_inner) and packed [item | time] planes (inner ap_inner = ldh + 5 ldt); optional time scores
// P (p_hi / p_lo, inner >= 160) and clipped rows (tclip) exactly as tcar_time_scores_clip computes them.  ldt == 64, ldh % 64 == 0.
extern "C" int tcar_attout_finish_scores(const tcar_dims_t* d, const float* const time_tab[5], int B, const float* slabs, int nd_ic,
                                         int nd_pt, int64_t stride, const float* bias_o, const float* bias_ot, float* attout,
                                         int64_t ld_out, void* a_hi, void* a_lo, int64_t a_inner, void* ap_hi, void* ap_lo,
                                         int64_t ap_inner, void* p_hi, void* p_lo, int64_t p_inner, float* tclip, void* stream) {
  return tcar_attout_finish_scores_a(d, time_tab, B, slabs, nd_ic, nd_pt, stride, bias_o, bias_ot, attout, ld_out, a_hi, a_lo, a_inner,
                                     ap_hi, ap_lo, ap_inner, p_hi, p_lo, p_inner, tclip, nullptr, nullptr, 0, stream);
}
// ... + the anchor of the anchored softmax form (label != NULL; needs the score planes): P[b, 139 + j] = minus the j-th 64-column
// partial of attout[b, :] . E[label[b], :] over the item | content columns (2 ldh / 64 <= 8 of them).  E [n_items, ldE] fp32 candidate
// rows (item | content columns first), label [B] 0-based
int tcar_attout_finish_scores_a(const tcar_dims_t* d, const float* const time_tab[5], int B, const float* slabs, int nd_ic, int nd_pt,
                                int64_t stride, const float* bias_o, const float* bias_ot, float* attout, int64_t ld_out, void* a_hi,
                                void* a_lo, int64_t a_inner, void* ap_hi, void* ap_lo, int64_t ap_inner, void* p_hi, void* p_lo,
                                int64_t p_inner, float* tclip, const int32_t* label, const float* E, int64_t ldE, void* stream) {
  if (label && (!p_hi || !E || ldE < 2 * d->ldh || (ldE & 3) || !tcar_aligned16(E) || 2 * d->ldh / 64 > TCAR_ANCHOR_COLS)) return TCAR_E_ARG;
  if (check_dims(d) || d->ldt != 64 || (d->ldh & 63) || !time_tab || B <= 0 || !slabs || nd_ic <= 0 || nd_pt <= 0 || !bias_o || !bias_ot ||
      !attout || (ld_out & 3) || !tcar_aligned16(slabs) || !tcar_aligned16(attout) || (stride & 3))
    return TCAR_E_ARG;
  if (a_hi && (!a_lo || (a_inner & 31) || a_inner < 2 * d->ldh + 5 * d->ldt)) return TCAR_E_ARG;
  if (ap_hi && (!ap_lo || (ap_inner & 31) || ap_inner < d->ldh + 5 * d->ldt)) return TCAR_E_ARG;
  if (p_hi && (!p_lo || (p_inner & 31) || p_inner < 160)) return TCAR_E_ARG;
  FinishArgs a{};
  a.s.d = *d;
  for (int k = 0; k < 5; ++k) a.s.tab[k] = time_tab[k];
  a.s.B = B; a.s.in32 = (int)(p_inner >> 5); a.s.ph = (__bf16*)p_hi; a.s.pl = (__bf16*)p_lo; a.s.tclip = tclip;
  a.slabs = slabs; a.nd_ic = nd_ic; a.nd_pt = nd_pt; a.stride = (long)stride; a.b_o = bias_o; a.b_ot = bias_ot;
  a.out = attout; a.ld_out = (long)ld_out;
  a.a_hi = (__bf16*)a_hi; a.a_lo = (__bf16*)a_lo; a.a_in32 = (int)(a_inner >> 5);
  a.ap_hi = (__bf16*)ap_hi; a.ap_lo = (__bf16*)ap_lo; a.ap_in32 = (int)(ap_inner >> 5);
  a.label = label; a.E = E; a.ldE = (long)ldE;
  const long Bp = p_hi ? (((long)B + 127) & ~127L) : (((long)B + 15) & ~15L);      // the score planes' padding rows are written (zeros)
  const size_t lds = ((size_t)(61 + 16) * 68 + 64) * sizeof(float);
  TCAR_LAUNCH(attout_finish_kernel, dim3((unsigned)(Bp / 16), 5 + 2 * d->ldh / 64), dim3(256), lds, (hipStream_t)stream, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

int tcar_anchor_scores(int ldh, int B, const float* attout, int64_t ld_att, const int32_t* label, const float* E, int64_t ldE,
                       int64_t n_rows, void* p_hi, void* p_lo, int64_t inner, void* stream) {
  if (B <= 0) return TCAR_OK;
  if (ldh <= 0 || (ldh & 1) || !attout || (ld_att & 3) || ld_att < 2 * ldh || !label || !E || (ldE & 3) || ldE < 2 * ldh || n_rows <= 0 ||
      !p_hi || !p_lo || (inner & 31) || inner < 160 || !tcar_aligned16(attout) || !tcar_aligned16(E))
    return TCAR_E_ARG;
  TCAR_LAUNCH(anchor_scores_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, 2 * ldh, B, attout, (long)ld_att,
              label, E, (long)ldE, (long)n_rows, (__bf16*)p_hi, (__bf16*)p_lo, (int)(inner >> 5));
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_time_scores_clip(const tcar_dims_t* d, const float* const time_tab[5], int B, const float* attout,
                                     int64_t ld_att, void* p_hi, void* p_lo, int64_t inner, float* tclip, void* stream) {
  if (check_dims(d) || !time_tab || B <= 0 || !attout || !p_hi || !p_lo || (inner & 31) || inner < 160 || (ld_att & 3)) return TCAR_E_ARG;
  ScoreArgs a{};
  a.d = *d;
  for (int k = 0; k < 5; ++k) a.tab[k] = time_tab[k];
  a.attout = attout; a.ld_att = ld_att; a.B = B; a.in32 = (int)(inner >> 5);
  a.ph = (__bf16*)p_hi; a.pl = (__bf16*)p_lo;
  a.tclip = tclip;
  const long Bp = ((long)B + 127) & ~127L;
  const size_t lds = ((size_t)(61 + 16) * (d->ldt + 4) + 64) * sizeof(float);
  const dim3 grid((unsigned)(Bp / 16), 5);
  if (d->ldt == 64) {
    TCAR_LAUNCH(time_scores_kernel<64>, grid, dim3(256), lds, (hipStream_t)stream, a);
  } else if (d->ldt == 128) {
    TCAR_LAUNCH(time_scores_kernel<128>, grid, dim3(256), lds, (hipStream_t)stream, a);
  } else {
    TCAR_SET_LDS_ONCE(time_scores_kernel<256>, lds);
    TCAR_LAUNCH(time_scores_kernel<256>, grid, dim3(256), lds, (hipStream_t)stream, a);
  }
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
