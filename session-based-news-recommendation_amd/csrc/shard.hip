// Kernels of the catalog-sharded data-parallel step (sharded.py).  With W ranks, rank r owns the catalog rows
// [n0, n0 + Nloc) of the candidate matrix, scores them against the sessions of ALL ranks (attout is all-gathered), and keeps
// their gradient, Adam moments and bf16 planes to itself: the dense item gradient never crosses a link.  What the split of
// the softmax over the catalog (model_combine.py:145) needs beyond the single-GPU kernels:
//   tcar_softmax_stats     per (session, shard): max, sum exp(x - max) and the label's logit when the label lives here
//   tcar_softmax_combine   folds the W all-gathered stat triples of every session into lse and the cross entropy
//   tcar_softmax_grad      dlogits = exp(x - lse) - onehot as bf16 hi / lo KB32 planes of the shard's columns
//   tcar_neg_scatter_range the negative-term rows of ALL sessions that fall into this shard (model_combine.py:142, backward)
#include "tcar_common.h"
#include "tcar_bf16_layout.h"

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_h;

namespace {

__global__ __launch_bounds__(256) void softmax_stats_kernel(int B, int N, const float* __restrict__ logits, long ld,
                                                            const int32_t* __restrict__ label, int n0, float* __restrict__ stats) {
  __shared__ float sh[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* row = logits + (long)b * ld;
  float m = -INFINITY;
  for (int c = tid * 4; c < N; c += 1024) {
    const float4 v = ld4(row + c);
    m = fmaxf(m, v.x);
    if (c + 1 < N) m = fmaxf(m, v.y);
    if (c + 2 < N) m = fmaxf(m, v.z);
    if (c + 3 < N) m = fmaxf(m, v.w);
  }
  m = wave_max(m);
  if (lane == 0) sh[w] = m;
  __syncthreads();
  const float gm = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = tid * 4; c < N; c += 1024) {
    const float4 v = ld4(row + c);
    s += expf(v.x - gm);
    if (c + 1 < N) s += expf(v.y - gm);
    if (c + 2 < N) s += expf(v.z - gm);
    if (c + 3 < N) s += expf(v.w - gm);
  }
  s = wave_sum(s);
  if (lane == 0) sh[w] = s;
  __syncthreads();
  if (tid == 0) {
    const int lab = label[b] - n0;
    stats[3L * b + 0] = gm;
    stats[3L * b + 1] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    stats[3L * b + 2] = (lab >= 0 && lab < N) ? row[lab] : 0.f;
  }
}

// the same from ONE read of the row: it is held in registers between the max and the sum (rows of up to NT * 4 * R columns)
template <int NT, int R>
__global__ __launch_bounds__(NT) void softmax_stats_rows_kernel(int B, int N, const float* __restrict__ logits, long ld,
                                                                const int32_t* __restrict__ label, int n0, float* __restrict__ stats) {
  __shared__ float sh[NT / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* row = logits + (long)b * ld;
  const float ninf = -INFINITY;
  float4 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int c = (tid + r * NT) * 4;
    v[r] = (c < N) ? ld4(row + c) : make_float4(ninf, ninf, ninf, ninf);        // ld >= ceil4(N): the load stays inside the row
  }
  float m = ninf;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int c = (tid + r * NT) * 4;
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
  for (int r = 0; r < R; ++r) s += (expf(v[r].x - gm) + expf(v[r].y - gm)) + (expf(v[r].z - gm) + expf(v[r].w - gm));
  s = wave_sum(s);
  if (lane == 0) sh[w] = s;
  __syncthreads();
  if (tid == 0) {
    float gs = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) gs += sh[i];
    const int lab = label[b] - n0;
    stats[3L * b + 0] = gm;
    stats[3L * b + 1] = gs;
    stats[3L * b + 2] = (lab >= 0 && lab < N) ? row[lab] : 0.f;
  }
}

// stats_all [W, B, 3] -> lse [B], ce [B] = lse - label logit (the label lives in exactly one shard; the others sent 0)
__global__ __launch_bounds__(256) void softmax_combine_kernel(int W, int B, const float* __restrict__ stats_all,
                                                              const int32_t* __restrict__ label, float* __restrict__ lse,
                                                              float* __restrict__ ce, float* __restrict__ rowstat) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  // rowstat (softmax-epilogue form): the pair tcar_ce_rescale scales the exp plane with — exp(m_g - lse) * 1
  if (rowstat) rowstat[2 * b + 1] = 1.f;
  if (label && label[b] < 0) {          // padding session of an uneven shard: lse = +inf makes its gradient row exactly zero
    lse[b] = INFINITY;
    if (rowstat) rowstat[2 * b] = INFINITY;
    if (ce) ce[b] = 0.f;
    return;
  }
  float m = -INFINITY;
  for (int w = 0; w < W; ++w) m = fmaxf(m, stats_all[((long)w * B + b) * 3]);
  float s = 0.f, lab = 0.f;
  for (int w = 0; w < W; ++w) {                      // fixed order: every rank folds the same numbers the same way
    const float* t = stats_all + ((long)w * B + b) * 3;
    s += t[1] * expf(t[0] - m);
    lab += t[2];
  }
  const float l = m + logf(s);
  lse[b] = l;
  if (rowstat) rowstat[2 * b] = l;
  if (ce) ce[b] = l - lab;
}

// anchored form: stats_all [W, B, 3] = (sum of the plane's rounded entries, sum of the exponentials, label's accumulator) per shard,
// every shard's exponentials relative to the SAME per-session anchor: plain sums in rank order.  lse (relative to the anchor) and ce
// from the exponentials' sum; rowstat[b] = (rounded sum, 1 / true sum) for tcar_ce_anchor_apply_o
__global__ __launch_bounds__(256) void softmax_combine_anchored_kernel(int W, int B, const float* __restrict__ stats_all,
                                                                       const int32_t* __restrict__ label, float* __restrict__ lse,
                                                                       float* __restrict__ ce, float* __restrict__ rowstat) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  if (label && label[b] < 0) {          // padding session: an exactly zero gradient row (the apply kernel tests the label itself)
    lse[b] = INFINITY;
    rowstat[2 * b] = INFINITY; rowstat[2 * b + 1] = 0.f;
    if (ce) ce[b] = 0.f;
    return;
  }
  float sr = 0.f, s = 0.f, lab = 0.f;
  for (int w = 0; w < W; ++w) {
    const float* t = stats_all + ((long)w * B + b) * 3;
    sr += t[0]; s += t[1]; lab += t[2];
  }
  const float l = logf(s);
  lse[b] = l;
  rowstat[2 * b] = sr; rowstat[2 * b + 1] = 1.0f / s;
  if (ce) ce[b] = l - lab;
}

__global__ __launch_bounds__(256) void softmax_grad_kernel(int B, int N, const float* __restrict__ logits, long ld,
                                                           const float* __restrict__ lse, const int32_t* __restrict__ label,
                                                           int n0, __bf16* __restrict__ dh, __bf16* __restrict__ dl) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int in32 = (int)(ld >> 5);
  if (b >= B) {   // padding rows of the planes
    for (int c = tid * 4; c < (int)ld; c += 1024) {
      const long o = kb32_off(b, c, in32);
      const bf16x4_h z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
      *reinterpret_cast<bf16x4_h*>(dh + o) = z;
      if (dl) *reinterpret_cast<bf16x4_h*>(dl + o) = z;
    }
    return;
  }
  const float* row = logits + (long)b * ld;
  const float l = lse[b];
  const int lab = label[b] - n0;
  for (int c = tid * 4; c < (int)ld; c += 1024) {
    const float4 v = ld4(row + c);
    float ov[4] = {v.x, v.y, v.z, v.w};
    bf16x4_h h, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float p = (c + j < N) ? expf(ov[j] - l) - ((c + j == lab) ? 1.f : 0.f) : 0.f;
      h[j] = (__bf16)p;
      lo[j] = (__bf16)(p - (float)h[j]);
    }
    const long o = kb32_off(b, c, in32);
    *reinterpret_cast<bf16x4_h*>(dh + o) = h;
    if (dl) *reinterpret_cast<bf16x4_h*>(dl + o) = lo;
  }
}

// one wave per (session, negative) of the GLOBAL batch; rows outside [n0, n0 + n_loc) belong to another rank
__global__ __launch_bounds__(256) void neg_scatter_range_kernel(long BK, int K, int n0, int n_loc, int ldh, long ld_att,
                                                                const int32_t* __restrict__ neg, const float* __restrict__ attout,
                                                                const float* __restrict__ coef, float* __restrict__ g_item) {
  const int lane = threadIdx.x & 63;
  const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wv >= BK) return;
  const long b = wv / K;
  const int n = neg[wv] - n0;
  const float cf = coef[b];
  if (n < 0 || n >= n_loc || cf == 0.f) return;
  const float* a = attout + b * ld_att;
  float* gdst = g_item + (long)n * ldh;
  for (int col = lane; col < ldh; col += 64) atomicAdd(gdst + col, cf * a[col]);
}


// ---- packed exchange rows (one all-gather instead of four) --------------------------------------------------------
// head[b] = [attout[b, 0:ek] | label | coefficient of the negative term | Kc negatives | pad], ints as bits, row stride ld;
// rows b >= B are padding sessions: zero attout / coefficient, label -1, negatives -1
__global__ __launch_bounds__(256) void pack_head_kernel(int B, int cap, int ek, int K, int Kc, const float* __restrict__ attout,
                                                        const int32_t* __restrict__ label, const float* __restrict__ coef,
                                                        const int32_t* __restrict__ neg, float* __restrict__ head, long ld) {
  const int b = blockIdx.x;
  float* row = head + (long)b * ld;
  const bool live = b < B;
  for (int c = threadIdx.x * 4; c < ek; c += 1024) st4(row + c, live ? ld4(attout + (long)b * ek + c) : zero4());
  for (int j = threadIdx.x; j < (int)ld - ek; j += 256) {
    int bits;
    if (j == 0) bits = live ? label[b] : -1;
    else if (j == 1) bits = (live && coef) ? __float_as_int(coef[b]) : 0;
    else if (j - 2 < Kc) bits = (live && neg && j - 2 < K) ? neg[(long)b * K + (j - 2)] : -1;
    else bits = 0;
    row[ek + j] = __int_as_float(bits);
  }
}

// the all-gathered rows back into the contiguous arrays the scoring kernels read
__global__ __launch_bounds__(256) void unpack_head_kernel(int Bq, int ek, int K, const float* __restrict__ head, long ld,
                                                          int32_t* __restrict__ label, float* __restrict__ coef,
                                                          int32_t* __restrict__ neg) {
  const long n = (long)Bq * (2 + K);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / (2 + K);
    const int j = (int)(i - b * (2 + K));
    const float v = head[b * ld + ek + j];
    if (j == 0) label[b] = __float_as_int(v);
    else if (j == 1) { if (coef) coef[b] = v; }
    else if (neg) neg[b * K + (j - 2)] = __float_as_int(v);
  }
}

// ids behind the packed item-row gradients (rows r >= n_live: id 0 = padding, skipped by the scatter), and the loss rows
__global__ __launch_bounds__(256) void pack_ids_kernel(long n_live, long n_total, int ldh, const int32_t* __restrict__ seq,
                                                       float* __restrict__ rows, long ld, int B, const float* __restrict__ ce,
                                                       const float* __restrict__ fb, float weight, float* __restrict__ loss) {
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < n_total; r += (long)gridDim.x * 256)
    rows[r * ld + ldh] = __int_as_float(r < n_live ? seq[r] : 0);
  if (loss)
    for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) loss[b] = ce[b] + weight * fb[b];
}

}  // namespace

extern "C" int tcar_softmax_stats(int B, int N, const float* logits, int64_t ld, const int32_t* label, int n0, float* stats,
                                  void* stream) {
  if (B <= 0) return TCAR_OK;
  if (N <= 0 || ld < N || (ld & 3) || !tcar_aligned16(logits) || !label || !stats) return TCAR_E_ARG;
  if (N <= 512 * 4 * 8)
    TCAR_LAUNCH((softmax_stats_rows_kernel<512, 8>), dim3(B), dim3(512), 0, (hipStream_t)stream, B, N, logits, (long)ld, label, n0, stats);
  else if (N <= 512 * 4 * 24)
    TCAR_LAUNCH((softmax_stats_rows_kernel<512, 24>), dim3(B), dim3(512), 0, (hipStream_t)stream, B, N, logits, (long)ld, label, n0, stats);
  else
    TCAR_LAUNCH(softmax_stats_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, B, N, logits, (long)ld, label, n0, stats);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_softmax_combine(int W, int B, const float* stats_all, const int32_t* label, float* lse, float* ce,
                                    void* stream) {
  return tcar_softmax_combine_rowstat(W, B, stats_all, label, lse, ce, nullptr, stream);
}
extern "C" int tcar_softmax_combine_rowstat(int W, int B, const float* stats_all, const int32_t* label, float* lse, float* ce,
                                            float* rowstat, void* stream) {
  if (B <= 0) return TCAR_OK;
  if (W <= 0 || !stats_all || !lse) return TCAR_E_ARG;
  TCAR_LAUNCH(softmax_combine_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, B, stats_all, label, lse, ce,
              rowstat);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

int tcar_softmax_combine_anchored(int W, int B, const float* stats_all, const int32_t* label, float* lse, float* ce, float* rowstat,
                                  void* stream) {
  if (B <= 0) return TCAR_OK;
  if (W <= 0 || !stats_all || !lse || !rowstat) return TCAR_E_ARG;
  TCAR_LAUNCH(softmax_combine_anchored_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, B, stats_all, label, lse, ce,
              rowstat);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_softmax_grad(int B, int N, const float* logits, int64_t ld, const float* lse, const int32_t* label, int n0,
                                 void* dl_hi, void* dl_lo, void* stream) {
  if (B <= 0) return TCAR_OK;
  if (N <= 0 || ld < N || (ld & 31) || !tcar_aligned16(logits) || !lse || !label || !dl_hi) return TCAR_E_ARG;
  TCAR_LAUNCH(softmax_grad_kernel, dim3((B + 127) & ~127), dim3(256), 0, (hipStream_t)stream, B, N, logits, (long)ld, lse, label, n0,
              (__bf16*)dl_hi, (__bf16*)dl_lo);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_neg_scatter_range(const tcar_dims_t* d, int64_t B, int K, int n0, int n_loc, const int32_t* neg,
                                      const float* attout, int64_t ld_att, const float* coef, float* g_item, void* stream) {
  if (!d || B <= 0 || K <= 0) return TCAR_OK;
  if (!neg || !attout || !coef || !g_item || n_loc <= 0) return TCAR_E_ARG;
  const long waves = (long)B * K;
  TCAR_LAUNCH(neg_scatter_range_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, waves, K, n0, n_loc,
              d->ldh, (long)ld_att, neg, attout, coef, g_item);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_shard_pack_head(int B, int cap, int ek, int K, int Kc, const float* attout, const int32_t* label,
                                    const float* coef, const int32_t* neg, float* head, int64_t ld, void* stream) {
  if (cap <= 0) return TCAR_OK;
  if (B < 0 || B > cap || !head || (ek & 3) || (ld & 3) || ld < ek + 2 + Kc || K > Kc || (B > 0 && (!attout || !label)) ||
      !tcar_aligned16(head) || (B > 0 && !tcar_aligned16(attout)))
    return TCAR_E_ARG;
  TCAR_LAUNCH(pack_head_kernel, dim3(cap), dim3(256), 0, (hipStream_t)stream, B, cap, ek, K, Kc, attout, label, coef, neg, head, (long)ld);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_shard_unpack_head(int Bq, int ek, int K, const float* head, int64_t ld, int32_t* label, float* coef, int32_t* neg,
                                      void* stream) {
  if (Bq <= 0) return TCAR_OK;
  if (!head || !label || ld < ek + 2 + K || (K > 0 && (!coef || !neg))) return TCAR_E_ARG;
  long n = (long)Bq * (2 + K);
  int grid = (int)((n + 255) / 256);
  if (grid > 1024) grid = 1024;
  TCAR_LAUNCH(unpack_head_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, Bq, ek, K, head, (long)ld, label, coef, neg);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_shard_pack_ids(int64_t n_live, int64_t n_total, int ldh, const int32_t* seq, float* rows, int64_t ld, int B,
                                   const float* ce, const float* neg_fb, float weight, float* loss, void* stream) {
  if (n_total <= 0) return TCAR_OK;
  if (!rows || ld < ldh + 1 || n_live < 0 || n_live > n_total || (n_live > 0 && !seq) || (loss && (!ce || !neg_fb))) return TCAR_E_ARG;
  int grid = (int)((n_total + 255) / 256);
  if (grid > 1024) grid = 1024;
  TCAR_LAUNCH(pack_ids_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (long)n_live, (long)n_total, ldh, seq, rows, (long)ld, B,
              ce, neg_fb, weight, loss);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
