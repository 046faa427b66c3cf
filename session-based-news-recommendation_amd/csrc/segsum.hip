// Deterministic sparse backward of the item table: sort by row + segmented wavefront reduction ("hard parts").
//
// The IndexedSlices gradient of the item lookups (model_combine.py:54,142,156) adds one row per gathered session click
// (B*T rows, values through the clip Jacobian) and one row per sampled negative (B*K rows, value coef[b] * attout[b]) into
// the dense gradient of the candidate block.  News batches are head heavy — one article can own 10 % of the rows of a
// batch — and the plain form (float atomics, embed.hip / score.hip) makes the sum depend on arrival order: replicas and
// repeated runs differ in the last bit.  Here the order is fixed:
//   1. keys: (tag << 25 | item row) for every source — tag 0 = session row, 1 = negative row — sorted ONCE per batch with a
//      radix sort (rocPRIM; it only depends on the feed, so it runs on the aux stream under the forward pass);
//   2. work items: every run of equal keys is cut into chunks of <= 16 sources (a news batch has runs of hundreds);
//   3. one wave per work item sums its rows in sorted order (the sort is stable: equal rows keep their source order);
//      single-chunk runs add straight into the dense gradient — ONE writer per destination row, no atomics — multi-chunk
//      runs leave partial rows that a second pass folds in chunk order;
//   4. the per-row norm pieces of tf.clip_by_norm are summed per work item and folded in item order.
// Bit-for-bit repeatable whatever the dispatch order, and one read-modify-write per touched row instead of one atomic
// per (row, source).
#include <cstring>
#include "tcar_common.h"
#include <rocprim/device/device_radix_sort.hpp>

namespace {

constexpr int CH = 16;                 // sources per work item
constexpr int IDX_THREADS = 1024;

struct SegWs {                         // carved out of the caller's workspace (all device pointers)
  unsigned* k_in; unsigned* v_in; unsigned* k_out; unsigned* v_out;
  int* item_start; int* item_len; int* item_flag;      // per work item: first sorted position, sources, bit0 first / bit1 last of its run
  int* counts;                                          // [0] items of the session list, [1] items of the negative list (after it)
  float* norm_part;                                     // per work item: sum of ||row||^2 of its sources (session list)
  float* partial;                                       // [items, ldh] partial rows of multi-chunk runs
  float* rows;                                          // [B*T, ldh] the session sources' gradient rows (written by the gather backward)
  void* sort_tmp; size_t sort_bytes;
};

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

size_t carve(SegWs& w, char* base, long n, int ldh, size_t sort_bytes) {
  size_t o = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += align_up(bytes); return p; };
  w.k_in = (unsigned*)take(4 * n); w.v_in = (unsigned*)take(4 * n); w.k_out = (unsigned*)take(4 * n); w.v_out = (unsigned*)take(4 * n);
  w.item_start = (int*)take(4 * n); w.item_len = (int*)take(4 * n); w.item_flag = (int*)take(4 * n);
  w.counts = (int*)take(16);
  w.norm_part = (float*)take(4 * n);
  w.partial = (float*)take((size_t)4 * n * ldh);
  w.rows = (float*)take((size_t)4 * n * ldh);
  w.sort_tmp = take(sort_bytes); w.sort_bytes = sort_bytes;
  return o;
}

size_t sort_tmp_bytes(long n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr,
                                  (unsigned*)nullptr, (size_t)n, 0, 26, (hipStream_t)0);
  return bytes;
}

__global__ __launch_bounds__(256) void make_keys_kernel(long BT, long BK, int n_items, const int32_t* __restrict__ seq,
                                                        const int32_t* __restrict__ neg, unsigned* __restrict__ k,
                                                        unsigned* __restrict__ v) {
  const long n = BT + BK;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    unsigned key;
    if (i < BT) key = (unsigned)(clampi(seq[i], 1, n_items) - 1);
    else key = (1u << 25) | (unsigned)clampi(neg[i - BT], 0, n_items - 1);
    k[i] = key;
    v[i] = (unsigned)i;
  }
}

// One workgroup walks the sorted keys and emits the work items of both lists (session sources first, then negatives).
// item boundary at position i: the key changes, or the run has reached a multiple of CH sources.
__global__ __launch_bounds__(IDX_THREADS) void build_items_kernel(long n, long BT, const unsigned* __restrict__ ks,
                                                                  int* __restrict__ item_start, int* __restrict__ item_len,
                                                                  int* __restrict__ item_flag, int* __restrict__ counts) {
  __shared__ int sh_cnt[IDX_THREADS];
  __shared__ int sh_run[IDX_THREADS];      // sorted position where the run reaching INTO this thread's span starts
  const int tid = threadIdx.x;
  const long per = (n + IDX_THREADS - 1) / IDX_THREADS;
  const long lo = (long)tid * per, hi = lo + per < n ? lo + per : n;
  // pass 1: the start of the run that is open at the end of my span (or -1: my span holds no run head and is empty)
  long last_head = -1;
  for (long i = lo; i < hi; ++i)
    if (i == 0 || ks[i] != ks[i - 1]) last_head = i;
  sh_run[tid] = (int)last_head;
  __syncthreads();
  // run start of the element just before my span
  long open = 0;
  for (int t = tid - 1; t >= 0; --t)
    if (sh_run[t] >= 0) { open = sh_run[t]; break; }
  // pass 2: count my item heads
  int cnt = 0;
  {
    long rs = open;
    for (long i = lo; i < hi; ++i) {
      if (i == 0 || ks[i] != ks[i - 1]) rs = i;
      if (((i - rs) % CH) == 0) ++cnt;
    }
  }
  sh_cnt[tid] = cnt;
  __syncthreads();
  // exclusive scan of the counts (1024 entries: a serial scan by thread 0 is ~1 us)
  if (tid == 0) {
    int acc = 0;
    for (int t = 0; t < IDX_THREADS; ++t) { const int c = sh_cnt[t]; sh_cnt[t] = acc; acc += c; }
    counts[2] = acc;                         // all items
  }
  __syncthreads();
  int it = sh_cnt[tid];
  {
    long rs = open;
    for (long i = lo; i < hi; ++i) {
      const bool head = (i == 0 || ks[i] != ks[i - 1]);
      if (head) rs = i;
      if (((i - rs) % CH) == 0) {
        // length: up to CH sources, cut at the end of the run
        long e = i + 1;
        while (e < n && e - i < CH && ks[e] == ks[i]) ++e;
        const bool last = (e >= n) || ks[e] != ks[i];
        item_start[it] = (int)i;
        item_len[it] = (int)(e - i);
        item_flag[it] = (head ? 1 : 0) | (last ? 2 : 0);
        if (i == BT && BT > 0) counts[0] = it;           // first item of the negative list = number of session items
        ++it;
      }
    }
  }
  if (tid == 0 && (BT == 0 || BT >= n)) counts[0] = (BT == 0) ? 0 : counts[2];
}

struct RowArgs {
  const unsigned* ks; const unsigned* vs;
  const int* item_start; const int* item_len; const int* item_flag; const int* counts;
  int mode;                 // 0: session rows from `rows` [BT, ldh]; 1: negative rows coef[b] * attout[b, 0:ldh]
  long BT; int K, ldh; long ld_att;
  const float* rows; const float* coef; const float* attout;
  float* g_item; float* partial; float* norm_part;
};

// one wave per work item
template <int NCH>
__global__ __launch_bounds__(256) void segsum_rows_kernel(const RowArgs a) {
  const int lane = threadIdx.x & 63;
  const int n_sess = a.counts[0], n_all = a.counts[2];
  const int first = a.mode ? n_sess : 0, count = a.mode ? n_all - n_sess : n_sess;
  for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < count; w += gridDim.x * 4) {
    const int it = first + w;
    const int s0 = a.item_start[it], len = a.item_len[it], fl = a.item_flag[it];
    const unsigned row = a.ks[s0] & ((1u << 25) - 1);
    float4 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = zero4();
    float nrm = 0.f;
    for (int j = 0; j < len; ++j) {                      // sorted order = source order inside a run (stable sort)
      const long src = a.vs[s0 + j];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < a.ldh) {
          float4 v;
          if (a.mode == 0) v = ld4(a.rows + src * a.ldh + col);
          else { const long b = (src - a.BT) / a.K; v = scale4(ld4(a.attout + b * a.ld_att + col), a.coef[b]); }
          acc[c] = add4(acc[c], v);
          nrm += dot4(v, v);
        }
      }
    }
    if (a.mode == 0) {
      nrm = wave_sum(nrm);
      if (lane == 0) a.norm_part[it] = nrm;
    }
    float* dst = (fl == 3) ? a.g_item + (long)row * a.ldh : a.partial + (long)it * a.ldh;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < a.ldh) st4(dst + col, (fl == 3) ? add4(ld4(dst + col), acc[c]) : acc[c]);
    }
  }
}

// multi-chunk runs: the wave of the run's FIRST item folds the partial rows in item order; one more wave (the last of the
// grid) folds the norm pieces of the session list in item order into sqn[slot]
template <int NCH>
__global__ __launch_bounds__(256) void segsum_fold_kernel(const RowArgs a, float* __restrict__ sqn_slot) {
  const int lane = threadIdx.x & 63;
  const int n_sess = a.counts[0], n_all = a.counts[2];
  const int first = a.mode ? n_sess : 0, count = a.mode ? n_all - n_sess : n_sess;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  if (a.mode == 0 && sqn_slot && wave_g == nw - 1) {
    float s = 0.f;
    for (int i = 0; i < count; i += 64) {               // 64 items per trip, folded in item order
      const float v = (i + lane < count) ? a.norm_part[first + i + lane] : 0.f;
      float t = v;                                      // fixed-shape tree inside the trip, trips in order
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) t += __shfl_xor(t, o);
      s += t;
    }
    if (lane == 0) *sqn_slot += s;
  }
  for (int w = wave_g; w < count; w += nw) {
    const int it = first + w;
    const int fl = a.item_flag[it];
    if ((fl & 1) == 0 || fl == 3) continue;              // not the first chunk of a run, or a single-chunk run (already added)
    const unsigned row = a.ks[a.item_start[it]] & ((1u << 25) - 1);
    float4 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      acc[c] = col < a.ldh ? ld4(a.g_item + (long)row * a.ldh + col) : zero4();
    }
    for (int j = it;; ++j) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < a.ldh) acc[c] = add4(acc[c], ld4(a.partial + (long)j * a.ldh + col));
      }
      if (a.item_flag[j] & 2) break;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < a.ldh) st4(a.g_item + (long)row * a.ldh + col, acc[c]);
    }
  }
}

// deterministic sum of squares: per-block partials in block order
__global__ __launch_bounds__(256) void sqnorm_part_kernel(const float* __restrict__ g, long len, float* __restrict__ part) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long base = (long)blockIdx.x * 8192; base < len; base += (long)gridDim.x * 8192) {
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long e = base + (i * 256 + threadIdx.x) * 4;
      v[i] = (e < len) ? ld4(g + e) : zero4();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += dot4(v[i], v[i]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(64) void sqnorm_fold_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  const int lane = threadIdx.x;
  float s = 0.f;
  for (int i = 0; i < n; i += 64) {
    float t = (i + lane < n) ? part[i + lane] : 0.f;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) t += __shfl_xor(t, o);
    s += t;
  }
  if (lane == 0) *out += s;
}

}  // namespace

extern "C" int64_t tcar_segsum_ws_bytes(const tcar_dims_t* d, int64_t max_sources) {
  if (!d || max_sources <= 0) return 0;
  SegWs w;
  return (int64_t)carve(w, nullptr, max_sources, d->ldh, sort_tmp_bytes(max_sources)) + 4 * 512;
}

// sort the item-row sources of a batch and cut them into work items (depends on the feed only)
extern "C" int tcar_segsum_index(const tcar_dims_t* d, const tcar_batch_t* bt, void* ws, int64_t ws_bytes, void* stream) {
  if (!d || !bt || !ws || bt->B <= 0 || bt->T <= 0) return TCAR_E_ARG;
  const long BT = (long)bt->B * bt->T, BK = (bt->K > 0 && bt->neg) ? (long)bt->B * bt->K : 0, n = BT + BK;
  if (n >= (1L << 25) || d->n_items > (1 << 25)) return TCAR_E_ARG;
  SegWs w;
  const size_t sb = sort_tmp_bytes(n);
  if ((int64_t)carve(w, (char*)ws, n, d->ldh, sb) + 4 * 512 > ws_bytes) return TCAR_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  TCAR_LAUNCH(make_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, BT, BK, d->n_items, bt->seq, bt->neg, w.k_in, w.v_in);
  TCAR_CHECK_LAUNCH();
  size_t bytes = w.sort_bytes;
  if (rocprim::radix_sort_pairs(w.sort_tmp, bytes, (const unsigned*)w.k_in, w.k_out, (const unsigned*)w.v_in, w.v_out, (size_t)n, 0,
                                26, st) != hipSuccess)
    return TCAR_E_LAUNCH;
  TCAR_LAUNCH(build_items_kernel, dim3(1), dim3(IDX_THREADS), 0, st, n, BT, (const unsigned*)w.k_out, w.item_start, w.item_len,
              w.item_flag, w.counts);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// mode 0: g_item[row] += sum of rows [B*T, ldh] over the session sources of `row`, and *sqn_slot += sum ||rows[r]||^2;
// mode 1: g_item[row] += sum over the negative sources (b, k) of `row` of coef[b] * attout[b, 0:ldh]
extern "C" int tcar_segsum_apply(const tcar_dims_t* d, const tcar_batch_t* bt, void* ws, int mode, const float* rows,
                                 const float* coef, const float* attout, int64_t ld_att, float* g_item, float* sqn_slot,
                                 void* stream) {
  if (!d || !bt || !ws || !g_item || (mode == 0 && !rows) || (mode == 1 && (!coef || !attout))) return TCAR_E_ARG;
  const long BT = (long)bt->B * bt->T, BK = (bt->K > 0 && bt->neg) ? (long)bt->B * bt->K : 0, n = BT + BK;
  if (mode == 1 && BK == 0) return TCAR_OK;
  SegWs w;
  carve(w, (char*)ws, n, d->ldh, sort_tmp_bytes(n));
  RowArgs a{};
  a.ks = w.k_out; a.vs = w.v_out; a.item_start = w.item_start; a.item_len = w.item_len; a.item_flag = w.item_flag; a.counts = w.counts;
  a.mode = mode; a.BT = BT; a.K = bt->K > 0 ? bt->K : 1; a.ldh = d->ldh; a.ld_att = ld_att;
  a.rows = rows; a.coef = coef; a.attout = attout; a.g_item = g_item; a.partial = w.partial; a.norm_part = w.norm_part;
  const long src = mode ? BK : BT;
  int grid = (int)((src + 3) / 4);                      // items <= sources
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  hipStream_t st = (hipStream_t)stream;
  if (d->ldh <= 256) {
    TCAR_LAUNCH(segsum_rows_kernel<1>, dim3(grid), dim3(256), 0, st, a);
    TCAR_CHECK_LAUNCH();
    TCAR_LAUNCH(segsum_fold_kernel<1>, dim3(grid), dim3(256), 0, st, a, sqn_slot);
  } else {
    TCAR_LAUNCH(segsum_rows_kernel<2>, dim3(grid), dim3(256), 0, st, a);
    TCAR_CHECK_LAUNCH();
    TCAR_LAUNCH(segsum_fold_kernel<2>, dim3(grid), dim3(256), 0, st, a, sqn_slot);
  }
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// where the gather backward writes the session sources' item-row gradients ([B*T, ldh], tcar_grads_t.rows_out) for mode 0
extern "C" float* tcar_segsum_rows_buffer(const tcar_dims_t* d, const tcar_batch_t* bt, void* ws) {
  if (!d || !bt || !ws) return nullptr;
  const long BT = (long)bt->B * bt->T, BK = (bt->K > 0 && bt->neg) ? (long)bt->B * bt->K : 0, n = BT + BK;
  SegWs w;
  carve(w, (char*)ws, n, d->ldh, sort_tmp_bytes(n));
  return w.rows;
}

__global__ __launch_bounds__(256) void loss_combine_kernel(int B, const float* __restrict__ ce, const float* __restrict__ fb,
                                                           float weight, float* __restrict__ loss) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b < B) loss[b] = ce[b] + weight * fb[b];
}
// loss[b] = ce[b] + weight * neg_fb[b] (model_combine.py:147)
extern "C" int tcar_loss_combine(int B, const float* ce, const float* neg_fb, float weight, float* loss, void* stream) {
  if (B <= 0) return TCAR_OK;
  if (!ce || !neg_fb || !loss) return TCAR_E_ARG;
  TCAR_LAUNCH(loss_combine_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, ce, neg_fb, weight, loss);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// *out += sum g[0:len]^2 in a fixed order (512 block partials folded by one wave); ws: >= 512 floats
extern "C" int tcar_sqnorm_det(const float* g, int64_t len, float* out, float* ws, void* stream) {
  if (len <= 0) return TCAR_OK;
  if (!g || !out || !ws || (len & 3) || !tcar_aligned16(g)) return TCAR_E_ARG;
  long blocks = (len + 8191) / 8192;
  if (blocks > 512) blocks = 512;
  TCAR_LAUNCH(sqnorm_part_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, (long)len, ws);
  TCAR_CHECK_LAUNCH();
  TCAR_LAUNCH(sqnorm_fold_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)ws, (int)blocks, out);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
