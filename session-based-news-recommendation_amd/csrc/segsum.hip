// Deterministic sparse backward of the item table: sort by row + segmented wavefront reduction ("hard parts").
//
// The IndexedSlices gradient of the item lookups (model_combine.py:54,142,156) adds one row per gathered session click
// (B*T rows, values through the clip Jacobian) and one row per sampled negative (B*K rows, value coef[b] * attout[b]) into
// the dense gradient of the candidate block.  News batches are head heavy — one article can own 10 % of the rows of a
// batch — and the plain form (float atomics, embed.hip / score.hip) makes the sum depend on arrival order: replicas and
// repeated runs differ in the last bit.  Here the order is fixed:
//   1. index (depends on the feed only; the step driver runs it on the aux stream under the forward pass): the sources of
//      each list — session clicks, negatives — sorted by destination row with a STABLE radix sort: lists of up to 16384
//      sources in ONE workgroup each, keys and counters resident in LDS; longer lists through the same scheme spread over
//      workgroups (count / offsets / scatter kernels per 4-bit pass) — no library call anywhere on the step's path;
//   2. rows: every wave owns 4 sorted positions (a workgroup: 64) and sums the runs that START there, in sorted order =
//      source order, then adds the sum into the dense gradient — ONE writer per destination row, no atomics, no partial
//      rows in memory.  A run of more than 64 sources (a popular article) is summed by the 16 waves of its workgroup
//      together: fixed 1/16 slices, folded through LDS in wave order;
//   3. the norm pieces of tf.clip_by_norm — one per source row, written by the gather backward — and the
//      block partials of the dense norm are folded in index order by workgroup 0 of the session-list pass.
// Bit-for-bit repeatable whatever the dispatch order.
#include <cstring>
#include "tcar_common.h"

namespace {

constexpr int SORT_T = 1024, SORT_E = 16, SORT_MAX = SORT_T * SORT_E;     // LDS sort: elements per thread, list limit
constexpr int SORT_LDS = SORT_MAX * 4 + SORT_MAX * 2 + 16 * SORT_T * 2 + 64;
constexpr int DENSE_PARTS = 512;       // block partials of tcar_sqnorm_det
constexpr int TAIL_BYTES = 4096;       // scratch at the end of the workspace: DENSE_PARTS floats

struct SegWs {                         // carved out of the caller's workspace (all device pointers)
  unsigned* ks; unsigned* vs;          // sorted destination rows / source indices: session list [0, BT), negatives [BT, BT + BK)
  unsigned* k_in; unsigned* v_in;      // ping-pong partner of (ks, vs) in the multi-workgroup sort of long lists
  float* src_norm;                     // [B*T] squared norm of every session source row (written by the gather backward)
  float* rows;                         // [B*T, ldh] the session sources' gradient rows (written by the gather backward)
  void* sort_tmp; size_t sort_bytes;
};

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

size_t carve(SegWs& w, char* base, long n, int ldh, size_t sort_bytes) {
  size_t o = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += align_up(bytes); return p; };
  w.ks = (unsigned*)take(4 * n); w.vs = (unsigned*)take(4 * n); w.k_in = (unsigned*)take(4 * n); w.v_in = (unsigned*)take(4 * n);
  w.src_norm = (float*)take(4 * n);
  w.rows = (float*)take((size_t)4 * n * ldh);
  w.sort_tmp = take(sort_bytes); w.sort_bytes = sort_bytes;
  return o;
}

// multi-workgroup sort of lists longer than SORT_MAX: 4096 sources per workgroup, 16 digit totals per workgroup and pass
constexpr int RS_T = 256, RS_E = 16, RS_CHUNK = RS_T * RS_E;
size_t sort_tmp_bytes(long n) {
  if (n <= SORT_MAX) return 0;
  const long G = (n + RS_CHUNK - 1) / RS_CHUNK;
  return (size_t)16 * G * sizeof(unsigned) + 256;
}

__device__ __forceinline__ unsigned key_of(int list, int id, int n_items) {
  return list == 0 ? (unsigned)(clampi(id, 1, n_items) - 1) : (unsigned)clampi(id, 0, n_items - 1);
}

struct SortArgs {
  const int32_t* ids[2]; long n[2]; long lo[2];          // the two lists: ids, length, first sorted position
  int n_items, npass;
  unsigned* ks; unsigned* vs;
};

// One workgroup per list: LSD radix sort, 4 bits per pass, everything in LDS.  Thread t owns the 16 consecutive slots
// [16t, 16t+16) of the current order (so equal digits keep their order: stable); the counters are digit-major, so one
// exclusive scan over all 16 * 1024 of them yields every thread's write position for every digit.  Unused slots hold key
// 0xffffffff: digit 15 in every pass, they stay behind the sources.
__global__ __launch_bounds__(SORT_T) void lds_sort_kernel(const SortArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* keys = (unsigned*)smem;
  unsigned short* vals = (unsigned short*)(smem + SORT_MAX * 4);
  unsigned short* cnt = (unsigned short*)(smem + SORT_MAX * 6);
  unsigned* wsum = (unsigned*)(smem + SORT_MAX * 6 + 16 * SORT_T * 2);
  const int list = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long n = a.n[list];
  if (n <= 0 || n > SORT_MAX) return;
  const int32_t* ids = a.ids[list];
  unsigned k[SORT_E];
  unsigned short v[SORT_E];
#pragma unroll
  for (int i = 0; i < SORT_E; ++i) {
    const int e = tid * SORT_E + i;
    k[i] = e < n ? key_of(list, ids[e], a.n_items) : 0xffffffffu;
    v[i] = (unsigned short)e;
  }
  for (int pass = 0; pass < a.npass; ++pass) {
    const int sh = pass * 4;
#pragma unroll
    for (int d = 0; d < 16; ++d) cnt[d * SORT_T + tid] = 0;
#pragma unroll
    for (int i = 0; i < SORT_E; ++i) cnt[((k[i] >> sh) & 15) * SORT_T + tid] += 1;
    __syncthreads();
    {  // exclusive scan of the 16384 counters in place
      unsigned short c[16];
      unsigned tot = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) { c[j] = cnt[tid * 16 + j]; tot += c[j]; }
      unsigned inc = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
      }
      if (lane == 63) wsum[wv] = inc;
      __syncthreads();
      unsigned base = inc - tot;
      for (int w2 = 0; w2 < wv; ++w2) base += wsum[w2];
#pragma unroll
      for (int j = 0; j < 16; ++j) { cnt[tid * 16 + j] = (unsigned short)base; base += c[j]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SORT_E; ++i) {
      const int ci = ((k[i] >> sh) & 15) * SORT_T + tid;
      const unsigned pos = cnt[ci];
      cnt[ci] = (unsigned short)(pos + 1);
      keys[pos] = k[i];
      vals[pos] = v[i];
    }
    __syncthreads();
    if (pass + 1 < a.npass) {
#pragma unroll
      for (int i = 0; i < SORT_E; ++i) { k[i] = keys[tid * SORT_E + i]; v[i] = vals[tid * SORT_E + i]; }
      __syncthreads();
    }
  }
  for (long e = tid; e < n; e += SORT_T) {
    a.ks[a.lo[list] + e] = keys[e];
    a.vs[a.lo[list] + e] = vals[e];
  }
}

// ---- lists longer than SORT_MAX: the same stable LSD radix sort (4 bits per pass), spread over workgroups ----------------
// Workgroup b owns the 4096 consecutive positions [4096 b, 4096 b + 4096) of the pass's input order and thread t the 16
// consecutive ones [16 t, 16 t + 16) of those, so equal digits keep their order everywhere.  Per pass:
//   count    digit totals of every workgroup                       -> tot[digit][workgroup]
//   offsets  ONE workgroup: exclusive scan of tot in digit-major order (= first output position of every (digit, workgroup))
//   scatter  every workgroup re-counts per thread, scans (digit-major, as in lds_sort_kernel) and writes its elements in order.
// Pass 0 reads the feed itself (key = destination row, value = source index); the buffers (k_in, v_in) / (ks, vs) alternate so
// that the last pass lands in (ks, vs).
struct RadixArgs {
  const int32_t* ids; int list, n_items, first;
  const unsigned* kin; const unsigned* vin; unsigned* kout; unsigned* vout;
  long n; int shift, G; unsigned* tot;
};
__device__ __forceinline__ void radix_elem(const RadixArgs& a, long e, unsigned& k, unsigned& v) {
  if (a.first) { k = key_of(a.list, a.ids[e], a.n_items); v = (unsigned)e; }
  else { k = a.kin[e]; v = a.vin[e]; }
}
__global__ __launch_bounds__(RS_T) void radix_count_kernel(const RadixArgs a) {
  __shared__ unsigned cnt[16 * RS_T];
  const int tid = threadIdx.x;
  const long e0 = (long)blockIdx.x * RS_CHUNK + (long)tid * RS_E;
  unsigned c[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) c[d] = 0;
  for (int i = 0; i < RS_E; ++i) {
    const long e = e0 + i;
    if (e < a.n) {
      unsigned k, v;
      radix_elem(a, e, k, v);
      const unsigned dg = (k >> a.shift) & 15u;
#pragma unroll
      for (int d = 0; d < 16; ++d) c[d] += (dg == (unsigned)d) ? 1u : 0u;
    }
  }
#pragma unroll
  for (int d = 0; d < 16; ++d) cnt[d * RS_T + tid] = c[d];
  __syncthreads();
  if (tid < 16) {
    unsigned t = 0;
    for (int j = 0; j < RS_T; ++j) t += cnt[tid * RS_T + j];
    a.tot[(long)tid * a.G + blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(1024) void radix_offsets_kernel(unsigned* __restrict__ tot, long m) {
  __shared__ unsigned wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long per = (m + 1023) / 1024, lo = (long)tid * per < m ? (long)tid * per : m, hi = lo + per < m ? lo + per : m;
  unsigned s = 0;
  for (long i = lo; i < hi; ++i) s += tot[i];
  unsigned inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned up = __shfl_up(inc, o);
    if (lane >= o) inc += up;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  unsigned base = inc - s;
  for (int w2 = 0; w2 < wv; ++w2) base += wsum[w2];
  for (long i = lo; i < hi; ++i) { const unsigned t = tot[i]; tot[i] = base; base += t; }
}
__global__ __launch_bounds__(RS_T) void radix_scatter_kernel(const RadixArgs a) {
  __shared__ unsigned cnt[16 * RS_T];
  __shared__ unsigned wsum[RS_T / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long e0 = (long)blockIdx.x * RS_CHUNK + (long)tid * RS_E;
  unsigned k[RS_E], v[RS_E];
  unsigned c[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) c[d] = 0;
#pragma unroll
  for (int i = 0; i < RS_E; ++i) {
    k[i] = 0xffffffffu; v[i] = 0;
    if (e0 + i < a.n) {
      radix_elem(a, e0 + i, k[i], v[i]);
      const unsigned dg = (k[i] >> a.shift) & 15u;
#pragma unroll
      for (int d = 0; d < 16; ++d) c[d] += (dg == (unsigned)d) ? 1u : 0u;
    }
  }
#pragma unroll
  for (int d = 0; d < 16; ++d) cnt[d * RS_T + tid] = c[d];
  __syncthreads();
  {  // exclusive scan of the 16 * 256 counters in digit-major order, in place (thread t owns entries [16 t, 16 t + 16))
    unsigned cc[16], tot = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { cc[j] = cnt[tid * 16 + j]; tot += cc[j]; }
    unsigned inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned up = __shfl_up(inc, o);
      if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned base = inc - tot;
    for (int w2 = 0; w2 < wv; ++w2) base += wsum[w2];
#pragma unroll
    for (int j = 0; j < 16; ++j) { cnt[tid * 16 + j] = base; base += cc[j]; }
  }
  __syncthreads();
  // position of this thread's first element of digit d: global base of (d, workgroup) + elements of digit d in earlier threads
  unsigned off[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) off[d] = a.tot[(long)d * a.G + blockIdx.x] + cnt[d * RS_T + tid] - cnt[d * RS_T];
#pragma unroll
  for (int i = 0; i < RS_E; ++i) {
    if (e0 + i < a.n) {
      const unsigned dg = (k[i] >> a.shift) & 15u;
      unsigned pos = 0;
#pragma unroll
      for (int d = 0; d < 16; ++d)
        if (dg == (unsigned)d) { pos = off[d]; off[d] += 1; }
      a.kout[pos] = k[i];
      a.vout[pos] = v[i];
    }
  }
}

struct RowArgs {
  const unsigned* ks; const unsigned* vs;
  long lo, hi;              // sorted positions of this list
  int mode;                 // 0: session rows from `rows` [BT, ldh]; 1: negative rows coef[b] * attout[b, 0:ldh], b = source / K
  int K, ldh; long ld_att;
  const float* rows; const float* coef; const float* attout;
  float* g_item;
  const float* src_norm; long n_norm; float* sqn_slot;        // mode 0: *sqn_slot += sum src_norm[0:n_norm] in index order
  const float* dense_part; int n_dense; float* dense_slot;     // mode 0: *dense_slot += sum dense_part[0:n_dense] in index order
  int B; const float* ce; const float* fb; float weight; float* loss;      // mode 1 (optional): loss[b] = ce[b] + weight * fb[b]
};

template <int NCH>
__device__ __forceinline__ void load_source(const RowArgs& a, unsigned src, int lane, float4 (&v)[NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
    v[c] = zero4();
    if (col < a.ldh) {
      if (a.mode == 0) v[c] = ld4(a.rows + (long)src * a.ldh + col);
      else { const long b = src / (unsigned)a.K; v[c] = scale4(ld4(a.attout + b * a.ld_att + col), a.coef[b]); }
    }
  }
}

// acc += sources [s0, s1) of the sorted list, in order, 4 row loads in flight
template <int NCH>
__device__ __forceinline__ void sum_sources(const RowArgs& a, long s0, long s1, int lane, float4 (&acc)[NCH]) {
  for (long c0 = s0; c0 < s1; c0 += 64) {
    const int m = (int)(s1 - c0 < 64 ? s1 - c0 : 64);
    const unsigned my = lane < m ? a.vs[c0 + lane] : 0u;
    int t = 0;
    for (; t + 4 <= m; t += 4) {
      float4 v[4][NCH];
#pragma unroll
      for (int u = 0; u < 4; ++u) load_source<NCH>(a, (unsigned)__builtin_amdgcn_readlane((int)my, t + u), lane, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = add4(acc[c], v[u][c]);
    }
    for (; t < m; ++t) {
      float4 v[NCH];
      load_source<NCH>(a, (unsigned)__builtin_amdgcn_readlane((int)my, t), lane, v);
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = add4(acc[c], v[c]);
    }
  }
}

// sum of v[0:n] in a fixed order by one workgroup of 1024 threads: strided per-thread sums, xor tree, waves in order
__device__ __forceinline__ float block_fold(const float* __restrict__ v, long n, float* sh16) {
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) s += v[i];
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh16[threadIdx.x >> 6] = s;
  __syncthreads();
  float tot = 0.f;
  for (int w = 0; w < 16; ++w) tot += sh16[w];
  return tot;
}

template <int NCH>
__global__ __launch_bounds__(1024) void segsum_rows_kernel(const RowArgs a) {
  __shared__ __attribute__((aligned(16))) float part[16][NCH * 256];
  __shared__ float sh16[16];
  __shared__ long long_p;
  __shared__ int long_len;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (long base = a.lo + (long)blockIdx.x * 64; base < a.hi; base += (long)gridDim.x * 64) {
    if (tid == 0) long_len = 0;
    __syncthreads();
    const long p0 = base + wv * 4;
    if (p0 < a.hi) {
      // lane j looks at sorted position p0 - 1 + j: the predecessor, my 4 positions, 59 of look-ahead
      const long q = p0 - 1 + lane;
      const unsigned kq = q < a.lo ? 0xfffffffeu : (q < a.hi ? a.ks[q] : 0xffffffffu);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long p = p0 + i;
        if (p >= a.hi) break;
        const unsigned kp = (unsigned)__builtin_amdgcn_readlane((int)kq, i + 1);
        if (kp == (unsigned)__builtin_amdgcn_readlane((int)kq, i)) continue;      // the run started earlier: its owner sums it
        // length of the run: equal keys from p on
        const unsigned long long same = __ballot(kq == kp) >> (i + 1);
        const int len = __builtin_ctzll(~same);                                      // <= 63 - i (zeros were shifted in)
        long e = p + len;
        if (len == 63 - i) {                                                         // reaches the end of the window: look on
          for (;;) {
            const long qq = e + lane;
            const unsigned k2 = qq < a.hi ? a.ks[qq] : 0xffffffffu;
            const unsigned long long ne = ~__ballot(k2 == kp);
            if (ne == 0ull) { e += 64; continue; }
            e += __builtin_ctzll(ne);
            break;
          }
        }
        if (e - p > 64) {                      // at most one per workgroup: it covers the rest of the slab
          if (lane == 0) { long_p = p; long_len = (int)(e - p); }
          continue;
        }
        float4 gi[NCH], acc[NCH];
        float* dst = a.g_item + (long)kp * a.ldh;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int col = c * 256 + lane * 4;
          gi[c] = col < a.ldh ? ld4(dst + col) : zero4();
          acc[c] = zero4();
        }
        sum_sources<NCH>(a, p, e, lane, acc);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int col = c * 256 + lane * 4;
          if (col < a.ldh) st4(dst + col, add4(gi[c], acc[c]));
        }
      }
    }
    __syncthreads();
    if (long_len > 0) {                        // the 16 waves take fixed sixteenths of the run; folded in wave order
      const long P = long_p;
      const int L = long_len, per = (L + 15) / 16;
      const long s0 = P + (long)wv * per, s1 = (P + L < s0 + per) ? P + L : s0 + per;
      float4 acc[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = zero4();
      if (s0 < s1) sum_sources<NCH>(a, s0, s1, lane, acc);
#pragma unroll
      for (int c = 0; c < NCH; ++c) st4(&part[wv][c * 256 + lane * 4], acc[c]);
      __syncthreads();
      const unsigned row = a.ks[P];
      for (int col = tid; col < a.ldh; col += 1024) {
        float s = 0.f;
        for (int w = 0; w < 16; ++w) s += part[w][col];
        a.g_item[(long)row * a.ldh + col] += s;
      }
    }
    __syncthreads();
  }
  if (a.mode == 1 && a.loss)
    for (int b = blockIdx.x * 1024 + tid; b < a.B; b += gridDim.x * 1024) a.loss[b] = a.ce[b] + a.weight * a.fb[b];
  if (a.mode == 0 && blockIdx.x == 0) {
    if (a.sqn_slot && a.src_norm) {
      const float s = block_fold(a.src_norm, a.n_norm, sh16);
      if (tid == 0) *a.sqn_slot += s;
    }
    if (a.dense_slot && a.dense_part) {
      const float s = block_fold(a.dense_part, a.n_dense, sh16);
      if (tid == 0) *a.dense_slot += s;
    }
  }
}

// deterministic sum of squares: block partials (folded in block order by the session-list pass of tcar_segsum_apply)
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

int list_lengths(const tcar_batch_t* bt, long& BT, long& BK) {
  if (!bt || bt->B <= 0 || bt->T <= 0) return TCAR_E_ARG;
  BT = (long)bt->B * bt->T;
  BK = (bt->K > 0 && bt->neg) ? (long)bt->B * bt->K : 0;
  return TCAR_OK;
}

}  // namespace

extern "C" int64_t tcar_segsum_ws_bytes(const tcar_dims_t* d, int64_t max_sources) {
  if (!d || max_sources <= 0) return 0;
  SegWs w;
  return (int64_t)carve(w, nullptr, max_sources, d->ldh, sort_tmp_bytes(max_sources)) + TAIL_BYTES;
}

// sort the item-row sources of a batch by destination row (depends on the feed only)
extern "C" int tcar_segsum_index(const tcar_dims_t* d, const tcar_batch_t* bt, void* ws, int64_t ws_bytes, void* stream) {
  long BT, BK;
  if (!d || !ws || list_lengths(bt, BT, BK) != TCAR_OK) return TCAR_E_ARG;
  const long n = BT + BK;
  if (n >= (1L << 25) || d->n_items > (1 << 25) || d->n_items < 1) return TCAR_E_ARG;
  SegWs w;
  if ((int64_t)carve(w, (char*)ws, n, d->ldh, sort_tmp_bytes(n)) + TAIL_BYTES > ws_bytes) return TCAR_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  int bits = 1;
  while ((1L << bits) < d->n_items) ++bits;
  SortArgs a{};
  a.ids[0] = bt->seq; a.n[0] = BT; a.lo[0] = 0;
  a.ids[1] = bt->neg; a.n[1] = BK; a.lo[1] = BT;
  a.n_items = d->n_items; a.npass = (bits + 3) / 4; a.ks = w.ks; a.vs = w.vs;
  if (BT <= SORT_MAX || (BK > 0 && BK <= SORT_MAX)) {
    TCAR_SET_LDS_ONCE(lds_sort_kernel, SORT_LDS);
    TCAR_LAUNCH(lds_sort_kernel, dim3(BK > 0 ? 2 : 1), dim3(SORT_T), SORT_LDS, st, a);
    TCAR_CHECK_LAUNCH();
  }
  for (int list = 0; list < 2; ++list) {
    const long nl = a.n[list];
    if (nl <= SORT_MAX) continue;
    if (!w.sort_tmp || sort_tmp_bytes(nl) > w.sort_bytes) return TCAR_E_ARG;    // carved for a sort of n >= nl elements
    RadixArgs r{};
    r.ids = a.ids[list]; r.list = list; r.n_items = d->n_items; r.n = nl;
    r.G = (int)((nl + RS_CHUNK - 1) / RS_CHUNK);
    r.tot = (unsigned*)w.sort_tmp;
    unsigned* bufk[2] = {w.k_in + a.lo[list], w.ks + a.lo[list]};
    unsigned* bufv[2] = {w.v_in + a.lo[list], w.vs + a.lo[list]};
    for (int pass = 0; pass < a.npass; ++pass) {
      const int dst = ((a.npass - 1 - pass) & 1) ? 0 : 1;          // the last pass writes (ks, vs)
      r.first = pass == 0; r.shift = 4 * pass;
      r.kin = bufk[1 - dst]; r.vin = bufv[1 - dst]; r.kout = bufk[dst]; r.vout = bufv[dst];
      TCAR_LAUNCH(radix_count_kernel, dim3((unsigned)r.G), dim3(RS_T), 0, st, r);
      TCAR_LAUNCH(radix_offsets_kernel, dim3(1), dim3(1024), 0, st, r.tot, (long)16 * r.G);
      TCAR_LAUNCH(radix_scatter_kernel, dim3((unsigned)r.G), dim3(RS_T), 0, st, r);
      TCAR_CHECK_LAUNCH();
    }
  }
  return TCAR_OK;
}

// mode 0: g_item[row] += sum of rows [B*T, ldh] over the session sources of `row`; with sqn_slot: *sqn_slot += the sum of the
//         per-source norms the gather backward left in tcar_segsum_norms_buffer; with dense_slot: *dense_slot += the fold of
//         tcar_sqnorm_det's partials (both in index order)
// mode 1: g_item[row] += sum over the negative sources (b, k) of `row` of coef[b] * attout[b, 0:ldh]; with `loss` also
//         loss[b] = ce[b] + weight * neg_fb[b] (model_combine.py:147)
extern "C" int tcar_segsum_apply(const tcar_dims_t* d, const tcar_batch_t* bt, void* ws, int64_t ws_bytes, int mode,
                                 const float* rows, const float* coef, const float* attout, int64_t ld_att, float* g_item,
                                 float* sqn_slot, float* dense_slot, const float* ce, const float* neg_fb, float weight,
                                 float* loss, void* stream) {
  long BT, BK;
  if (!d || !ws || !g_item || list_lengths(bt, BT, BK) != TCAR_OK) return TCAR_E_ARG;
  if ((mode == 0 && !rows) || (mode == 1 && (!coef || !attout || BK == 0)) || (loss && (!ce || !neg_fb))) return TCAR_E_ARG;
  if (d->ldh > 512 || (d->ldh & 3)) return TCAR_E_ARG;  // two 256-column chunks per wave
  const long n = BT + BK;
  SegWs w;
  if ((int64_t)carve(w, (char*)ws, n, d->ldh, sort_tmp_bytes(n)) + TAIL_BYTES > ws_bytes) return TCAR_E_ARG;
  RowArgs a{};
  a.ks = w.ks; a.vs = w.vs;
  a.lo = mode ? BT : 0; a.hi = mode ? n : BT;
  a.mode = mode; a.K = bt->K > 0 ? bt->K : 1; a.ldh = d->ldh; a.ld_att = ld_att;
  a.rows = rows; a.coef = coef; a.attout = attout; a.g_item = g_item;
  a.src_norm = w.src_norm; a.n_norm = BT; a.sqn_slot = mode == 0 ? sqn_slot : nullptr;
  a.dense_part = (const float*)((char*)ws + ws_bytes - TAIL_BYTES); a.n_dense = DENSE_PARTS; a.dense_slot = mode == 0 ? dense_slot : nullptr;
  a.B = bt->B; a.ce = ce; a.fb = neg_fb; a.weight = weight; a.loss = loss;
  long grid = (a.hi - a.lo + 63) / 64;
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  hipStream_t st = (hipStream_t)stream;
  if (d->ldh <= 256) TCAR_LAUNCH(segsum_rows_kernel<1>, dim3((unsigned)grid), dim3(1024), 0, st, a);
  else TCAR_LAUNCH(segsum_rows_kernel<2>, dim3((unsigned)grid), dim3(1024), 0, st, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// where the gather backward writes the session sources' item-row gradients ([B*T, ldh], tcar_grads_t.rows_out) and their
// squared norms ([B*T], tcar_grads_t.norms_out) for mode 0
extern "C" float* tcar_segsum_rows_buffer(const tcar_dims_t* d, const tcar_batch_t* bt, void* ws) {
  long BT, BK;
  if (!d || !ws || list_lengths(bt, BT, BK) != TCAR_OK) return nullptr;
  SegWs w;
  carve(w, (char*)ws, BT + BK, d->ldh, sort_tmp_bytes(BT + BK));
  return w.rows;
}
extern "C" float* tcar_segsum_norms_buffer(const tcar_dims_t* d, const tcar_batch_t* bt, void* ws) {
  long BT, BK;
  if (!d || !ws || list_lengths(bt, BT, BK) != TCAR_OK) return nullptr;
  SegWs w;
  carve(w, (char*)ws, BT + BK, d->ldh, sort_tmp_bytes(BT + BK));
  return w.src_norm;
}

// block partials of sum g[0:len]^2 into the LAST 4096 bytes of the segsum workspace (512 floats, unused ones zero); the
// session-list pass of tcar_segsum_apply folds them in block order into its dense_slot
extern "C" int tcar_sqnorm_det(const float* g, int64_t len, void* ws, int64_t ws_bytes, void* stream) {
  if (!g || !ws || ws_bytes < TAIL_BYTES || len < 0 || (len & 3) || !tcar_aligned16(g)) return TCAR_E_ARG;
  float* part = (float*)((char*)ws + ws_bytes - TAIL_BYTES);
  long blocks = (len + 8191) / 8192;
  if (blocks > DENSE_PARTS) blocks = DENSE_PARTS;
  if (blocks < DENSE_PARTS && hipMemsetAsync(part, 0, DENSE_PARTS * sizeof(float), (hipStream_t)stream) != hipSuccess) return TCAR_E_LAUNCH;
  if (blocks > 0) {
    TCAR_LAUNCH(sqnorm_part_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, (long)len, part);
    TCAR_CHECK_LAUNCH();
  }
  return TCAR_OK;
}
