// Per-variable gradient clipping + TF-1 Adam for gfx950 (HBM-bound: 16 B read + 12 B written per parameter).
//
// Reference: model_combine.py:155-163 — tf.train.AdamOptimizer(lr) (beta1 .9, beta2 .999, eps 1e-8),
// per-variable tf.clip_by_norm(grad, max_grad), apply_gradients.  TF-1 Adam:
//   lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)            (computed by the caller)
//   m = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2 ;  w -= lr_t * m / (sqrt(v) + eps)
// Its sparse apply de-duplicates indices and still decays / updates every row, so it equals this dense update
// on the summed gradient.  The clip norm of a variable is sqrt(use_dense*sqn_dense + sqn_pieces)
// where the pieces are the IndexedSlices value blocks accumulated by the embedding backward kernels (S5).
#include "tcar_common.h"
#include <stdlib.h>
#include "tcar_bf16_layout.h"

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

namespace {

struct SegArgs {
  tcar_segments_t s;
};

__device__ __forceinline__ float clip_factor(const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense,
                                             int slot, float clip) {
  if (clip <= 0.f) return 1.f;
  const float n2 = (use_dense[slot] ? sqn_dense[slot] : 0.f) + sqn_pieces[slot];
  const float n = sqrtf(n2);
  return clip / fmaxf(n, clip);
}

// grid = (chunks <= 512, nseg); each workgroup strides over 8192-float pieces of one segment and issues ONE
// atomic (same-address float atomics serialise at ~12 ns each: thousands of them cost more than the read)
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, const SegArgs a, float* __restrict__ out) {
  __shared__ float sh[4];
  const int seg = blockIdx.y;
  const long off = a.s.off[seg];
  const long len = a.s.len[seg];
  if ((long)blockIdx.x * 8192 >= len) return;
  float s = 0.f;
  for (long base = (long)blockIdx.x * 8192; base < len; base += (long)gridDim.x * 8192) {
    float4 v[8];                      // 8 loads in flight per thread: the pass is latency bound otherwise
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long e = base + (i * 256 + threadIdx.x) * 4;
      v[i] = (e < len) ? ld4(g + off + e) : zero4();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += dot4(v[i], v[i]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out + a.s.slot[seg], sh[0] + sh[1] + sh[2] + sh[3]);
}

// Small segments (the dense weights: <= 262,144 floats each): ONE workgroup of 1024 threads per segment, fixed summation
// order (thread-strided partial sums, xor tree inside the wave, the 16 waves in order).  The result depends on the data
// only, so ranks that hold identical gradients compute identical norms — no broadcast needed to keep them in step.
__global__ __launch_bounds__(1024) void sqnorm_seg_kernel(const float* __restrict__ g, const SegArgs a, float* __restrict__ out,
                                                          const TcarSignal sig) {
  __shared__ float sh[16];
  const int seg = blockIdx.x;
  const long off = a.s.off[seg];
  const long len = a.s.len[seg];
  float s = 0.f;
  for (long base = 0; base < len; base += 1024 * 4 * 8) {
    float4 v[8];                      // 8 loads in flight per thread: the largest weight (1 MB) is 8 dependent trips, not 16
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long e = base + (i * 1024 + threadIdx.x) * 4;
      v[i] = (e < len) ? ld4(g + off + e) : zero4();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += dot4(v[i], v[i]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += sh[w];
    atomicAdd(out + a.s.slot[seg], t);                  // one add per segment
  }
  tcar_signal_done(sig);        // (the fused step joins the third stream into the main one behind this launch)
}

// The same norms with SEVERAL workgroups per segment (round 4): one workgroup reads its 1-MB weight at one CU's load rate
// (12.5 us in the step, the last launch of the third stream's chain).  Here every 32,768-float chunk is a workgroup (35 for the
// reference's shapes): one trip of 8 loads per thread, chunk partial to scratch; the LAST workgroup to arrive folds each segment's
// partials in CHUNK order — fixed order, bit-for-bit repeatable, identical on ranks with identical gradients — and adds once per
// segment.  scratch: [0] arrival counter (zero between launches), [1 ..] chunk partials.
constexpr int SQ_CHUNK = 32768;
struct Seg2Args {
  tcar_segments_t s;
  int first[TCAR_NSLOT + 1];          // first chunk of every segment (prefix sums)
};
__global__ __launch_bounds__(1024) void sqnorm_seg2_kernel(const float* __restrict__ g, const Seg2Args a, float* __restrict__ out,
                                                           unsigned* __restrict__ scratch, const TcarSignal sig) {
  __shared__ float sh[16];
  __shared__ int last;
  const int nchunks = gridDim.x;
  int seg = 0;
  while (seg + 1 < a.s.nseg && (int)blockIdx.x >= a.first[seg + 1]) ++seg;
  const long c0 = (long)((int)blockIdx.x - a.first[seg]) * SQ_CHUNK;
  const long len = a.s.len[seg] - c0 < SQ_CHUNK ? a.s.len[seg] - c0 : (long)SQ_CHUNK;
  const float* p = g + a.s.off[seg] + c0;
  float4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long e = (long)(i * 1024 + (int)threadIdx.x) * 4;
    v[i] = (e < len) ? ld4(p + e) : zero4();
  }
  float sm = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) sm += dot4(v[i], v[i]);
  sm = wave_sum(sm);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sm;
  __syncthreads();
  float* part = reinterpret_cast<float*>(scratch + 1);
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += sh[w];
    __hip_atomic_store(part + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // release the partial, acquire everybody else's when this is the last arrival
    const unsigned old = __hip_atomic_fetch_add(scratch, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = (old == (unsigned)nchunks - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (last && threadIdx.x < 64) {
    const int l = threadIdx.x;
    if (l < a.s.nseg) {
      float t = 0.f;
      for (int c = a.first[l]; c < a.first[l + 1]; ++c) t += __hip_atomic_load(part + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicAdd(out + a.s.slot[l], t);                 // one add per segment
    }
    if (l == 0) __hip_atomic_store(scratch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // reusable by the next launch
  }
  tcar_signal_done(sig);
}

// ---- split-K slabs folded in split order: dst[e] += sum_k slabs[k * stride + e] ---------------------------------------
struct FoldArgs {
  int nseg;
  float* dst[9]; const float* slabs[9]; long n[9]; int ks[9]; long stride[9]; long first[10];      // first: flat float4 offset
};
__global__ __launch_bounds__(256) void fold_slabs_kernel(const FoldArgs a) {
  const long total = a.first[a.nseg];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int seg = 0;
    while (seg + 1 < a.nseg && i >= a.first[seg + 1]) ++seg;
    const long e = (i - a.first[seg]) * 4;
    const float* p = a.slabs[seg] + e;
    float4 s = ld4(a.dst[seg] + e);
    for (int k = 0; k < a.ks[seg]; ++k) s = add4(s, ld4(p + (long)k * a.stride[seg]));
    st4(a.dst[seg] + e, s);
  }
}

// ---- column sums in a fixed order (bias gradients, residual-weight gradients) ---------------------------------------
// dst[c] += sum_r x[r, c]: a workgroup owns 64 columns; its 16 waves take the rows r = phase (mod 16) in order, eight loads
// in flight, and the 16 partial sums are folded in phase order — the result depends on the data only.
struct ColsumArgs {
  int nseg;
  const float* x[8]; long ld[8]; int rows[8]; int cols[8]; float* dst[8]; int first_block[9];
};
__global__ __launch_bounds__(1024) void colsum_det_kernel(const ColsumArgs a) {
  __shared__ float sh[16][64];
  int seg = 0;
  while (seg + 1 < a.nseg && (int)blockIdx.x >= a.first_block[seg + 1]) ++seg;
  const int col = ((int)blockIdx.x - a.first_block[seg]) * 64 + (threadIdx.x & 63);
  const int phase = threadIdx.x >> 6;
  const float* x = a.x[seg];
  const long ld = a.ld[seg];
  const int rows = a.rows[seg];
  const bool ok = col < a.cols[seg];
  float s = 0.f;
  for (int r0 = phase; r0 < rows; r0 += 16 * 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = r0 + 16 * j;
      v[j] = (ok && r < rows) ? x[(long)r * ld + col] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  sh[phase][threadIdx.x & 63] = s;
  __syncthreads();
  if (phase == 0 && ok) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) t += sh[p][threadIdx.x];
    a.dst[seg][col] += t;
  }
}

// ---- column sums AND the dense norms in ONE launch (the last two launches of the step's last chain) ------------------------------
// Workgroups [0, ncs) are column-sum workgroups (as colsum_det_kernel); each also leaves the squared norm of the <= 64 gradient
// elements it has just written.  Workgroups [ncs, ncs + nchunks) are norm chunks (as sqnorm_seg2_kernel) of the segments NO column
// sum writes.  The last arrival folds every segment's partials in a fixed order — a column-sum segment from its column-sum
// workgroups in workgroup order, the others from their chunks in chunk order — and adds once per segment.
struct CsSqArgs {
  Seg2Args sq;                       // first[] counts chunks of the chunked segments only
  ColsumArgs cs;
  int ncs;                           // column-sum workgroups
  int seg_cs[TCAR_NSLOT];            // dense segment -> column-sum segment that writes it, or -1
};
__global__ __launch_bounds__(1024) void colsum_sqnorm_kernel(const float* __restrict__ g, const CsSqArgs a, float* __restrict__ out,
                                                             unsigned* __restrict__ scratch, const TcarSignal sig) {
  __shared__ float sh[16][64];
  __shared__ int last;
  const int nblk = gridDim.x;
  float partial = 0.f;               // (valid in thread 0)
  if ((int)blockIdx.x < a.ncs) {
    int seg = 0;
    while (seg + 1 < a.cs.nseg && (int)blockIdx.x >= a.cs.first_block[seg + 1]) ++seg;
    const int col = ((int)blockIdx.x - a.cs.first_block[seg]) * 64 + (threadIdx.x & 63);
    const int phase = threadIdx.x >> 6;
    const float* x = a.cs.x[seg];
    const long ld = a.cs.ld[seg];
    const int rows = a.cs.rows[seg];
    const bool ok = col < a.cs.cols[seg];
    float s = 0.f;
    for (int r0 = phase; r0 < rows; r0 += 16 * 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = r0 + 16 * j;
        v[j] = (ok && r < rows) ? x[(long)r * ld + col] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    sh[phase][threadIdx.x & 63] = s;
    __syncthreads();
    if (phase == 0) {                // the first wave: fold in phase order, write, norm of what was written
      float val = 0.f;
      if (ok) {
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) t += sh[p][threadIdx.x];
        val = a.cs.dst[seg][col] + t;
        // behind a completion flag (the step's tail join is a light poll: no L2 write-back) the Adam update of ANOTHER stream reads
        // these vectors: write-through, like every other output of a flag-carrying launch (tcar_common.h; gemm_x3_kernel)
        if (sig.cnt) __hip_atomic_store(&a.cs.dst[seg][col], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a.cs.dst[seg][col] = val;
      }
      partial = wave_sum(val * val);
    }
  } else {
    const int chunk = (int)blockIdx.x - a.ncs;
    int seg = 0;
    while (seg + 1 < a.sq.s.nseg && chunk >= a.sq.first[seg + 1]) ++seg;
    const long c0 = (long)(chunk - a.sq.first[seg]) * SQ_CHUNK;
    const long len = a.sq.s.len[seg] - c0 < SQ_CHUNK ? a.sq.s.len[seg] - c0 : (long)SQ_CHUNK;
    const float* p = g + a.sq.s.off[seg] + c0;
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long e = (long)(i * 1024 + (int)threadIdx.x) * 4;
      v[i] = (e < len) ? ld4(p + e) : zero4();
    }
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sm += dot4(v[i], v[i]);
    sm = wave_sum(sm);
    if ((threadIdx.x & 63) == 0) sh[0][threadIdx.x >> 6] = sm;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < 16; ++w) t += sh[0][w];
      partial = t;
    }
  }
  float* part = reinterpret_cast<float*>(scratch + 1);
  if (threadIdx.x == 0) {
    __hip_atomic_store(part + blockIdx.x, partial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // release the partial, acquire everybody else's when this is the last arrival
    const unsigned old = __hip_atomic_fetch_add(scratch, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = (old == (unsigned)nblk - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (last && threadIdx.x < 64) {
    const int l = threadIdx.x;
    if (l < a.sq.s.nseg) {
      float t = 0.f;
      const int j = a.seg_cs[l];
      if (j >= 0) {
        for (int b = a.cs.first_block[j]; b < a.cs.first_block[j + 1]; ++b) t += __hip_atomic_load(part + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        for (int c = a.sq.first[l]; c < a.sq.first[l + 1]; ++c) t += __hip_atomic_load(part + a.ncs + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      atomicAdd(out + a.sq.s.slot[l], t);                 // one add per segment
    }
    if (l == 0) __hip_atomic_store(scratch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // reusable by the next launch
  }
  tcar_signal_done(sig);
}

// One Adam element (TF-1 form) with EVERY rounding pinned (explicit mul / fma / div intrinsics): the update is
// compiled into several kernels (all-in-one, early, rest, streaming rest) that must agree bit for bit, so nothing is left to the
// compiler's per-kernel contraction choices.
__device__ __forceinline__ void adam1(float& w, float g, float& m, float& v, float sc, float lr_t, float b1, float b2, float eps) {
  const float gx = __fmul_rn(g, sc);
  m = __fmaf_rn(b1, m, __fmul_rn(1.f - b1, gx));
  v = __fmaf_rn(b2, v, __fmul_rn(__fmul_rn(1.f - b2, gx), gx));
  w = __fsub_rn(w, __fdiv_rn(__fmul_rn(lr_t, m), __fadd_rn(__fsqrt_rn(v), eps)));
}
__device__ __forceinline__ void adam4(float4& w, const float4 g, float4& m, float4& v, float sc, float lr_t, float b1,
                                      float b2, float eps) {
  adam1(w.x, g.x, m.x, v.x, sc, lr_t, b1, b2, eps);
  adam1(w.y, g.y, m.y, v.y, sc, lr_t, b1, b2, eps);
  adam1(w.z, g.z, m.z, v.z, sc, lr_t, b1, b2, eps);
  adam1(w.w, g.w, m.w, v.w, sc, lr_t, b1, b2, eps);
}

__device__ __forceinline__ void arena_adam_block(int seg, int chunk, float* __restrict__ w, const float* __restrict__ g,
                                                 float* __restrict__ m, float* __restrict__ v, const SegArgs& a,
                                                 const float* __restrict__ sqn_dense, const float* __restrict__ sqn_pieces,
                                                 const int32_t* __restrict__ use_dense, float clip, float lr_t, float b1,
                                                 float b2, float eps) {
  const long off = a.s.off[seg];
  const long len = a.s.len[seg];
  const long base = (long)chunk * 4096;
  if (base >= len) return;
  const float sc = clip_factor(sqn_dense, sqn_pieces, use_dense, a.s.slot[seg], clip);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long e = base + (i * 256 + threadIdx.x) * 4;
    if (e < len) {
      const long p = off + e;
      float4 ww = ld4(w + p), mm = ld4(m + p), vv = ld4(v + p);
      adam4(ww, ld4(g + p), mm, vv, sc, lr_t, b1, b2, eps);
      st4(w + p, ww); st4(m + p, mm); st4(v + p, vv);
    }
  }
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, const SegArgs a,
                                                        const float* __restrict__ sqn_dense,
                                                        const float* __restrict__ sqn_pieces,
                                                        const int32_t* __restrict__ use_dense, float clip, float lr_t,
                                                        float b1, float b2, float eps) {
  arena_adam_block(blockIdx.y, blockIdx.x, w, g, m, v, a, sqn_dense, sqn_pieces, use_dense, clip, lr_t, b1, b2, eps);
}

// item table inside E: w [rows, cols] with leading dim ldw; g, m, v compact [rows, cols]; blocks [0, nblk) stride over it
template <bool NT = false>
__device__ __forceinline__ void item_adam_blocks(int blk, int nblk, float* __restrict__ w, long ldw,
                                                 const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                 long rows, int cols, int slot, const float* __restrict__ sqn_dense,
                                                 const float* __restrict__ sqn_pieces, const int32_t* __restrict__ use_dense,
                                                 float clip, float lr_t, float b1, float b2, float eps,
                                                 __bf16* __restrict__ eh, __bf16* __restrict__ el, long ld16,
                                                 const uint32_t* __restrict__ skip = nullptr) {
  const float sc = clip_factor(sqn_dense, sqn_pieces, use_dense, slot, clip);
  // (eight columns per thread — eight 16-byte loads in flight, 16-byte plane stores — measured SLOWER for the streaming rest pass:
  //  85 against 65-70 us, profiles/r04_ab_experiments.txt)
  const int c4 = cols >> 2;
  const long total = rows * c4;
  for (long i = (long)blk * 256 + threadIdx.x; i < total; i += (long)nblk * 256) {
    const long r = i / c4;
    if (skip && ((skip[r >> 5] >> (r & 31)) & 1u)) continue;      // row already updated by the early pass
    const int c = (int)(i - r * c4) * 4;
    const long p = r * cols + c;
    float* wp = w + r * ldw + c;
    float4 ww, mm, vv, gg;
    if (NT) { ww = ld4_nt(wp); mm = ld4_nt(m + p); vv = ld4_nt(v + p); gg = ld4_nt(g + p); }
    else { ww = ld4(wp); mm = ld4(m + p); vv = ld4(v + p); gg = ld4(g + p); }
    adam4(ww, gg, mm, vv, sc, lr_t, b1, b2, eps);
    if (NT) { st4_stream(wp, ww); st4_stream(m + p, mm); st4_stream(v + p, vv); }
    else { st4(wp, ww); st4(m + p, mm); st4(v + p, vv); }
    if (eh) {   // keep the bf16 hi / lo planes of the candidate matrix in step with the fp32 master (gemm_bf16.hip)
      const float wv[4] = {ww.x, ww.y, ww.z, ww.w};
      bf16x4_t h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) { h[j] = (__bf16)wv[j]; l[j] = (__bf16)(wv[j] - (float)h[j]); }
      const long o = kb32_off(r, c, (int)(ld16 >> 5));          // KB32 blocked plane, inner dimension ld16
      *reinterpret_cast<bf16x4_t*>(eh + o) = h;
      *reinterpret_cast<bf16x4_t*>(el + o) = l;
    }
  }
}

__global__ __launch_bounds__(256) void clip_adam_2d_kernel(float* __restrict__ w, long ldw, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, long rows,
                                                           int cols, int slot, const float* __restrict__ sqn_dense,
                                                           const float* __restrict__ sqn_pieces,
                                                           const int32_t* __restrict__ use_dense, float clip, float lr_t,
                                                           float b1, float b2, float eps, __bf16* __restrict__ eh,
                                                           __bf16* __restrict__ el, long ld16) {
  item_adam_blocks(blockIdx.x, gridDim.x, w, ldw, g, m, v, rows, cols, slot, sqn_dense, sqn_pieces, use_dense, clip, lr_t, b1,
                   b2, eps, eh, el, ld16);
}

// every variable in ONE launch: blocks [0, n2d) take the item table, the rest the (segment, chunk) pairs of the arena
struct AdamAll {
  float* w2; long ldw; const float* g2; float* m2; float* v2; long rows; int cols, slot; __bf16* eh; __bf16* el; long ld16;
  int n2d, gx;
  float* w; const float* g; float* m; float* v;
  const float* sqn_dense; const float* sqn_pieces; const int32_t* use_dense;
  float clip, lr_t, b1, b2, eps;
};
__global__ __launch_bounds__(256) void clip_adam_all_kernel(const AdamAll p, const SegArgs a) {
  if ((int)blockIdx.x < p.n2d) {
    item_adam_blocks(blockIdx.x, p.n2d, p.w2, p.ldw, p.g2, p.m2, p.v2, p.rows, p.cols, p.slot, p.sqn_dense, p.sqn_pieces,
                     p.use_dense, p.clip, p.lr_t, p.b1, p.b2, p.eps, p.eh, p.el, p.ld16);
  } else {
    const int idx = blockIdx.x - p.n2d;
    arena_adam_block(idx / p.gx, idx % p.gx, p.w, p.g, p.m, p.v, a, p.sqn_dense, p.sqn_pieces, p.use_dense, p.clip, p.lr_t,
                     p.b1, p.b2, p.eps);
  }
}

// ---- split update (deferred item-table Adam) ---------------------------------------------------------------------
// The NEXT step's gather needs only the item rows of its own sessions and its small kernels leave the chip idle; the
// item table's Adam is an HBM-bound pass over all rows.  So the update is issued in two parts: EARLY = the arena + the
// item rows listed in `ids` (the next batch's session items; a bitmap makes every row update exactly once although ids
// repeat), REST = every other row, which the step driver runs on the aux stream beside the next forward pass.
struct AdamEarly {
  AdamAll p;
  const int32_t* ids; long n_ids; uint32_t* bitmap; int n_rowblk;
  const int32_t* ids2; long n_ids2;      // a second list, 0-BASED rows (the batch's labels: step.hip, anchored softmax form); may be empty
};
__device__ __forceinline__ void clip_adam_early_body(const AdamEarly& e, const SegArgs& a) {
  const AdamAll& p = e.p;
  if ((int)blockIdx.x >= e.n_rowblk) {
    const int idx = blockIdx.x - e.n_rowblk;
    arena_adam_block(idx / p.gx, idx % p.gx, p.w, p.g, p.m, p.v, a, p.sqn_dense, p.sqn_pieces, p.use_dense, p.clip, p.lr_t,
                     p.b1, p.b2, p.eps);
    return;
  }
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // one wave per listed id
  if (i >= e.n_ids + e.n_ids2) return;
  long r = i < e.n_ids ? (long)e.ids[i] - 1 : (long)e.ids2[i - e.n_ids];      // ids are 1-based item ids (sampler.py:67), ids2 0-based rows
  r = r < 0 ? 0 : (r >= p.rows ? p.rows - 1 : r);
  unsigned old = 0;
  if (lane == 0) old = atomicOr(e.bitmap + (r >> 5), 1u << (r & 31));
  old = __shfl(old, 0);
  if ((old >> (r & 31)) & 1u) return;                                 // another wave owns this row
  const float sc = clip_factor(p.sqn_dense, p.sqn_pieces, p.use_dense, p.slot, p.clip);
  for (int c = lane * 4; c < p.cols; c += 256) {
    const long q = r * p.cols + c;
    float* wp = p.w2 + r * p.ldw + c;
    float4 ww = ld4(wp), mm = ld4(p.m2 + q), vv = ld4(p.v2 + q);
    adam4(ww, ld4(p.g2 + q), mm, vv, sc, p.lr_t, p.b1, p.b2, p.eps);
    st4(wp, ww); st4(p.m2 + q, mm); st4(p.v2 + q, vv);
    if (p.eh) {
      const float wv[4] = {ww.x, ww.y, ww.z, ww.w};
      bf16x4_t h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) { h[j] = (__bf16)wv[j]; l[j] = (__bf16)(wv[j] - (float)h[j]); }
      const long o = kb32_off(r, c, (int)(p.ld16 >> 5));
      *reinterpret_cast<bf16x4_t*>(p.eh + o) = h;
      *reinterpret_cast<bf16x4_t*>(p.el + o) = l;
    }
  }
}
__global__ __launch_bounds__(256) void clip_adam_early_kernel(const AdamEarly e, const SegArgs a) {
  clip_adam_early_body(e, a);
}
template <bool NT>
__global__ __launch_bounds__(256) void clip_adam_rest_kernel(const AdamAll p, const uint32_t* __restrict__ skip) {
  item_adam_blocks<NT>(blockIdx.x, gridDim.x, p.w2, p.ldw, p.g2, p.m2, p.v2, p.rows, p.cols, p.slot, p.sqn_dense, p.sqn_pieces,
                       p.use_dense, p.clip, p.lr_t, p.b1, p.b2, p.eps, p.eh, p.el, p.ld16, skip);
}

int seg_grid_x(const tcar_segments_t* s) {
  int64_t mx = 0;
  for (int i = 0; i < s->nseg; ++i) mx = s->len[i] > mx ? s->len[i] : mx;
  return (int)((mx + 4095) / 4096);
}
int check_segs(const tcar_segments_t* s) {
  if (!s || s->nseg < 0 || s->nseg > TCAR_NSLOT) return TCAR_E_ARG;
  for (int i = 0; i < s->nseg; ++i)
    if ((s->off[i] & 3) || (s->len[i] & 3) || s->slot[i] < 0 || s->slot[i] >= TCAR_NSLOT) return TCAR_E_ARG;
  return TCAR_OK;
}

}  // namespace

static int sqnorm_grid_x(const tcar_segments_t* segs) {
  long mx = 1;
  for (int i = 0; i < segs->nseg; ++i) mx = segs->len[i] > mx ? segs->len[i] : mx;
  long gx = (mx + 8191) / 8192;
  return (int)(gx > 512 ? 512 : gx);
}

extern "C" int tcar_sqnorm(const float* g, const tcar_segments_t* segs, float* sqn_dense, void* stream) {
  return tcar_sqnorm_o(g, segs, sqn_dense, stream, nullptr);
}
// (flag-capable in its one-workgroup-per-segment form: the norm slots are published with atomics)
int tcar_sqnorm_o(const float* g, const tcar_segments_t* segs, float* sqn_dense, void* stream, TcarOpt* o) {
  if (check_segs(segs) || !g || !sqn_dense || !tcar_aligned16(g)) return TCAR_E_ARG;
  if (segs->nseg == 0) return TCAR_OK;
  SegArgs a;
  a.s = *segs;
  long longest = 0;
  for (int i = 0; i < segs->nseg; ++i) longest = segs->len[i] > longest ? segs->len[i] : longest;
  if (o && o->scratch && o->scratch_words > 1 && segs->nseg <= 64) {
    // several workgroups per segment + an order-fixed fold by the last arrival (needs zeroed scratch words of the context)
    Seg2Args b;
    b.s = *segs;
    int n = 0;
    for (int i = 0; i < segs->nseg; ++i) { b.first[i] = n; n += (int)((segs->len[i] + SQ_CHUNK - 1) / SQ_CHUNK); }
    b.first[segs->nseg] = n;
    if (n > 0 && n + 1 <= o->scratch_words) {
      TCAR_LAUNCH(sqnorm_seg2_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, g, b, sqn_dense, o->scratch, tcar_sig(o));
      TCAR_CHECK_LAUNCH();
      return TCAR_OK;
    }
  }
  if (longest <= 262144)
    TCAR_LAUNCH(sqnorm_seg_kernel, dim3(segs->nseg), dim3(1024), 0, (hipStream_t)stream, g, a, sqn_dense, tcar_sig(o));
  else
    TCAR_LAUNCH(sqnorm_kernel, dim3(sqnorm_grid_x(segs), segs->nseg), dim3(256), 0, (hipStream_t)stream, g, a, sqn_dense);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_clip_adam(float* w, const float* g, float* m, float* v, const tcar_segments_t* segs,
                              const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense, float clip,
                              float lr_t, float b1, float b2, float eps, void* stream) {
  if (check_segs(segs) || !w || !g || !m || !v || !sqn_dense || !sqn_pieces || !use_dense) return TCAR_E_ARG;
  if (segs->nseg == 0) return TCAR_OK;
  SegArgs a;
  a.s = *segs;
  TCAR_LAUNCH(clip_adam_kernel, dim3(seg_grid_x(segs), segs->nseg), dim3(256), 0, (hipStream_t)stream, w, g, m, v,
                     a, sqn_dense, sqn_pieces, use_dense, clip, lr_t, b1, b2, eps);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_clip_adam_2d_bf16(float* w, int64_t ldw, const float* g, float* m, float* v, int64_t rows,
                                      int32_t cols, int32_t slot, const float* sqn_dense, const float* sqn_pieces,
                                      const int32_t* use_dense, float clip, float lr_t, float b1, float b2, float eps,
                                      void* e16_hi, void* e16_lo, int64_t ld16, void* stream) {
  if (e16_hi && (!e16_lo || (ld16 & 31))) return TCAR_E_ARG;
  if (!w || !g || !m || !v || rows <= 0 || cols <= 0 || (cols & 3) || (ldw & 3) || slot < 0 || slot >= TCAR_NSLOT)
    return TCAR_E_ARG;
  long total = rows * (cols >> 2);
  int grid = (int)((total + 256 * 4 - 1) / (256 * 4));
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  TCAR_LAUNCH(clip_adam_2d_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (long)ldw, g, m, v, (long)rows,
                     (int)cols, (int)slot, sqn_dense, sqn_pieces, use_dense, clip, lr_t, b1, b2, eps, (__bf16*)e16_hi,
              (__bf16*)e16_lo, (long)ld16);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_clip_adam_2d(float* w, int64_t ldw, const float* g, float* m, float* v, int64_t rows, int32_t cols,
                                 int32_t slot, const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense,
                                 float clip, float lr_t, float b1, float b2, float eps, void* stream) {
  return tcar_clip_adam_2d_bf16(w, ldw, g, m, v, rows, cols, slot, sqn_dense, sqn_pieces, use_dense, clip, lr_t, b1, b2,
                                eps, nullptr, nullptr, 0, stream);
}

extern "C" int tcar_clip_adam_all(float* w, const float* g, float* m, float* v, const tcar_segments_t* segs, float* w2d,
                                  int64_t ldw, const float* g2d, float* m2d, float* v2d, int64_t rows, int32_t cols,
                                  int32_t slot, const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense,
                                  float clip, float lr_t, float b1, float b2, float eps, void* e16_hi, void* e16_lo,
                                  int64_t ld16, void* stream) {
  if (check_segs(segs) || !w || !g || !m || !v || !sqn_dense || !sqn_pieces || !use_dense) return TCAR_E_ARG;
  if (e16_hi && (!e16_lo || (ld16 & 31))) return TCAR_E_ARG;
  if (!w2d || !g2d || !m2d || !v2d || rows <= 0 || cols <= 0 || (cols & 3) || (ldw & 3) || slot < 0 || slot >= TCAR_NSLOT)
    return TCAR_E_ARG;
  AdamAll p;
  p.w2 = w2d; p.ldw = ldw; p.g2 = g2d; p.m2 = m2d; p.v2 = v2d; p.rows = rows; p.cols = cols; p.slot = slot;
  p.eh = (__bf16*)e16_hi; p.el = (__bf16*)e16_lo; p.ld16 = ld16;
  long total = rows * (cols >> 2);
  long n2d = (total + 256 * 4 - 1) / (256 * 4);
  p.n2d = (int)(n2d > 4096 ? 4096 : (n2d < 1 ? 1 : n2d));
  p.gx = segs->nseg ? seg_grid_x(segs) : 1;
  if (p.gx < 1) p.gx = 1;
  p.w = w; p.g = g; p.m = m; p.v = v;
  p.sqn_dense = sqn_dense; p.sqn_pieces = sqn_pieces; p.use_dense = use_dense;
  p.clip = clip; p.lr_t = lr_t; p.b1 = b1; p.b2 = b2; p.eps = eps;
  SegArgs a;
  a.s = *segs;
  const int grid = p.n2d + p.gx * segs->nseg;
  TCAR_LAUNCH(clip_adam_all_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

static int fill_adam_all(AdamAll& p, float* w, const float* g, float* m, float* v, const tcar_segments_t* segs, float* w2d,
                         int64_t ldw, const float* g2d, float* m2d, float* v2d, int64_t rows, int32_t cols, int32_t slot,
                         const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense, float clip, float lr_t,
                         float b1, float b2, float eps, void* e16_hi, void* e16_lo, int64_t ld16) {
  if (!sqn_dense || !sqn_pieces || !use_dense) return TCAR_E_ARG;
  if (e16_hi && (!e16_lo || (ld16 & 31))) return TCAR_E_ARG;
  if (!w2d || !g2d || !m2d || !v2d || rows <= 0 || cols <= 0 || (cols & 3) || (ldw & 3) || slot < 0 || slot >= TCAR_NSLOT)
    return TCAR_E_ARG;
  p.w2 = w2d; p.ldw = ldw; p.g2 = g2d; p.m2 = m2d; p.v2 = v2d; p.rows = rows; p.cols = cols; p.slot = slot;
  p.eh = (__bf16*)e16_hi; p.el = (__bf16*)e16_lo; p.ld16 = ld16;
  long total = rows * (cols >> 2);
  long n2d = (total + 256 * 4 - 1) / (256 * 4);
  p.n2d = (int)(n2d > 4096 ? 4096 : (n2d < 1 ? 1 : n2d));
  p.gx = (segs && segs->nseg) ? seg_grid_x(segs) : 1;
  if (p.gx < 1) p.gx = 1;
  p.w = w; p.g = g; p.m = m; p.v = v;
  p.sqn_dense = sqn_dense; p.sqn_pieces = sqn_pieces; p.use_dense = use_dense;
  p.clip = clip; p.lr_t = lr_t; p.b1 = b1; p.b2 = b2; p.eps = eps;
  return TCAR_OK;
}

extern "C" int tcar_clip_adam_early(float* w, const float* g, float* m, float* v, const tcar_segments_t* segs, float* w2d,
                                    int64_t ldw, const float* g2d, float* m2d, float* v2d, int64_t rows, int32_t cols,
                                    int32_t slot, const float* sqn_dense, const float* sqn_pieces,
                                    const int32_t* use_dense, float clip, float lr_t, float b1, float b2, float eps,
                                    void* e16_hi, void* e16_lo, int64_t ld16, const int32_t* ids, int64_t n_ids,
                                    uint32_t* bitmap, void* stream) {
  return tcar_clip_adam_early_2(w, g, m, v, segs, w2d, ldw, g2d, m2d, v2d, rows, cols, slot, sqn_dense, sqn_pieces, use_dense, clip, lr_t,
                                b1, b2, eps, e16_hi, e16_lo, ld16, ids, n_ids, nullptr, 0, bitmap, stream);
}
// ... with a second list of 0-BASED rows that join the early part (internal: the labels of the batch, whose rows the anchored
// softmax form reads in the forward pass, before the rest pass has ended)
int tcar_clip_adam_early_2(float* w, const float* g, float* m, float* v, const tcar_segments_t* segs, float* w2d, int64_t ldw,
                           const float* g2d, float* m2d, float* v2d, int64_t rows, int32_t cols, int32_t slot, const float* sqn_dense,
                           const float* sqn_pieces, const int32_t* use_dense, float clip, float lr_t, float b1, float b2, float eps,
                           void* e16_hi, void* e16_lo, int64_t ld16, const int32_t* ids, int64_t n_ids, const int32_t* ids2,
                           int64_t n_ids2, uint32_t* bitmap, void* stream) {
  if (check_segs(segs) || !w || !g || !m || !v || !bitmap || n_ids < 0 || (n_ids > 0 && !ids) || n_ids2 < 0 || (n_ids2 > 0 && !ids2))
    return TCAR_E_ARG;
  AdamEarly e;
  const int rc = fill_adam_all(e.p, w, g, m, v, segs, w2d, ldw, g2d, m2d, v2d, rows, cols, slot, sqn_dense, sqn_pieces,
                               use_dense, clip, lr_t, b1, b2, eps, e16_hi, e16_lo, ld16);
  if (rc) return rc;
  e.ids = ids; e.n_ids = n_ids; e.bitmap = bitmap;
  e.ids2 = ids2; e.n_ids2 = n_ids2;
  e.n_rowblk = (int)((n_ids + n_ids2 + 3) / 4);
  SegArgs a;
  a.s = *segs;
  const int grid = e.n_rowblk + e.p.gx * segs->nseg;
  if (grid <= 0) return TCAR_OK;
  TCAR_LAUNCH(clip_adam_early_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, e, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_clip_adam_rest_keep(float* w2d, int64_t ldw, const float* g2d, float* m2d, float* v2d, int64_t rows,
                                   int32_t cols, int32_t slot, const float* sqn_dense, const float* sqn_pieces,
                                   const int32_t* use_dense, float clip, float lr_t, float b1, float b2, float eps,
                                   void* e16_hi, void* e16_lo, int64_t ld16, uint32_t* bitmap, void* stream) {
  return tcar_clip_adam_rest_keep_o(w2d, ldw, g2d, m2d, v2d, rows, cols, slot, sqn_dense, sqn_pieces, use_dense, clip, lr_t, b1, b2,
                                    eps, e16_hi, e16_lo, ld16, bitmap, stream, tcar_fixed::rest_grid);
}
int tcar_clip_adam_rest_keep_o(float* w2d, int64_t ldw, const float* g2d, float* m2d, float* v2d, int64_t rows, int32_t cols,
                               int32_t slot, const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense, float clip,
                               float lr_t, float b1, float b2, float eps, void* e16_hi, void* e16_lo, int64_t ld16,
                               uint32_t* bitmap, void* stream, int rest_grid) {
  if (!bitmap) return TCAR_E_ARG;
  AdamAll p;
  const int rc = fill_adam_all(p, nullptr, nullptr, nullptr, nullptr, nullptr, w2d, ldw, g2d, m2d, v2d, rows, cols, slot,
                               sqn_dense, sqn_pieces, use_dense, clip, lr_t, b1, b2, eps, e16_hi, e16_lo, ld16);
  if (rc) return rc;
  // a modest grid: the pass shares the chip with the latency-bound kernels of the forward head and has ~100 us to finish
  const int cap = rest_grid;
  if (cap > 0 && p.n2d > cap) p.n2d = cap;
  // the pass streams 376 MB once: non-temporal loads and stores, so that it does not evict the weights and activations of the
  // latency-bound session-side kernels it runs beside (A/B over three interleaved rounds: 0.6197 vs 0.6327 ms per step; the same
  // treatment of dE's result, of the candidate-time backward's read, of the slab reduce and of the dense-norm partials measured
  // within +-0.5 % and was not kept)
  TCAR_LAUNCH(clip_adam_rest_kernel<true>, dim3(p.n2d), dim3(256), 0, (hipStream_t)stream, p, (const uint32_t*)bitmap);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}


extern "C" int tcar_clip_adam_rest(float* w2d, int64_t ldw, const float* g2d, float* m2d, float* v2d, int64_t rows,
                                   int32_t cols, int32_t slot, const float* sqn_dense, const float* sqn_pieces,
                                   const int32_t* use_dense, float clip, float lr_t, float b1, float b2, float eps,
                                   void* e16_hi, void* e16_lo, int64_t ld16, uint32_t* bitmap, void* stream) {
  const int rc = tcar_clip_adam_rest_keep(w2d, ldw, g2d, m2d, v2d, rows, cols, slot, sqn_dense, sqn_pieces, use_dense, clip, lr_t, b1,
                                          b2, eps, e16_hi, e16_lo, ld16, bitmap, stream);
  if (rc) return rc;
  // every row is up to date now: clear the marks for the next step (stream ordered behind the kernel)
  if (hipMemsetAsync(bitmap, 0, (size_t)(((rows + 31) / 32 + 15) & ~15) * sizeof(uint32_t), (hipStream_t)stream) != hipSuccess)
    return TCAR_E_LAUNCH;
  return TCAR_OK;
}

extern "C" int tcar_colsum_det(int nseg, const tcar_colsum_t* segs, void* stream) {
  if (nseg <= 0) return TCAR_OK;
  if (nseg > 8 || !segs) return TCAR_E_ARG;
  ColsumArgs a{};
  int blocks = 0;
  for (int i = 0; i < nseg; ++i) {
    if (!segs[i].x || !segs[i].dst || segs[i].rows < 0 || segs[i].cols <= 0 || segs[i].ld < segs[i].cols) return TCAR_E_ARG;
    a.x[i] = segs[i].x; a.ld[i] = (long)segs[i].ld; a.rows[i] = segs[i].rows; a.cols[i] = segs[i].cols; a.dst[i] = segs[i].dst;
    a.first_block[i] = blocks;
    blocks += (segs[i].cols + 63) / 64;
  }
  a.first_block[nseg] = blocks;
  a.nseg = nseg;
  TCAR_LAUNCH(colsum_det_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// tcar_colsum_det + tcar_sqnorm in one launch (internal; the step driver's last chain).  Every column-sum destination must be the
// START of one segment of `segs` (its gradient vector) with cols <= that segment's length; TCAR_E_ARG otherwise, or when the scratch
// words of `o` do not hold 1 + workgroups — the caller then launches the two separately.  Flag-capable like tcar_sqnorm_o.
int tcar_colsum_sqnorm_o(const float* g, const tcar_segments_t* segs, int ncs, const tcar_colsum_t* cs, float* sqn_dense, void* stream,
                         TcarOpt* o) {
  if (check_segs(segs) || !g || !sqn_dense || !tcar_aligned16(g) || ncs <= 0 || ncs > 8 || !cs || !o || !o->scratch || segs->nseg > 64)
    return TCAR_E_ARG;
  CsSqArgs a{};
  a.sq.s = *segs;
  for (int i = 0; i < TCAR_NSLOT; ++i) a.seg_cs[i] = -1;
  int blocks = 0;
  a.cs.nseg = ncs;
  for (int j = 0; j < ncs; ++j) {
    if (!cs[j].x || !cs[j].dst || cs[j].rows < 0 || cs[j].cols <= 0 || cs[j].ld < cs[j].cols) return TCAR_E_ARG;
    int hit = -1;
    for (int i = 0; i < segs->nseg; ++i)
      if (g + segs->off[i] == cs[j].dst && cs[j].cols <= segs->len[i] && a.seg_cs[i] < 0) { hit = i; break; }
    if (hit < 0) return TCAR_E_ARG;
    a.seg_cs[hit] = j;
    a.cs.x[j] = cs[j].x; a.cs.ld[j] = (long)cs[j].ld; a.cs.rows[j] = cs[j].rows; a.cs.cols[j] = cs[j].cols; a.cs.dst[j] = cs[j].dst;
    a.cs.first_block[j] = blocks;
    blocks += (cs[j].cols + 63) / 64;
  }
  a.cs.first_block[ncs] = blocks;
  a.ncs = blocks;
  int n = 0;
  for (int i = 0; i < segs->nseg; ++i) {
    a.sq.first[i] = n;
    // (a column-sum segment whose vector is shorter than its segment: the rest of the segment is padding the arena keeps zero)
    if (a.seg_cs[i] < 0) n += (int)((segs->len[i] + SQ_CHUNK - 1) / SQ_CHUNK);
  }
  a.sq.first[segs->nseg] = n;
  if (blocks + n + 1 > o->scratch_words) return TCAR_E_ARG;
  TCAR_LAUNCH(colsum_sqnorm_kernel, dim3(blocks + n), dim3(1024), 0, (hipStream_t)stream, g, a, sqn_dense, o->scratch, tcar_sig(o));
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_fold_slabs(int nseg, const tcar_fold_t* segs, void* stream) {
  if (nseg <= 0) return TCAR_OK;
  if (nseg > 9 || !segs) return TCAR_E_ARG;
  FoldArgs a{};
  long at = 0;
  for (int i = 0; i < nseg; ++i) {
    if (!segs[i].dst || !segs[i].slabs || segs[i].n <= 0 || (segs[i].n & 3) || segs[i].ks < 1 || (segs[i].stride & 3) ||
        !tcar_aligned16(segs[i].dst) || !tcar_aligned16(segs[i].slabs))
      return TCAR_E_ARG;
    a.dst[i] = segs[i].dst; a.slabs[i] = segs[i].slabs; a.n[i] = (long)segs[i].n; a.ks[i] = segs[i].ks; a.stride[i] = (long)segs[i].stride;
    a.first[i] = at;
    at += segs[i].n / 4;
  }
  a.first[nseg] = at;
  a.nseg = nseg;
  long blocks = (at + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  TCAR_LAUNCH(fold_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
