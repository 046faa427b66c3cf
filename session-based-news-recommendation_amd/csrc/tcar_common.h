// Shared device helpers for the TCAR gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tcar_hip.h"

#define TCAR_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) {                                  \
      fprintf(stderr, "tcar: HIP error %d (%s) at %s:%d\n", (int)e__, hipGetErrorString(e__), __FILE__, __LINE__); \
      return TCAR_E_LAUNCH;                                   \
    }                                                         \
  } while (0)

// hipGetLastError() is sticky per thread: clear whatever an earlier (foreign) runtime call left behind, so the
// check after the launch reports THIS launch only.
#define TCAR_LAUNCH(...)                 \
  do {                                   \
    (void)hipGetLastError();             \
    hipLaunchKernelGGL(__VA_ARGS__);     \
  } while (0)

static inline bool tcar_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// One-time, per-DEVICE kernel attributes (hipFuncSetAttribute for > 64 KB of dynamic LDS).  One bit per device ordinal:
// correct with several devices in one process and safe from several host threads (two threads may both set the
// attribute once: the call is idempotent).  This is the only mutable state the library keeps.
#include <atomic>
struct TcarOnce {
  std::atomic<unsigned long long> mask{0};
};
static inline bool tcar_first_on_device(TcarOnce& o) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (o.mask.load(std::memory_order_acquire) & bit) return false;
  o.mask.fetch_or(bit, std::memory_order_acq_rel);
  return true;
}
#define TCAR_SET_LDS_ONCE(kernel, bytes)                                                                     \
  do {                                                                                                        \
    static TcarOnce once__;                                                                                   \
    if (tcar_first_on_device(once__))                                                                         \
      (void)hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
  } while (0)

// Diagnostic tuning switches (environment, read ONCE per process at first use — C++11 thread-safe static — never per
// launch).  Defaults are the shipped configuration; README.md lists them.  Defined in step.hip.
struct TcarTuning {
  int bf16_tile;        // TCAR_BF16_TILE      force a workgroup tile of the bf16 GEMM (0 = heuristic; tests pin every tile shape with it)
  int rest_grid;        // TCAR_REST_GRID      grid cap of the deferred Adam rest pass
  int softmax_variant;  // TCAR_SOFTMAX_VARIANT
  int wgrad_ks;         // TCAR_WGRAD_KS       K chunk of the weight-gradient split
  int gather_big_rows;  // TCAR_GATHER_BIG_ROWS  session rows from which the forward gather runs its throughput form
  int gather_wg_per_cu; // TCAR_GATHER_WG      1024-thread workgroups per CU of that form (2 x 78 KB of LDS fit)
  int mha_mfma;         // TCAR_MHA_MFMA       0: multihead_attention core always in its scalar form
  int sort_scatter;     // TCAR_SORT_SCATTER   0: item-row scatter with float atomics instead of the sorted segmented sum
  int bf16_ks;          // TCAR_BF16_KS        64-deep LDS stages of the hi-only bf16 GEMM: 1 never, 2 dX / logits layouts, 3 all
  int det_small;        // TCAR_DET_SMALL      0: position / time / dwell table gradients through LDS + float atomics (sorted mode)
  int x3_oneshot;       // TCAR_X3_ONESHOT     0: short-K small-GEMM launches keep the one-stage register ring
  int fused_ce;         // TCAR_FUSED_CE       0: training steps materialise the fp32 logits and run the row-resident softmax kernel
  int onehot_time;      // TCAR_ONEHOT_TIME    0: the logits GEMM of a training step contracts the 5 ldt clipped candidate time columns instead of the 160-column one-hot form
  int proj_split;       // TCAR_PROJ_SPLIT     0: the session-side projections / output-transform input gradients as un-split GEMMs
  int fork_delay;       // TCAR_FORK_DELAY     us a DELAYED flag fork holds its consumer back behind the producer's end (step.hip)
  int flag_fork;        // TCAR_FLAG_FORK      0: every fork of the main stream records an event (6-7 us of bubble on it) instead of
                        //                     letting the producing kernel publish a device flag a polling kernel of the side stream waits for
};
const TcarTuning& tcar_tuning();

// Completion flag of a kernel (step.hip: fork_arm / fork_go).  A kernel that carries one publishes `epoch` to *flag when its
// LAST workgroup is through: the side stream's consumers sit behind a one-wave polling kernel instead of behind an event the main
// stream would have to record (measured, tools/micro/event_cost: a record between two kernels costs the recording stream 6.5 us,
// the flag costs it nothing and releases the consumer 0.4 us after the producer's end).  cnt == nullptr: no flag.
struct TcarSignal { unsigned* cnt; unsigned* flag; unsigned epoch; unsigned slot; };
// the flag the NEXT flag-capable launch of this host thread carries (it takes it: tcar_take_signal); defined in step.hip
TcarSignal& tcar_pending_signal();
// the epoch of the flag the launches of this host thread last TOOK for a slot (step.hip: a fork is released by a polling kernel
// only when the producing launch really carries its flag, whatever else happened to the pending one in between)
unsigned& tcar_taken_epoch(unsigned slot);
inline TcarSignal tcar_take_signal() {
  TcarSignal& p = tcar_pending_signal();
  const TcarSignal s = p;
  p = TcarSignal{};
  if (s.cnt) tcar_taken_epoch(s.slot) = s.epoch;
  return s;
}
// 16-byte write-through store (sc1): the bytes bypass the write-back state of this XCD's L2, so a consumer behind a completion
// flag needs no release fence / L2 write-back from the producer (cdna_hip_programming.md Guideline 16, R1).  The compiler does not
// count this store: drain with an explicit s_waitcnt vmcnt(0) before signalling (tcar_signal_done does).
__device__ __forceinline__ void st4_sc1(float* p, float4 v) {
  typedef float f4_t __attribute__((ext_vector_type(4)));
  const f4_t x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
// Every thread of every workgroup calls this as the kernel's last statement (no early returns ahead of it).  The workgroup's
// stores are drained into its XCD's L2 (vmcnt) before it is counted; NO release fence here — an agent-scope release in every
// workgroup is an L2 write-back per workgroup (measured: the CE-rescale kernel 24 -> 176 us, and every kernel beside it slower).
// The write-back of the eight L2s is done ONCE per XCD by the polling kernel of the consumer stream (step.hip poll_flag_kernel),
// after the flag and before the consumer's first kernel starts.
__device__ __forceinline__ void tcar_signal_done(const TcarSignal& s) {
  if (!s.cnt) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    if (__hip_atomic_fetch_add(s.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y * gridDim.z - 1u) {
      __hip_atomic_store(s.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // reusable by the slot's next launch
      __hip_atomic_store(s.flag, s.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// sum over aligned groups of `width` consecutive lanes (width = 16, 32 or 64)
__device__ __forceinline__ float group_sum(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// explicit fma chain: hipcc contracts a plain a*b + c*d + ... differently from one inlining context to the next, and two
// kernels that must agree bit for bit (the two forms of the forward gather) would round their row norms differently
__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming (non-temporal) forms: data that is touched once per step by an HBM-bound pass should not evict what the
// latency-bound kernels beside it keep in L2 / the Infinity Cache
typedef float tcar_v4f_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* p) {
  const tcar_v4f_nt v = __builtin_nontemporal_load(reinterpret_cast<const tcar_v4f_nt*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4_stream(float* p, float4 v) {
  const tcar_v4f_nt x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<tcar_v4f_nt*>(p));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float4 a, float s, float4 c) {
  return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}
__device__ __forceinline__ void atomic_add4(float* p, float4 v) {
  atomicAdd(p + 0, v.x);
  atomicAdd(p + 1, v.y);
  atomicAdd(p + 2, v.z);
  atomicAdd(p + 3, v.w);
}

// tf.clip_by_norm(row, 1.0): y = x / max(||x||, 1).  scale for a row with sum of squares ss.
__device__ __forceinline__ float clip_scale(float ss) { return ss > 1.0f ? 1.0f / sqrtf(ss) : 1.0f; }
// backward of the row clip: gx = gy/n - x (x.gy)/n^3 when n > 1, identity otherwise.
// returns (a, b) with gx = a*gy - b*x
__device__ __forceinline__ void clip_bwd_coef(float ss, float d, float& a, float& b) {
  if (ss > 1.0f) {
    float inv = 1.0f / sqrtf(ss);
    a = inv;
    b = d * inv * inv * inv;
  } else {
    a = 1.0f;
    b = 0.0f;
  }
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
