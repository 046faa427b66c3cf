// DESIGN.md §7, observation 1 — minimal repro attempt.
// Question: can a kernel whose workgroups fill their LDS allocation by direct global -> LDS DMA (global_load_lds_dwordx4, the
// staging of gemm_bf16.hip) disturb the LDS contents, or the LDS reads, of an UNRELATED kernel that shares its CUs?
//
// Victims (stream 2), all checked word by word:
//   canary     256 threads, 20 KB static LDS: every thread writes a pattern (ds_write_b128), barrier, optional delay, every thread reads
//              back what ANOTHER wave wrote (ds_read_b128) and compares; counts mismatches and keeps the first one.
//   oldreduce  the LDS phase of round 4's first reduce_dact_onehot_kernel (git f933989): 61 x 64 clipped rows staged with
//              ld4 -> ds_write_b128, a [16][64] coefficient tile with ds_write_b32, ONE barrier, then per thread
//              acc += tl[r][cg*4..] * dpl[rp][r]; compared with the same sum read straight from global memory (no LDS).
// Aggressors (stream 1), launched back to back so that workgroups retire and start all the time:
//   dma9       9 waves, 48 KB dynamic LDS, two stages of 24 one-KB copies, the copy -> barrier -> read loop of the dE GEMM
//              (transposing reads of the other stage between the copies), two workgroups per CU;
//   dma8big    8 waves, 160 KB (the logits GEMM's footprint: nothing else fits on the CU — control for "no co-residency");
//   dma_tail   dma9 that leaves its LAST stage's copies in flight at s_endpgm (no wait, no barrier): does a fill that lands after the
//              workgroup has retired hit the next tenant of that LDS range?
//   regstage   the same bytes staged global -> VGPR -> ds_write_b128 (no DMA): control.
//   none       victim alone.
// Output: one line per (aggressor, victim): launches, victim workgroup-iterations, mismatches, first mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(1))) const void* glb_vp;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

struct Report { unsigned mism; unsigned iters; unsigned first[6]; };     // first: wg, iter, word index, got, want, lane

// ------------------------------------------------------------------------------------------------ aggressors
template <int NW, int STAGE_COPIES, bool TAIL>
__global__ __launch_bounds__(64 * NW) void dma_kernel(const char* __restrict__ src, long src_kb, int iters, unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = STAGE_COPIES * 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int CPW = (STAGE_COPIES + NW - 1) / NW;
  long kb = ((long)blockIdx.x * 7919) % src_kb;
  auto issue = [&](int st) {
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int c = wave + NW * i;
      if (c < STAGE_COPIES) {
        const long k = (kb + c) % src_kb;
        __builtin_amdgcn_global_load_lds((glb_vp)(src + k * 1024 + lane * 16), (lds_vp)(smem + st * STAGE + c * 1024), 16, 0, 0);
      }
    }
    kb = (kb + STAGE_COPIES) % src_kb;
  };
  unsigned acc = 0;
  issue(0);
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    if (it + 1 < iters) issue((it + 1) & 1);
    // read the current stage like the GEMM's transposing fragment loader does (ds_read_b64_tr_b16), a few KB per wave
    const char* St = smem + (it & 1) * STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      typedef __attribute__((address_space(3))) bf16x4* lds_p;
      const int off = ((wave * 4 + j) * 2048 + (lane & 15) * 64 + (lane >> 4) * 8) % (STAGE - 512);
      const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(St + (off & ~7)));
      acc += (unsigned)__builtin_bit_cast(unsigned short, v[0]) + (unsigned)__builtin_bit_cast(unsigned short, v[3]);
    }
    __syncthreads();
  }
  if (TAIL) issue(iters & 1);          // copies in flight at s_endpgm: nothing waits for them
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int NW, int STAGE_COPIES>
__global__ __launch_bounds__(64 * NW) void regstage_kernel(const char* __restrict__ src, long src_kb, int iters, unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = STAGE_COPIES * 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int CPW = (STAGE_COPIES + NW - 1) / NW;
  long kb = ((long)blockIdx.x * 7919) % src_kb;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const int st = it & 1;
    for (int i = 0; i < CPW; ++i) {
      const int c = wave + NW * i;
      if (c < STAGE_COPIES) {
        const long k = (kb + c) % src_kb;
        const uint4 v = *reinterpret_cast<const uint4*>(src + k * 1024 + lane * 16);
        *reinterpret_cast<uint4*>(smem + st * STAGE + c * 1024 + lane * 16) = v;
      }
    }
    kb = (kb + STAGE_COPIES) % src_kb;
    __syncthreads();
    acc += *reinterpret_cast<const unsigned*>(smem + st * STAGE + ((tid * 52) % STAGE & ~3));
    __syncthreads();
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// ------------------------------------------------------------------------------------------------ victims
__device__ __forceinline__ unsigned mix(unsigned a, unsigned b, unsigned c) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (c + 0x165667B1u) * 0xC2B2AE35u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
  return h | 1u;
}
constexpr int CAN_WORDS = 5120;      // 20 KB
__global__ __launch_bounds__(256) void canary_kernel(int iters, int delay, unsigned salt, Report* rep) {
  __shared__ __attribute__((aligned(16))) unsigned lds[CAN_WORDS];
  const int tid = threadIdx.x;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned key = salt + it * 977u + blockIdx.x * 131071u;
    for (int w = tid * 4; w < CAN_WORDS; w += 1024) {
      const uint4 v = make_uint4(mix(key, w, 0), mix(key, w + 1, 0), mix(key, w + 2, 0), mix(key, w + 3, 0));
      *reinterpret_cast<uint4*>(lds + w) = v;
    }
    __syncthreads();
    for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(32);
    const int rt = (tid + 64) & 255;             // read what the NEXT wave wrote
    for (int w = rt * 4; w < CAN_WORDS; w += 1024) {
      const uint4 v = *reinterpret_cast<const uint4*>(lds + w);
      const unsigned got[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned want = mix(key, w + j, 0);
        if (got[j] != want) {
          if (atomicAdd(&rep->mism, 1u) == 0u) {
            rep->first[0] = blockIdx.x; rep->first[1] = it; rep->first[2] = w + j; rep->first[3] = got[j]; rep->first[4] = want;
            rep->first[5] = tid & 63;
          }
          ++bad;
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0) atomicAdd(&rep->iters, (unsigned)iters);
  if (bad == 0xFFFFFFFFu) rep->first[0] = bad;
}

// the LDS phase of the first reduce_dact_onehot_kernel (git f933989, score.hip:430-470), k = 4 (minute table: nk = 61 rows)
__global__ __launch_bounds__(256) void oldreduce_kernel(int iters, int M, const float* __restrict__ tclip, const float* __restrict__ coef,
                                                        Report* rep) {
  __shared__ float dpl[16 * 64];
  __shared__ __attribute__((aligned(16))) float tl[61 * 64];
  const int tid = threadIdx.x, cg = tid & 15, rp = tid >> 4;
  constexpr int nk = 61;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const int r0 = ((blockIdx.x + it * 37) % (M / 16)) * 16, row = r0 + rp;
    for (int i = tid; i < nk * 16; i += 256) *reinterpret_cast<float4*>(tl + i * 4) = *reinterpret_cast<const float4*>(tclip + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cg + 16 * j;
      float v = 0.f;
      if (c < nk) v = coef[(long)row * 64 + c];
      dpl[rp * 64 + c] = v;
    }
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < nk; ++r) {
      const float4 t = *reinterpret_cast<const float4*>(tl + r * 64 + cg * 4);
      const float s = dpl[rp * 64 + r];
      acc.x = fmaf(t.x, s, acc.x); acc.y = fmaf(t.y, s, acc.y); acc.z = fmaf(t.z, s, acc.z); acc.w = fmaf(t.w, s, acc.w);
    }
    // the same sum without LDS (global / L1 reads), same order, same fmaf chain
    float4 ref = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < nk; ++r) {
      const float4 t = *reinterpret_cast<const float4*>(tclip + r * 64 + cg * 4);
      const float s = coef[(long)row * 64 + r];
      ref.x = fmaf(t.x, s, ref.x); ref.y = fmaf(t.y, s, ref.y); ref.z = fmaf(t.z, s, ref.z); ref.w = fmaf(t.w, s, ref.w);
    }
    const float g[4] = {acc.x, acc.y, acc.z, acc.w}, w[4] = {ref.x, ref.y, ref.z, ref.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (__float_as_uint(g[j]) != __float_as_uint(w[j])) {
        if (atomicAdd(&rep->mism, 1u) == 0u) {
          rep->first[0] = blockIdx.x; rep->first[1] = it; rep->first[2] = j; rep->first[3] = __float_as_uint(g[j]);
          rep->first[4] = __float_as_uint(w[j]); rep->first[5] = tid;
        }
        ++bad;
      }
    __syncthreads();
  }
  if (tid == 0) atomicAdd(&rep->iters, (unsigned)iters);
  if (bad == 0xFFFFFFFFu) rep->first[0] = bad;
}

// ------------------------------------------------------------------------------------------------ driver
int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 0.6;       // per experiment
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const long src_kb = 256 * 1024;            // 256 MB of source bytes: the fills miss L2
  char* src; CK(hipMalloc(&src, src_kb * 1024));
  {
    std::vector<unsigned> h(src_kb * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u) | 0x80008000u;      // never equals a canary word pattern by design? (checked by value)
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  unsigned* sink; CK(hipMalloc(&sink, 64)); CK(hipMemset(sink, 0, 64));
  Report* rep; CK(hipMalloc(&rep, sizeof(Report)));
  const int M = 512;
  float *tclip, *coef;
  CK(hipMalloc(&tclip, 61 * 64 * 4)); CK(hipMalloc(&coef, (size_t)M * 64 * 4));
  {
    std::vector<float> a(61 * 64), b((size_t)M * 64);
    unsigned x = 12345;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : a) v = rnd();
    for (auto& v : b) v = rnd() * 1e-3f;
    CK(hipMemcpy(tclip, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(coef, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipFuncSetAttribute((const void*)dma_kernel<8, 80, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)dma_kernel<9, 24, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  CK(hipFuncSetAttribute((const void*)dma_kernel<9, 24, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  CK(hipFuncSetAttribute((const void*)regstage_kernel<9, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));

  const char* aggr[] = {"none", "dma9", "dma_tail", "dma8big", "regstage"};
  const char* vict[] = {"canary", "canary_delay", "oldreduce"};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int a = 0; a < 5; ++a)
    for (int v = 0; v < 3; ++v) {
      CK(hipMemset(rep, 0, sizeof(Report)));
      CK(hipDeviceSynchronize());
      // the aggressor stream is kept busy for as long as victim launches are in flight: it is topped up whenever its last chunk has drained
      int launches = 0, alaunches = 0;
      CK(hipEventRecord(e0, s2));
      const int chunk = 100;
      double elapsed = 0;
      unsigned salt = 1;
      hipEvent_t ea, ev; CK(hipEventCreate(&ea)); CK(hipEventCreate(&ev));
      bool a_pending = false;
      auto launch_aggr = [&]() {
        for (int i = 0; i < chunk; ++i) {
          if (a == 1) hipLaunchKernelGGL((dma_kernel<9, 24, false>), dim3(720), dim3(576), 2 * 24 * 1024, s1, src, src_kb, 16, sink);
          if (a == 2) hipLaunchKernelGGL((dma_kernel<9, 24, true>), dim3(720), dim3(576), 2 * 24 * 1024, s1, src, src_kb, 16, sink);
          if (a == 3) hipLaunchKernelGGL((dma_kernel<8, 80, false>), dim3(240), dim3(512), 2 * 80 * 1024, s1, src, src_kb, 26, sink);
          if (a == 4) hipLaunchKernelGGL((regstage_kernel<9, 24>), dim3(720), dim3(576), 2 * 24 * 1024, s1, src, src_kb, 16, sink);
        }
        alaunches += chunk;
        CK(hipEventRecord(ea, s1));
        a_pending = true;
      };
      while (elapsed < seconds * 1e3) {
        if (a != 0) launch_aggr();
        for (int i = 0; i < chunk; ++i) {
          if (v == 0) hipLaunchKernelGGL(canary_kernel, dim3(416), dim3(256), 0, s2, 6, 0, salt++, rep);
          if (v == 1) hipLaunchKernelGGL(canary_kernel, dim3(416), dim3(256), 0, s2, 3, 12, salt++, rep);
          if (v == 2) hipLaunchKernelGGL(oldreduce_kernel, dim3(416), dim3(256), 0, s2, 8, M, tclip, coef, rep);
        }
        launches += chunk;
        CK(hipEventRecord(ev, s2));
        while (hipEventQuery(ev) == hipErrorNotReady) {
          if (a != 0 && a_pending && hipEventQuery(ea) == hipSuccess) launch_aggr();
        }
        CK(hipEventRecord(e1, s2));
        CK(hipStreamSynchronize(s2));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        elapsed = ms;
      }
      CK(hipStreamSynchronize(s1));
      CK(hipEventDestroy(ea)); CK(hipEventDestroy(ev));
      (void)alaunches;
      CK(hipGetLastError());
      Report h; CK(hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost));
      printf("%-9s | %-12s | victim launches %6d (aggressor %6d)  victim wg-iterations %9u  %.0f ms | mismatches %u", aggr[a], vict[v], launches, alaunches, h.iters, elapsed, h.mism);
      if (h.mism) printf("  first: wg %u iter %u word/comp %u got %08x want %08x lane/tid %u", h.first[0], h.first[1], h.first[2], h.first[3], h.first[4], h.first[5]);
      printf("\n");
      fflush(stdout);
    }
  return 0;
}
