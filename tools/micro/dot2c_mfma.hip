// v_dot2c_f32_bf16 beside MFMAs (round 6).  In the logits GEMM's epilogue `sum = dot2c(packed bf16 pair, (1, 1), sum)` gave wrong sums; this
// is the bare instruction, exact by construction — eight pairs of small dyadic values per chain, every partial sum representable in
// fp32 whatever the rounding order —, checked bit for bit against scalar adds, alone and beside a dense v_mfma_f32_32x32x16_bf16 loop
// on a second stream, in three victim shapes: plain chains, chains with 200 accumulator registers alive (the epilogue's pressure), and the
// literal-constant form the compiler emitted there (v_dot2c_f32_bf16_e32 v, 0x3f803f80, v).  Result: 0 wrong — the instruction is fine; the
// kernel's wrong sums were a hipcc miscompile of the bit-cast operand (docs/EXPERIMENTS.md, profiles/r06_dot2c_in_step.txt).
//   hipcc --offload-arch=gfx950 -O3 -o dot2c_mfma dot2c_mfma.hip ; ./dot2c_mfma [seconds per cell]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
struct Report { unsigned mism, iters, by_lane16[4]; };

template <int FORM>
__global__ __launch_bounds__(256) void victim(int iters, Report* rep) {
  unsigned seed = threadIdx.x * 977u + blockIdx.x * 131071u + 12345u;
  float keep[FORM == 1 ? 200 : 1];
  if (FORM == 1)
    for (int i = 0; i < 200; ++i) keep[i] = (float)(threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
    float acc = 0.f, want = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      seed = seed * 1664525u + 1013904223u;
      const float x = 1.0f + ((seed >> 9) & 127) / 128.0f, y = 1.0f + ((seed >> 17) & 127) / 128.0f;      // exact in bf16
      const bf16x2 p = {(__bf16)x, (__bf16)y};
      if (FORM == 2) asm volatile("v_dot2c_f32_bf16 %0, 0x3f803f80, %1" : "+v"(acc) : "v"(p));
      else { const bf16x2 one = {(__bf16)1.0f, (__bf16)1.0f}; acc = __builtin_amdgcn_fdot2_f32_bf16(p, one, acc, false); }
      want += x; want += y;
      if (FORM == 1) keep[(q * 25 + it) % 200] += acc;
    }
    if (__float_as_uint(acc) != __float_as_uint(want)) {
      atomicAdd(&rep->mism, 1u);
      atomicAdd(&rep->by_lane16[(threadIdx.x & 63) >> 4], 1u);
    }
  }
  if (FORM == 1) { float t = 0.f; for (int i = 0; i < 200; ++i) t += keep[i]; if (t == 12345.678f) rep->mism = 0xFFFFFFFFu; }
  if (threadIdx.x == 0) atomicAdd(&rep->iters, (unsigned)iters);
}

__global__ __launch_bounds__(512) void mfma_kernel(int iters, unsigned* __restrict__ sink) {
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x ^ j)); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, c3, 0, 0, 0);
  }
  if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678f) sink[0] = 1;
}

template <int FORM>
void cell(const char* name, double seconds, hipStream_t s1, hipStream_t s2, Report* rep, unsigned* sink) {
  for (int aggr = 0; aggr < 2; ++aggr) {
    CK(hipMemset(rep, 0, sizeof(Report)));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1, ea, ev;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&ev));
    CK(hipEventRecord(e0, s2));
    double elapsed = 0;
    bool pending = false;
    auto la = [&]() { for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(mfma_kernel, dim3(512), dim3(512), 0, s1, 400, sink);
                      CK(hipEventRecord(ea, s1)); pending = true; };
    while (elapsed < seconds * 1e3) {
      if (aggr) la();
      for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((victim<FORM>), dim3(1024), dim3(256), 0, s2, 2000, rep);
      CK(hipEventRecord(ev, s2));
      while (hipEventQuery(ev) == hipErrorNotReady) if (aggr && pending && hipEventQuery(ea) == hipSuccess) la();
      CK(hipEventRecord(e1, s2));
      CK(hipStreamSynchronize(s2));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); elapsed = ms;
    }
    CK(hipStreamSynchronize(s1));
    Report h; CK(hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-58s | %-11s | %6.2f G chains | wrong %8u (lanes 0-15 / 16-31 / 32-47 / 48-63: %u / %u / %u / %u)\n", name,
           aggr ? "beside MFMA" : "alone", (double)h.iters * 256 / 1e9, h.mism, h.by_lane16[0], h.by_lane16[1], h.by_lane16[2], h.by_lane16[3]);
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); CK(hipEventDestroy(ea)); CK(hipEventDestroy(ev));
  }
}

int main(int argc, char** argv) {
  const double sec = argc > 1 ? atof(argv[1]) : 0.5;
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  Report* rep; CK(hipMalloc(&rep, sizeof(Report)));
  unsigned* sink; CK(hipMalloc(&sink, 64));
  cell<0>("v_dot2c_f32_bf16, chains of eight (builtin)", sec, s1, s2, rep, sink);
  cell<1>("... with 200 accumulator registers alive", sec, s1, s2, rep, sink);
  cell<2>("... the literal form v_dot2c_f32_bf16 v, 0x3f803f80, v", sec, s1, s2, rep, sink);
  return 0;
}
