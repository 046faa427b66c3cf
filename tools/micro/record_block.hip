// Does a host call on a stream whose head is a spinning (polling) kernel block the host?  s1: long kernel, then a kernel that sets a
// flag; s2: poll kernel (waits for the flag), a small kernel, then hipEventRecord / hipMemsetAsync / hipLaunchKernel — each timed on
// the host.  If a call takes ~ the long kernel's duration, it waits for the poll kernel.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(long n, float* out) {
  float v = threadIdx.x;
  for (long i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
  if (v == 123.456f) out[0] = v;
}
__global__ void set_flag(unsigned* flag, unsigned epoch) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void poll(const unsigned* flag, unsigned epoch, unsigned* err) {
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64();
  while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > 50000000LL) { atomicAdd(err, 1u); break; }
  }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  float* out; CK(hipMalloc(&out, 64));
  unsigned* flag; CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
  float* buf; CK(hipMalloc(&buf, 4096));
  hipStream_t s1, s2;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const dim3 g(256), b(256);
  for (int rep = 0; rep < 6; ++rep) {
    const unsigned epoch = rep + 1;
    hipLaunchKernelGGL(spin, g, b, 0, s1, 400000L, out);          // ~1 ms
    hipLaunchKernelGGL(set_flag, dim3(1), dim3(64), 0, s1, flag, epoch);
    double t0 = now();
    hipLaunchKernelGGL(poll, dim3(8), dim3(64), 0, s2, (const unsigned*)flag, epoch, flag + 8);
    double t1 = now();
    hipLaunchKernelGGL(spin, g, b, 0, s2, 400L, out);
    double t2 = now();
    CK(hipEventRecord(e, s2));
    double t3 = now();
    CK(hipMemsetAsync(buf, 0, 4096, s2));
    double t4 = now();
    hipLaunchKernelGGL(spin, g, b, 0, s2, 400L, out);
    double t5 = now();
    CK(hipStreamWaitEvent(s1, e, 0));
    double t6 = now();
    hipLaunchKernelGGL(spin, g, b, 0, s1, 400L, out);
    double t7 = now();
    CK(hipDeviceSynchronize());
    double t8 = now();
    printf("rep %d: launch poll %.1f  launch kernel behind poll %.1f  eventRecord %.1f  memsetAsync %.1f  launch %.1f  waitEvent(s1) %.1f  launch(s1) %.1f  sync %.1f us\n",
           rep, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7);
  }
  unsigned h[16]; CK(hipMemcpy(h, flag, 64, hipMemcpyDeviceToHost));
  printf("timeouts %u\n", h[8]);
  return 0;
}
