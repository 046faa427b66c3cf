// What does an event record / wait between two kernels of one stream cost on the device?  Pairs of kernels (spin_a -> spin_b_<variant>)
// are launched back to back; the variants put different things between them.  Run under rocprofv3 --kernel-trace and read the gap
// between the end of spin_a and the start of the variant's kernel (tools/micro/event_cost.py).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ void spin(long n, float* out) {
  float v = threadIdx.x;
  for (long i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
  if (v == 123.456f) out[0] = v;
}
__global__ void spin_a(long n, float* out) { spin(n, out); }
#define VARIANT(name) __global__ void name(long n, float* out) { spin(n, out); }
VARIANT(b_plain) VARIANT(b_record) VARIANT(b_ext_stop) VARIANT(b_wait_done) VARIANT(b_record_wait_other) VARIANT(b_two_waits)
VARIANT(c_other) VARIANT(b_wait_live) VARIANT(long_first) VARIANT(b_ahead_done) VARIANT(b_ahead_live) VARIANT(b_flag_join)
VARIANT(b_ahead_record) VARIANT(b_flag_fork) VARIANT(c_flag_consumer)

// device-side join: the last workgroup of the producer publishes an epoch; a one-wave kernel on the consumer stream polls it
__global__ void c_signalling(long n, float* out, unsigned* cnt, unsigned* flag, unsigned epoch) {
  spin(n, out);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(cnt, 1u) == gridDim.x - 1) {
      *cnt = 0;
      __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
__global__ void poll_flag(const unsigned* flag, unsigned epoch) {
  if (threadIdx.x == 0) {
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) __builtin_amdgcn_s_sleep(2);
  }
}

int main() {
  float* out; CK(hipMalloc(&out, 64));
  hipStream_t s1, s2, s3;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
  hipEvent_t e1, e2, e3, e4;
  CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&e3, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e4, hipEventDisableTiming));
  const long n = 4000;   // ~10 us
  const dim3 g(256), b(256);
  for (int rep = 0; rep < 30; ++rep) {
    // plain
    hipLaunchKernelGGL(spin_a, g, b, 0, s1, n, out); hipLaunchKernelGGL(b_plain, g, b, 0, s1, n, out);
    CK(hipDeviceSynchronize());
    // record between (another stream waits for it and runs something)
    hipLaunchKernelGGL(spin_a, g, b, 0, s1, n, out);
    CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s2, e1, 0));
    hipLaunchKernelGGL(c_other, g, b, 0, s2, n, out);
    hipLaunchKernelGGL(b_record, g, b, 0, s1, n, out);
    CK(hipDeviceSynchronize());
    // the same fork through the producer's own completion signal
    hipExtLaunchKernelGGL(spin_a, g, b, 0, s1, nullptr, e1, 0, n, out);
    CK(hipStreamWaitEvent(s2, e1, 0));
    hipLaunchKernelGGL(c_other, g, b, 0, s2, n, out);
    hipLaunchKernelGGL(b_ext_stop, g, b, 0, s1, n, out);
    CK(hipDeviceSynchronize());
    // wait for an event of another stream that completed long ago
    hipLaunchKernelGGL(c_other, g, b, 0, s2, n, out); CK(hipEventRecord(e2, s2));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(spin_a, g, b, 0, s1, n, out);
    CK(hipStreamWaitEvent(s1, e2, 0));
    hipLaunchKernelGGL(b_wait_done, g, b, 0, s1, n, out);
    CK(hipDeviceSynchronize());
    // two waits for completed events
    hipLaunchKernelGGL(c_other, g, b, 0, s2, n, out); CK(hipEventRecord(e2, s2));
    hipLaunchKernelGGL(c_other, g, b, 0, s3, n, out); CK(hipEventRecord(e3, s3));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(spin_a, g, b, 0, s1, n, out);
    CK(hipStreamWaitEvent(s1, e2, 0)); CK(hipStreamWaitEvent(s1, e3, 0));
    hipLaunchKernelGGL(b_two_waits, g, b, 0, s1, n, out);
    CK(hipDeviceSynchronize());
    // wait for an event of another stream whose kernel ends at about the same time as spin_a (live join)
    hipLaunchKernelGGL(spin_a, g, b, 0, s1, n, out);
    hipLaunchKernelGGL(c_other, g, b, 0, s2, n, out); CK(hipEventRecord(e4, s2));
    CK(hipStreamWaitEvent(s1, e4, 0));
    hipLaunchKernelGGL(b_wait_live, g, b, 0, s1, n, out);
    CK(hipDeviceSynchronize());
  }
  unsigned* cnt; unsigned* flag; CK(hipMalloc(&cnt, 8)); CK(hipMemset(cnt, 0, 8)); flag = cnt + 1;
  const long nl = 80000;  // ~200 us: the host runs ahead of the device behind it
  unsigned epoch = 0;
  for (int rep = 0; rep < 30; ++rep) {
    // host ahead, event long complete when the device reaches the wait
    hipLaunchKernelGGL(long_first, g, b, 0, s1, nl, out);
    hipLaunchKernelGGL(c_other, g, b, 0, s2, n, out); CK(hipEventRecord(e2, s2));
    hipLaunchKernelGGL(spin_a, g, b, 0, s1, n, out);
    CK(hipStreamWaitEvent(s1, e2, 0));
    hipLaunchKernelGGL(b_ahead_done, g, b, 0, s1, n, out);
    CK(hipDeviceSynchronize());
    // host ahead, record between
    hipLaunchKernelGGL(long_first, g, b, 0, s1, nl, out);
    hipLaunchKernelGGL(spin_a, g, b, 0, s1, n, out);
    CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s2, e1, 0));
    hipLaunchKernelGGL(c_other, g, b, 0, s2, n, out);
    hipLaunchKernelGGL(b_ahead_record, g, b, 0, s1, n, out);
    CK(hipDeviceSynchronize());
    // host ahead, live join: s2's kernel is forked from s1 right before spin_a's predecessor ends, runs as long as spin_a
    hipLaunchKernelGGL(long_first, g, b, 0, s1, nl, out);
    CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s2, e1, 0));
    hipLaunchKernelGGL(c_other, g, b, 0, s2, n + n / 2, out); CK(hipEventRecord(e4, s2));
    hipLaunchKernelGGL(spin_a, g, b, 0, s1, n + n, out);
    CK(hipStreamWaitEvent(s1, e4, 0));
    hipLaunchKernelGGL(b_ahead_live, g, b, 0, s1, n, out);
    CK(hipDeviceSynchronize());
    // the same join through a device flag (no event on either side of the join)
    ++epoch;
    hipLaunchKernelGGL(long_first, g, b, 0, s1, nl, out);
    CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s2, e1, 0));
    hipLaunchKernelGGL(c_signalling, g, b, 0, s2, n + n / 2, out, cnt, flag, epoch);
    hipLaunchKernelGGL(spin_a, g, b, 0, s1, n + n, out);
    hipLaunchKernelGGL(poll_flag, dim3(1), dim3(64), 0, s1, flag, epoch);
    hipLaunchKernelGGL(b_flag_join, g, b, 0, s1, n, out);
    CK(hipDeviceSynchronize());
    // fork through a device flag: s2's consumer was enqueued EARLY behind a polling kernel; spin_a (signalling) releases it
    ++epoch;
    hipLaunchKernelGGL(poll_flag, dim3(1), dim3(64), 0, s2, flag, epoch);
    hipLaunchKernelGGL(c_flag_consumer, g, b, 0, s2, n, out);
    hipLaunchKernelGGL(long_first, g, b, 0, s1, nl, out);
    hipLaunchKernelGGL(c_signalling, g, b, 0, s1, n, out, cnt, flag, epoch);      // plays spin_a
    hipLaunchKernelGGL(b_flag_fork, g, b, 0, s1, n, out);
    CK(hipDeviceSynchronize());
  }
  printf("done\n");
  return 0;
}
