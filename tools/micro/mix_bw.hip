// What does HBM give a kernel that READS r bytes and WRITES w bytes in the embedding gather's ratio (1.36 GB in : 2.36 GB out per
// launch, profiles/r05_pmc_gather_fwd.json) when both sides are perfectly sequential?  The upper bound for the gather's real-bytes
// figure: its reads are random 1-KB rows on top of this.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mix_bw tools/micro/mix_bw.hip && tools/micro/mix_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
// every thread: per trip RD float4 loads (sequential, coalesced) and WR float4 stores (non-temporal or plain)
template <int RD, int WR, bool NT>
__global__ __launch_bounds__(1024) void mix(const f4* __restrict__ in, f4* __restrict__ out, long trips) {
  const long nthr = (long)gridDim.x * blockDim.x, tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long t = 0; t < trips; ++t) {
    f4 v[RD > 0 ? RD : 1];
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < RD; ++i) v[i] = __builtin_nontemporal_load(in + (t * RD + i) * nthr + tid);
#pragma unroll
    for (int i = 0; i < RD; ++i) acc += v[i];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      f4 o = acc; o.x += (float)i;
      if (NT) __builtin_nontemporal_store(o, out + (t * WR + i) * nthr + tid);
      else out[(t * WR + i) * nthr + tid] = o;
    }
    if (WR == 0 && acc.x == 123456.f) out[tid] = acc;          // (read-only form: keep the loads alive)
  }
}
template <int RD, int WR, bool NT>
int run(const char* name, f4* in, f4* out, int wg_per_cu) {
  const int grid = 256 * wg_per_cu, block = 1024;
  const long nthr = (long)grid * block;
  const long total_f4 = (long)3700 * 1000 * 1000 / 16;                 // ~3.7 GB per launch, as the gather
  const long trips = total_f4 / (nthr * (RD + WR));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mix<RD, WR, NT>), dim3(grid), dim3(block), 0, 0, in, out, trips);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mix<RD, WR, NT>), dim3(grid), dim3(block), 0, 0, in, out, trips);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double rb = (double)trips * RD * nthr * 16, wb = (double)trips * WR * nthr * 16;
  printf("%-34s %d WG/CU: read %.2f GB + write %.2f GB in %.3f ms = %.2f TB/s (%.3f of 8)\n", name, wg_per_cu, rb / 1e9, wb / 1e9, ms,
         (rb + wb) / ms / 1e9, (rb + wb) / ms / 1e9 / 8.0);
  return 0;
}
int main() {
  f4 *in, *out;
  const size_t bytes = (size_t)4 << 30;
  CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
  CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
  // (block = 1024 threads; 1 and 2 workgroups per CU.  Round 5, one box: the gather's mix 5.28 TB/s = 0.660 of 8, write only 5.49,
  //  copy 5.44 with one workgroup per CU; two per CU 5-8 % lower — the gather kernel itself moves its real bytes at 5.43 TB/s)
  for (int w = 1; w <= 2; ++w) {
    if (run<4, 7, true>("read 4 : write 7 (the gather's mix), nt", in, out, w)) return 1;
    if (run<4, 7, false>("read 4 : write 7, plain stores", in, out, w)) return 1;
    if (run<8, 0, true>("read only", in, out, w)) return 1;
    if (run<0, 8, true>("write only, nt", in, out, w)) return 1;
    if (run<4, 4, true>("copy 1 : 1, nt", in, out, w)) return 1;
  }
  return 0;
}
