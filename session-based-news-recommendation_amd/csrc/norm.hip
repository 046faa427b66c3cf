// Optional op: `normalize` of modules.py:194-218 (layer normalisation over the last axis) for gfx950.  NOT on TCAR's executed
// graph (its two call sites, modules.py:301 and :335, are commented out in the reference); built as an op of its own for the
// transformer-style blocks of modules.py:220-336 (multihead_attention / feedforward), parity against an fp64 restatement.
//     mean, variance = moments(x, axis -1)         (biased variance, tf.nn.moments)          modules.py:213
//     y = gamma * (x - mean) / (variance + eps)^0.5 + beta                                   modules.py:216-217
// HBM-bound: one 64-lane wave owns one row, the row is read once (kept in registers for C <= 2048), written once.
#include "tcar_common.h"

namespace {

constexpr int LN_MAXJ = 32;      // columns per lane: C <= 64 * 32

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(long M, int C, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, float* __restrict__ y, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int nj = (C + 63) >> 6;
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += (long)gridDim.x * 4) {
    const float* xr = x + row * C;
    float v[LN_MAXJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
      const int c = j * 64 + lane;
      v[j] = (j < nj && c < C) ? xr[c] : 0.f;
      s += v[j];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
      const int c = j * 64 + lane;
      const float d = (j < nj && c < C) ? v[j] - mean : 0.f;
      q = fmaf(d, d, q);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    float* yr = y + row * C;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
      const int c = j * 64 + lane;
      if (j < nj && c < C) yr[c] = fmaf(gamma[c], (v[j] - mean) * rstd, beta[c]);
    }
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
  }
}

// dx = rstd * (dn - mean(dn) - n * mean(dn * n)), dn = dy * gamma, n = (x - mean) * rstd;  dgamma += dy * n, dbeta += dy
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(long M, int C, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ stats,
                                                            const float* __restrict__ dy, float* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int lane = threadIdx.x & 63;
  const int nj = (C + 63) >> 6;
  float gg[LN_MAXJ], gb[LN_MAXJ];
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) { gg[j] = 0.f; gb[j] = 0.f; }
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += (long)gridDim.x * 4) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float* xr = x + row * C;
    const float* gr = dy + row * C;
    float n[LN_MAXJ], dn[LN_MAXJ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
      const int c = j * 64 + lane;
      const bool ok = j < nj && c < C;
      const float g = ok ? gr[c] : 0.f;
      n[j] = ok ? (xr[c] - mean) * rstd : 0.f;
      dn[j] = ok ? g * gamma[c] : 0.f;
      s1 += dn[j];
      s2 = fmaf(dn[j], n[j], s2);
      gg[j] = fmaf(g, n[j], gg[j]);
      gb[j] += g;
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
    float* dr = dx + row * C;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
      const int c = j * 64 + lane;
      if (j < nj && c < C) dr[c] = rstd * (dn[j] - m1 - n[j] * m2);
    }
  }
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    const int c = j * 64 + lane;
    if (j < nj && c < C) {
      if (gg[j] != 0.f) atomicAdd(dgamma + c, gg[j]);
      if (gb[j] != 0.f) atomicAdd(dbeta + c, gb[j]);
    }
  }
}

}  // namespace

extern "C" int tcar_layernorm_fwd(int64_t M, int C, const float* x, const float* gamma, const float* beta, float eps, float* y,
                                  float* stats, void* stream) {
  if (M <= 0) return TCAR_OK;
  if (!x || !gamma || !beta || !y || !stats || C <= 0 || C > 64 * LN_MAXJ) return TCAR_E_ARG;
  long grid = (M + 3) / 4;
  if (grid > 4096) grid = 4096;
  TCAR_LAUNCH(layernorm_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (long)M, C, x, gamma, beta, eps, y, stats);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

extern "C" int tcar_layernorm_bwd(int64_t M, int C, const float* x, const float* gamma, const float* stats, const float* dy,
                                  float* dx, float* dgamma, float* dbeta, void* stream) {
  if (M <= 0) return TCAR_OK;
  if (!x || !gamma || !stats || !dy || !dx || !dgamma || !dbeta || C <= 0 || C > 64 * LN_MAXJ) return TCAR_E_ARG;
  long grid = (M + 3) / 4;
  if (grid > 512) grid = 512;          // each wave keeps its column sums in registers over its rows: one atomic per column and wave
  TCAR_LAUNCH(layernorm_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (long)M, C, x, gamma, stats, dy, dx,
              dgamma, dbeta);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}
