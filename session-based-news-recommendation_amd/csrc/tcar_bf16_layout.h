// "KB32" blocked layout of the bf16 operand planes (hi / lo) used by gemm_bf16.hip.
//
// A logical matrix X[rows][inner] (inner % 32 == 0; rows padded to a multiple of 128, padding rows zero) is stored as
// blocks of 128 rows x 32 inner elements, 8 KB each, block (rb, kb) at element offset (rb * inner/32 + kb) * 4096.
// Inside a block, row r occupies 64 consecutive bytes = four 16-byte pieces; piece c is stored at position
// c ^ ((r >> 2) & 3).  Consequences, all by construction of this one layout:
//   * a GEMM stage of a k-contiguous operand tile (128 rows x 32 k) IS one block: a single contiguous 8-KB chunk,
//     copied global -> LDS linearly (direct-to-LDS DMA, no per-lane address arithmetic, full 128-byte lines);
//   * the XOR makes the 16-lane groups of ds_read_b128 fragment reads hit 16 distinct 16-byte bank slots although the
//     LDS image is unpadded (64-byte rows);
//   * a stage of an m/n-contiguous ("transposed-read") operand tile (32 k-rows x 128 cols) is four contiguous 2-KB
//     chunks (32 consecutive rows of four neighbouring blocks); ds_read_b64_tr_b16 on 64-byte rows is conflict free
//     (4 k-rows x 2 column groups x 4 pieces = 32 distinct 8-byte slots of the 256-byte bank row).
// The planes are private buffers of this library, so the layout costs nothing: their producers (tcar_split_bf16, the
// Adam and candidate-time kernels, the softmax gradient) write it directly.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KB32_HD __host__ __device__ __forceinline__
#else
#define KB32_HD inline
#endif

// element offset of (row, k) — k multiple of 4 addresses 4 consecutive elements (8 bytes) contiguously
KB32_HD long kb32_off(long row, int k, int inner32) {
  const long blk = (row >> 7) * inner32 + (k >> 5);
  const int r = (int)(row & 127), kk = k & 31;
  return blk * 4096 + r * 32 + ((((kk >> 3) ^ ((r >> 2) & 3)) << 3) | (kk & 7));
}
