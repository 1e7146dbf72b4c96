// Shared helpers for libftx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ftx.h"

namespace ftx {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return FTX_ELAUNCH;
  }
  return FTX_OK;
}

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8, no more.
inline unsigned grid_for(int64_t work, int block) {
  int64_t g = ceil_div(work, block);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}

#define FTX_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      ftx::set_error(__VA_ARGS__);             \
      return FTX_EINVAL;                       \
    }                                          \
  } while (0)

// FNV-1a over the four int32 words of a coordinate row, folded to 60 bits
// (torchsparse v1.1.0 hash kernel; call sites models/utils.py:19,74-78).
__host__ __device__ inline int64_t fnv_hash4(int32_t x, int32_t y, int32_t z, int32_t b) {
  uint64_t h = 14695981039346656037ULL;
  h ^= (uint32_t)x; h *= 1099511628211ULL;
  h ^= (uint32_t)y; h *= 1099511628211ULL;
  h ^= (uint32_t)z; h *= 1099511628211ULL;
  h ^= (uint32_t)b; h *= 1099511628211ULL;
  h = (h >> 60) ^ (h & 0x0FFFFFFFFFFFFFFFULL);
  return (int64_t)h;
}

// Slot hash for the open-addressing table (murmur3 finaliser).
__device__ inline uint64_t slot_mix(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

constexpr int64_t kEmptyKey = -1;  // hashes are 60-bit non-negative, so -1 never collides

}  // namespace ftx
