// Shared helpers for librspnet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rspnet_hip.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define RSP_WAVE 64

void rsp_set_error(const char* msg);
void rsp_note_reset(void);                    // arm the per-call kernel-name note (errors.hip)
void rsp_note_kernel(const char* fmt, ...);    // first matrix kernel launched since the last reset

static inline int rsp_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    rsp_set_error(buf);
    return RSP_ELAUNCH;
  }
  return RSP_OK;
}

#define RSP_REQUIRE(cond, msg)     \
  do {                             \
    if (!(cond)) {                 \
      rsp_set_error(msg);          \
      return RSP_EINVAL;           \
    }                              \
  } while (0)

static inline size_t rsp_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int rsp_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline bool rsp_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Bijective XCD-aware block remap (cdna_hip_programming.md §5 "XCD swizzle must be bijective"):
// blocks b and b+8 share an XCD; give each XCD a contiguous chunk of tile ids so neighbouring tiles
// (which share input halos / weight panels) hit the same 4 MiB L2.
__device__ __forceinline__ int rsp_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

__device__ __forceinline__ float rsp_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double rsp_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float rsp_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Division by a launch constant as multiply-high + shift (the prologue of every tile decodes 4 GEMM rows per thread into
// (n, d, h, w) and its k position into (tap, channel): with hardware-free integer division that was ~1000 instructions
// per wave).  Exact for 0 <= n < 2^31: mul = ceil(2^(31+s) / d), s = ceil(log2 d); q = umulhi(n, mul) >> (s - 1).
struct FastDiv {
  unsigned mul;   // 0: divisor 1
  unsigned shr;
  int d;
};

static inline FastDiv fastdiv_make(int d) {
  FastDiv f = {0u, 0u, d};
  if (d <= 1) return f;
  int s = 0;
  while ((1ll << s) < d) ++s;
  const unsigned long long pw = 1ull << (31 + s);
  f.mul = (unsigned)((pw + (unsigned long long)d - 1) / (unsigned long long)d);
  f.shr = (unsigned)(s - 1);
  return f;
}

__device__ __forceinline__ int fastdiv(int n, const FastDiv f) {
  return f.mul ? (int)(__umulhi((unsigned)n, f.mul) >> f.shr) : n;
}
