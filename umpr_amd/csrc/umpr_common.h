// Shared device/host helpers for libumpr_hip (gfx950 / MI355X only).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error reporting (C ABI: 0 ok, <0 error, message via umpr_last_error()) -------------------
void umpr_set_error(const char* fmt, ...);
#define UMPR_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      umpr_set_error(__VA_ARGS__);         \
      return -1;                           \
    }                                      \
  } while (0)
#define UMPR_LAUNCH_CHECK(name)                                                     \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) {                                                        \
      umpr_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return -2;                                                                    \
    }                                                                               \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Environment switches.  Plain functions on purpose: two namespace-scope `[] { getenv("A") ... }()` initialisers that differ
// only in their string literal were given the same closure symbol by hipcc in api.hip, and the second switch silently read the
// first one's variable (UMPR_WGRAD_STREAM=0 had no effect for half a round; tests/test_cabi_symbols.py now checks that every
// getenv name of the sources is present in the built library).
static inline bool umpr_env_on(const char* name) { const char* v = getenv(name); return !(v && v[0] == '0'); }
static inline int umpr_env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

// ---- device helpers --------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32: A lane l -> A[i=l&31][k=l>>5], B lane l -> B[k=l>>5][j=l&31],
// D reg r of lane l -> D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]   (cdna_hip_programming.md section 3)
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// epilogue activation codes shared by GEMM / conv
enum { UMPR_ACT_NONE = 0, UMPR_ACT_RELU = 1, UMPR_ACT_TANH = 2, UMPR_ACT_SIGMOID = 3 };
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == UMPR_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == UMPR_ACT_TANH) return tanhf(v);
  if (act == UMPR_ACT_SIGMOID) return sigmoidf_(v);
  return v;
}
