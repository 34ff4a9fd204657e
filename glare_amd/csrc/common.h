// Shared device/host helpers for the gfx950 kernels of libglare_hip.so.
// Everything here is wave64 / CDNA4 specific on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/glare_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef uint16_t bf16_t;  // raw storage type of a bf16 element in HBM

#define GLARE_WAVE 64

// bf16 <-> f32 on raw bits. Round-to-nearest-even, NaN kept quiet.
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// f32 -> bf16 goes through the gfx950 hardware convert (v_cvt_pk_bf16_f32, round-to-nearest-even)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x / (1.0f + __expf(-x)); }

// wave64 reductions (DPP-free: __shfl_xor lowers to ds_bpermute/DPP as hipcc sees fit)
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

static inline int glare_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GLARE_OK : GLARE_ERR_LAUNCH;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }
