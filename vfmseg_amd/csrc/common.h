// Shared device/host helpers for libvfmseg_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vfmseg_hip.h"

typedef uint16_t bf16_t;  // raw bf16 bits

extern thread_local char g_vfm_err[512];

#define VFM_FAIL(code, ...)                                   \
  do {                                                        \
    snprintf(g_vfm_err, sizeof(g_vfm_err), __VA_ARGS__);      \
    return (code);                                            \
  } while (0)

#define VFM_CHECK(cond, code, ...) \
  do {                             \
    if (!(cond)) VFM_FAIL(code, __VA_ARGS__); \
  } while (0)

#define VFM_LAUNCH_CHECK()                                                                   \
  do {                                                                                       \
    hipError_t e_ = hipGetLastError();                                                       \
    if (e_ != hipSuccess) VFM_FAIL(VFM_E_HIP, "%s:%d launch: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
  } while (0)

static inline int vfm_dtype_size(int dt) { return dt == VFM_BF16 ? 2 : (dt == VFM_F32 ? 4 : (dt == VFM_U8 ? 1 : 8)); }

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}

template <typename T>
__device__ __forceinline__ float ld_f32(const T* p);
template <>
__device__ __forceinline__ float ld_f32<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld_f32<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

template <typename T>
__device__ __forceinline__ void st_f32(T* p, float v);
template <>
__device__ __forceinline__ void st_f32<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void st_f32<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// dtype-erased element access for epilogues (dt is wave-uniform)
__device__ __forceinline__ float ld_any(const void* p, long idx, int dt) {
  return dt == VFM_BF16 ? bf16_to_f32(((const bf16_t*)p)[idx]) : ((const float*)p)[idx];
}
__device__ __forceinline__ void st_any(void* p, long idx, int dt, float v) {
  if (dt == VFM_BF16) ((bf16_t*)p)[idx] = f32_to_bf16(v);
  else ((float*)p)[idx] = v;
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// CLIP QuickGELU (clip.py:18-20): x * sigmoid(1.702 x) and its derivative
__device__ __forceinline__ float qgelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float qgelu_grad_f(float x) {
  const float sg = 1.0f / (1.0f + __expf(-1.702f * x));
  return sg * (1.0f + 1.702f * x * (1.0f - sg));
}

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

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `sh` needs 16 floats. Result valid in all threads.
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}

// counter-based RNG (splitmix-style hash): deterministic per (seed, index), no state
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
