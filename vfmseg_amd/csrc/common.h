// Shared device/host helpers for libvfmseg_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vfmseg_hip.h"

// The 16-bit floating type of this build.  The library is compiled twice from the same sources:
//     libvfmseg_hip.so      (default)          bf16 storage, v_mfma_f32_32x32x16_bf16      - the throughput configuration
//     libvfmseg_hip_f16.so  (-DVFM_HALF_F16)   IEEE fp16 storage, v_mfma_f32_32x32x16_f16  - what the reference's `--amp` computes in
//                                              (tools/train.py:87-102 -> mmengine AmpOptimWrapper: fp16 autocast + dynamic loss scale)
// Everything 16-bit goes through the names below (bf16_t = raw bits of "the half type", vfm_h = its arithmetic type, VFM_MFMA16 = its
// 32x32x16 MFMA, VFM_DOT2 = its packed dot product, h16_lo / h16_hi = the two halves of a packed pair, VFM_H_ONE = bits of 1.0), so the two builds differ in nothing
// else; dtype code VFM_BF16 in the C ABI means "the half type of the library that was loaded" (vfm_half_kind() says which).
typedef uint16_t bf16_t;  // raw bits of the half type
#ifdef VFM_HALF_F16
typedef _Float16 vfm_h;
#define VFM_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define VFM_MFMA16S(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)   /* 16 x 16 x 32: same rate, the chip holds a higher clock on it */
#define VFM_DOT2(a, b, acc) __builtin_amdgcn_fdot2(a, b, acc, false)           /* v_dot2_f32_f16 */
#define VFM_H_ONE 0x3C00u
#define VFM_HALF_KIND 1
#else
typedef __bf16 vfm_h;
#define VFM_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define VFM_MFMA16S(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define VFM_DOT2(a, b, acc) __builtin_amdgcn_fdot2_f32_bf16(a, b, acc, false)  /* v_dot2c_f32_bf16 */
#define VFM_H_ONE 0x3F80u
#define VFM_HALF_KIND 0
#endif

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

#ifdef VFM_HALF_F16
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// low / high half of a packed pair -> fp32
__device__ __forceinline__ float h16_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float h16_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
#else
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float h16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#endif
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 / v_cvt_f16_f32 (RNE, NaN-preserving; fp16 overflows to inf like torch's .half())
#ifdef VFM_HALF_F16
  // fp16 only: keep the compiler from folding the producing fma into the conversion (v_fma_mixlo_f16 rounds the exact fma once, the
  // separate instructions round to fp32 first).  Either is a correct rounding, but WHICH one is picked depends on the surrounding code,
  // and then two kernels that are meant to be bit-identical (fused vs separate LayerNorm + dropout) differ in one element of 2^13;
  // "round the fp32 value" is also what torch's .half() of an fp32 result does.
  asm("" : "+v"(f));
#endif
  vfm_h b = (vfm_h)f;
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
