// Device helpers shared by the bf16 MFMA GEMM kernels (gemm_bf16.hip, gemm_pp.hip).
#pragma once
#include "gemm_epilogue.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef vfm_h bf16x8 __attribute__((ext_vector_type(8)));

#define BK 64

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "gfx9 vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// vector epilogue on 4 consecutive columns (all pointers/leading dims 16-byte compatible; checked on the host)
__device__ __forceinline__ void epi_store4(const EpiParams& e, long zoff, long m, long n, float4 v) {
  float x[4] = {v.x * e.alpha, v.y * e.alpha, v.z * e.alpha, v.w * e.alpha};
  if (e.bias) {
    const long bn = n % e.bias_mod;  // bias_mod % 4 == 0 on this path
    const float4 b = *reinterpret_cast<const float4*>(e.bias + bn);
    x[0] += b.x, x[1] += b.y, x[2] += b.z, x[3] += b.w;
  }
  if (e.C2) {
    const long o = zoff + m * e.ldc2 + n;
    float y[4] = {x[0], x[1], x[2], x[3]};
    if (e.ep_mode == VFM_EP_GELU_DGELU) {
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = gelu_grad_f(x[i]);
    }
    if (e.c2_dt == VFM_BF16) {
      ushort4 p = {f32_to_bf16(y[0]), f32_to_bf16(y[1]), f32_to_bf16(y[2]), f32_to_bf16(y[3])};
      *reinterpret_cast<ushort4*>((bf16_t*)e.C2 + o) = p;
    } else {
      *reinterpret_cast<float4*>((float*)e.C2 + o) = make_float4(y[0], y[1], y[2], y[3]);
    }
  }
  if (e.ep_mode == VFM_EP_GELU || e.ep_mode == VFM_EP_GELU_DGELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = gelu_f(x[i]);
  } else if (e.ep_mode == VFM_EP_RELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = fmaxf(x[i], 0.f);
  } else if (e.ep_mode == VFM_EP_QGELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = qgelu_f(x[i]);
  } else if (e.ep_mode == VFM_EP_MUL_GELU_GRAD || e.ep_mode == VFM_EP_MUL || e.ep_mode == VFM_EP_MUL_QGELU_GRAD) {
    float a[4];
    const long o = m * e.ld_aux + n;
    if (e.aux_dt == VFM_BF16) {
      const ushort4 p = *reinterpret_cast<const ushort4*>((const bf16_t*)e.aux + o);
      a[0] = bf16_to_f32(p.x), a[1] = bf16_to_f32(p.y), a[2] = bf16_to_f32(p.z), a[3] = bf16_to_f32(p.w);
    } else {
      const float4 p = *reinterpret_cast<const float4*>((const float*)e.aux + o);
      a[0] = p.x, a[1] = p.y, a[2] = p.z, a[3] = p.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      x[i] *= (e.ep_mode == VFM_EP_MUL ? a[i] : (e.ep_mode == VFM_EP_MUL_QGELU_GRAD ? qgelu_grad_f(a[i]) : gelu_grad_f(a[i])));
  }
  if (e.colscale) {
    const float4 s = *reinterpret_cast<const float4*>(e.colscale + n);
    x[0] *= s.x, x[1] *= s.y, x[2] *= s.z, x[3] *= s.w;
  }
  if (e.residual) {
    const long o = zoff + m * e.ldr + n;
    if (e.r_dt == VFM_BF16) {
      const ushort4 p = *reinterpret_cast<const ushort4*>((const bf16_t*)e.residual + o);
      x[0] += bf16_to_f32(p.x), x[1] += bf16_to_f32(p.y), x[2] += bf16_to_f32(p.z), x[3] += bf16_to_f32(p.w);
    } else {
      const float4 p = *reinterpret_cast<const float4*>((const float*)e.residual + o);
      x[0] += p.x, x[1] += p.y, x[2] += p.z, x[3] += p.w;
    }
  }
  const long o = zoff + m * e.ldc + n;
  if (e.c_dt == VFM_BF16) {
    ushort4 p = {f32_to_bf16(x[0]), f32_to_bf16(x[1]), f32_to_bf16(x[2]), f32_to_bf16(x[3])};
    *reinterpret_cast<ushort4*>((bf16_t*)e.C + o) = p;
  } else if (e.c_dt == VFM_SPLIT3) {
    const ushort4 hi = {f32_to_bf16(x[0]), f32_to_bf16(x[1]), f32_to_bf16(x[2]), f32_to_bf16(x[3])};
    const ushort4 lo = {f32_to_bf16(x[0] - bf16_to_f32(hi.x)), f32_to_bf16(x[1] - bf16_to_f32(hi.y)), f32_to_bf16(x[2] - bf16_to_f32(hi.z)),
                        f32_to_bf16(x[3] - bf16_to_f32(hi.w))};
    bf16_t* cp = (bf16_t*)e.C + o;
    *reinterpret_cast<ushort4*>(cp) = hi;
    *reinterpret_cast<ushort4*>(cp + e.c_plane) = hi;
    *reinterpret_cast<ushort4*>(cp + 2 * e.c_plane) = lo;
  } else {
    *reinterpret_cast<float4*>((float*)e.C + o) = make_float4(x[0], x[1], x[2], x[3]);
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Wave-tile epilogues.  A wave owns an (MI*32) x (NI*32) fp32 accumulator tile in MFMA layout; it is moved through a
// per-wave LDS image (GROUP 32-row slabs at a time, row stride NI*32+4 floats) so that global traffic is 16-byte rows.
//
// epi_fast: one straight-line instance per (mode, C dtype, residual, C2) combination the backbones use.  gfx9 counts loads
// and stores on ONE in-order counter (vmcnt), so a load issued after a store cannot be waited for without waiting for that
// store's acknowledgement: every residual / aux load of the whole wave tile is therefore issued BEFORE the first store.
// The code is kept small on purpose (fast erf, nothing data-type generic inside the unrolled passes): the generic
// epi_store4 unrolled 32x was ~170 KB of cold, branchy code and cost more than the K=1024 main loop.
template <int V>
struct IC {
  static constexpr int value = V;
};

// Accumulator views: what the wave-tile epilogues need of a wave's accumulators is "write 32 x 32 block (bm, j) into rows [s*32, s*32+32),
// columns [j*32, j*32+32) of the per-wave LDS image".  Acc32: the tile is (MI x NI) blocks of v_mfma_f32_32x32x16 (16 registers per block,
// register r of lane (fr, fh) = row (r&3) + 8 (r>>2) + 4 fh, column fr).  Acc16: the same tile as (2 MI x 2 NI) blocks of
// v_mfma_f32_16x16x32 (4 registers per block, register r of lane (fc, fq) = row 4 fq + r, column fc).
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int MI, int NI>
struct Acc32 {
  f32x16 (&a)[MI][NI];
  __device__ __forceinline__ void dump(float* img, int LD, int s, int bm, int j, int lane) const {
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) img[(s * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * LD + j * 32 + fr] = a[bm][j][r];
  }
};
template <int MI16, int NI16>
struct Acc16 {
  f32x4v (&a)[MI16][NI16];
  __device__ __forceinline__ void dump(float* img, int LD, int s, int bm, int j, int lane) const {
    const int fc = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int r = 0; r < 4; ++r) img[(s * 32 + a2 * 16 + fq * 4 + r) * LD + j * 32 + b2 * 16 + fc] = a[2 * bm + a2][2 * j + b2][r];
  }
};

__device__ __forceinline__ float erf_fast(float x) {  // Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float y = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_fast(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_fast(float x) {
  // Phi(x) + x phi(x): the exponential inside erf(x / sqrt 2) IS exp(-x^2 / 2), so one v_exp serves both terms
  const float az = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float ex = __expf(-az * az);
  const float erfv = copysignf(1.0f - p * t * ex, x);
  return fmaf(x, 0.39894228040143267794f * ex, fmaf(0.5f, erfv, 0.5f));
}

// RES: 0 none, 1 fp32 residual, 2 bf16 residual.  MODE: VFM_EP_*; modes 3/4 read a bf16 aux.  CDT: dtype of C.  HASC2: bf16 copy of the
// pre-activation value.
template <int MODE, int CDT, int RES, bool HASC2, int MI, int NI, int GROUP, class ACC>
__device__ __forceinline__ void epi_fast(const EpiParams& e, long zoff, const ACC& acc, float* img, int lane, long m_base,
                                         long n_base, long M, long N) {
  constexpr int WN = NI * 32, LD = WN + 4, LPR = WN / 4, RPP = 64 / LPR, PPS = 32 / RPP, NP = MI * PPS, NG = MI / GROUP;
  static_assert(MI % GROUP == 0 && NG <= 4, "slab grouping");
  constexpr bool AUX = (MODE == VFM_EP_MUL_GELU_GRAD || MODE == VFM_EP_MUL || MODE == VFM_EP_MUL_QGELU_GRAD);
  const int rr = lane / LPR, cc = (lane % LPR) * 4;
  const long n = n_base + cc;
  const bool n_ok = n < N;  // N % 4 == 0 on the vector path
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(1.f, 1.f, 1.f, 1.f);
  if (e.bias && n_ok) b4 = *reinterpret_cast<const float4*>(e.bias + n % e.bias_mod);
  if (e.colscale && n_ok) s4 = *reinterpret_cast<const float4*>(e.colscale + n);
  const float alpha = e.alpha;

  auto dump = [&](auto Gc) {
    constexpr int gi = decltype(Gc)::value;
#pragma unroll
    for (int s = 0; s < GROUP; ++s)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc.dump(img, LD, s, gi * GROUP + s, j, lane);
  };
  dump(IC<0>{});

  // ---- every load of the wave tile, before any store
  float4 res[RES == 1 ? NP : 1];
  ushort4 resh[RES == 2 ? NP : 1];
  ushort4 aux[AUX ? NP : 1];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const long m = m_base + (p / PPS) * 32 + (p % PPS) * RPP + rr;
    const bool ok = n_ok && m < M;
    if constexpr (RES == 1) {
      res[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) res[p] = *reinterpret_cast<const float4*>((const float*)e.residual + zoff + m * e.ldr + n);
    }
    if constexpr (RES == 2) {
      resh[p] = make_ushort4(0, 0, 0, 0);
      if (ok) resh[p] = *reinterpret_cast<const ushort4*>((const bf16_t*)e.residual + zoff + m * e.ldr + n);
    }
    if constexpr (AUX) {
      aux[p] = make_ushort4(0, 0, 0, 0);
      if (ok) aux[p] = *reinterpret_cast<const ushort4*>((const bf16_t*)e.aux + m * e.ld_aux + n);
    }
  }

  auto group = [&](auto Gc) {
    constexpr int gi = decltype(Gc)::value;
    if constexpr (gi > 0) dump(Gc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < GROUP * PPS; ++q) {
      constexpr int dummy = 0;
      (void)dummy;
      const int p = gi * GROUP * PPS + q;
      const int irow = (q / PPS) * 32 + (q % PPS) * RPP + rr;
      const long m = m_base + (p / PPS) * 32 + (p % PPS) * RPP + rr;
      const float4 v = *reinterpret_cast<const float4*>(img + irow * LD + cc);
      float x[4] = {fmaf(v.x, alpha, b4.x), fmaf(v.y, alpha, b4.y), fmaf(v.z, alpha, b4.z), fmaf(v.w, alpha, b4.w)};
      const bool ok = n_ok && m < M;
      if constexpr (HASC2) {
        const ushort4 c2 = {f32_to_bf16(x[0]), f32_to_bf16(x[1]), f32_to_bf16(x[2]), f32_to_bf16(x[3])};
        if (ok) *reinterpret_cast<ushort4*>((bf16_t*)e.C2 + zoff + m * e.ldc2 + n) = c2;
      }
      if constexpr (MODE == VFM_EP_GELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = gelu_fast(x[i]);
      } else if constexpr (MODE == VFM_EP_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = fmaxf(x[i], 0.f);
      } else if constexpr (MODE == VFM_EP_MUL_GELU_GRAD) {
        x[0] *= gelu_grad_fast(bf16_to_f32(aux[p].x)), x[1] *= gelu_grad_fast(bf16_to_f32(aux[p].y));
        x[2] *= gelu_grad_fast(bf16_to_f32(aux[p].z)), x[3] *= gelu_grad_fast(bf16_to_f32(aux[p].w));
      } else if constexpr (MODE == VFM_EP_QGELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = qgelu_f(x[i]);
      } else if constexpr (MODE == VFM_EP_MUL_QGELU_GRAD) {
        x[0] *= qgelu_grad_f(bf16_to_f32(aux[p].x)), x[1] *= qgelu_grad_f(bf16_to_f32(aux[p].y));
        x[2] *= qgelu_grad_f(bf16_to_f32(aux[p].z)), x[3] *= qgelu_grad_f(bf16_to_f32(aux[p].w));
      } else if constexpr (MODE == VFM_EP_MUL) {
        x[0] *= bf16_to_f32(aux[p].x), x[1] *= bf16_to_f32(aux[p].y), x[2] *= bf16_to_f32(aux[p].z), x[3] *= bf16_to_f32(aux[p].w);
      }
      x[0] *= s4.x, x[1] *= s4.y, x[2] *= s4.z, x[3] *= s4.w;
      if constexpr (RES == 1) x[0] += res[p].x, x[1] += res[p].y, x[2] += res[p].z, x[3] += res[p].w;
      if constexpr (RES == 2)
        x[0] += bf16_to_f32(resh[p].x), x[1] += bf16_to_f32(resh[p].y), x[2] += bf16_to_f32(resh[p].z), x[3] += bf16_to_f32(resh[p].w);
      if (ok) {
        const long o = zoff + m * e.ldc + n;
        if constexpr (CDT == VFM_BF16) {
          const ushort4 pk = {f32_to_bf16(x[0]), f32_to_bf16(x[1]), f32_to_bf16(x[2]), f32_to_bf16(x[3])};
          *reinterpret_cast<ushort4*>((bf16_t*)e.C + o) = pk;
        } else {
          *reinterpret_cast<float4*>((float*)e.C + o) = make_float4(x[0], x[1], x[2], x[3]);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  group(IC<0>{});
  if constexpr (NG > 1) group(IC<1>{});
  if constexpr (NG > 2) group(IC<2>{});
  if constexpr (NG > 3) group(IC<3>{});
}

// ------------------------------------------------------------------------------------------------------------------
// epi_fast8: the same straight-line scheme at EIGHT columns per lane (one 16-byte store per bf16 output row piece: the store tail
// of a tile is issue-bound, not bandwidth-bound) with the arithmetic on pairs (v_pk_fma_f32 / v_pk_mul_f32: two columns per
// VALU issue).  C2MODE: 0 no second output, 1 = bf16 pre-activation, 2 = bf16 derivative of the activation at the pre-activation
// (GELU: Phi(v) + v phi(v); its exponential is the one inside erf, so both outputs cost one rcp + one exp per element) - the
// backward GEMM then only multiplies (VFM_EP_MUL) instead of re-evaluating the derivative from a bf16-rounded copy.
// Measured inside the train step: the epilogue of the fc1 / fc2-dgrad GEMMs (GELU on 16.8 M values, 64 MB of
// stores) cost 27-34 us of a 54-59 us launch.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// 16-byte store, nontemporal on request: the store-heavy GEMMs (fc1 forward: 67 MB per launch) lose ~4 us to write-allocated lines
// that only leave the L2 at the end of the kernel; streamed out as they are produced they overlap the remaining tiles' loops
__device__ __forceinline__ void st16(void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d, int nt) {
  if (nt) __builtin_nontemporal_store(u32x4_t{a, b, c, d}, reinterpret_cast<u32x4_t*>(p));
  else *reinterpret_cast<uint4*>(p) = make_uint4(a, b, c, d);
}

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

// g = gelu(v), dg = gelu'(v) for a pair (Abramowitz-Stegun 7.1.26 erf, |err| <= 1.5e-7, as erf_fast / gelu_grad_fast)
template <bool WANT_DG>
__device__ __forceinline__ void gelu_pair(f32x2 v, f32x2& g, f32x2& dg) {
  const f32x2 av = f32x2{fabsf(v.x), fabsf(v.y)};
  const f32x2 den = pk_fma(splat2(0.3275911f * 0.70710678118654752440f), av, splat2(1.0f));
  const f32x2 t = f32x2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  f32x2 p = pk_fma(t, splat2(1.061405429f), splat2(-1.453152027f));
  p = pk_fma(t, p, splat2(1.421413741f));
  p = pk_fma(t, p, splat2(-0.284496736f));
  p = pk_fma(t, p, splat2(0.254829592f));
  const f32x2 e2 = v * v * splat2(-0.5f * 1.44269504088896340736f);        // exp(-v^2/2) = exp2(-v^2/2 * log2 e)
  const f32x2 ex = f32x2{__builtin_amdgcn_exp2f(e2.x), __builtin_amdgcn_exp2f(e2.y)};
  const f32x2 q = p * t * ex;                                               // 1 - erf(|v| / sqrt 2)
  const f32x2 erfa = splat2(1.0f) - q;
  const f32x2 erfv = f32x2{copysignf(erfa.x, v.x), copysignf(erfa.y, v.y)};
  const f32x2 cdf = pk_fma(splat2(0.5f), erfv, splat2(0.5f));
  g = v * cdf;
  if constexpr (WANT_DG) dg = pk_fma(v * splat2(0.39894228040143267794f), ex, cdf);
}

__device__ __forceinline__ uint32_t pack_bf16x2(f32x2 v) {
  typedef vfm_h bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);   // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
  return *reinterpret_cast<const uint32_t*>(&b);
}
__device__ __forceinline__ f32x2 unpack_bf16x2(uint32_t w) { return f32x2{h16_lo(w), h16_hi(w)}; }

template <int MODE, int CDT, int RES, int C2MODE, int MI, int NI, int GROUP, class ACC>
__device__ __forceinline__ void epi_fast8(const EpiParams& e, long zoff, const ACC& acc, float* img, int lane, long m_base,
                                          long n_base, long M, long N) {
  constexpr int WN = NI * 32, LD = WN + 4, LPR = WN / 8, RPP = 64 / LPR, PPS = 32 / RPP, NP = MI * PPS, NG = MI / GROUP;
  static_assert(MI % GROUP == 0 && NG <= 4, "slab grouping");
  constexpr bool AUX = (MODE == VFM_EP_MUL_GELU_GRAD || MODE == VFM_EP_MUL || MODE == VFM_EP_MUL_QGELU_GRAD);
  const int rr = lane / LPR, cc = (lane % LPR) * 8;
  const long n = n_base + cc;
  const bool n_ok = n < N;  // N % 8 == 0 on this path
  f32x2 b2[4], s2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) b2[i] = splat2(0.f), s2[i] = splat2(1.f);
  if (e.bias && n_ok) {
    const long bn = n % e.bias_mod;
    const float4 lo = *reinterpret_cast<const float4*>(e.bias + bn), hi = *reinterpret_cast<const float4*>(e.bias + bn + 4);
    b2[0] = f32x2{lo.x, lo.y}, b2[1] = f32x2{lo.z, lo.w}, b2[2] = f32x2{hi.x, hi.y}, b2[3] = f32x2{hi.z, hi.w};
  }
  if (e.colscale && n_ok) {
    const float4 lo = *reinterpret_cast<const float4*>(e.colscale + n), hi = *reinterpret_cast<const float4*>(e.colscale + n + 4);
    s2[0] = f32x2{lo.x, lo.y}, s2[1] = f32x2{lo.z, lo.w}, s2[2] = f32x2{hi.x, hi.y}, s2[3] = f32x2{hi.z, hi.w};
  }
  const f32x2 alpha2 = splat2(e.alpha);

  auto dump = [&](auto Gc) {
    constexpr int gi = decltype(Gc)::value;
#pragma unroll
    for (int s = 0; s < GROUP; ++s)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc.dump(img, LD, s, gi * GROUP + s, j, lane);
  };
  dump(IC<0>{});

  // ---- every load of the wave tile, before any store (loads and stores share one in-order vmcnt)
  float4 res[RES == 1 ? 2 * NP : 1];
  uint4 resh[RES == 2 ? NP : 1];
  uint4 aux[AUX ? NP : 1];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const long m = m_base + (p / PPS) * 32 + (p % PPS) * RPP + rr;
    const bool ok = n_ok && m < M;
    if constexpr (RES == 1) {
      res[2 * p] = res[2 * p + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        const float* rp = (const float*)e.residual + zoff + m * e.ldr + n;
        res[2 * p] = *reinterpret_cast<const float4*>(rp), res[2 * p + 1] = *reinterpret_cast<const float4*>(rp + 4);
      }
    }
    if constexpr (RES == 2) {
      resh[p] = make_uint4(0, 0, 0, 0);
      if (ok) resh[p] = *reinterpret_cast<const uint4*>((const bf16_t*)e.residual + zoff + m * e.ldr + n);
    }
    if constexpr (AUX) {
      aux[p] = make_uint4(0, 0, 0, 0);
      if (ok) aux[p] = *reinterpret_cast<const uint4*>((const bf16_t*)e.aux + m * e.ld_aux + n);
    }
  }

  auto group = [&](auto Gc) {
    constexpr int gi = decltype(Gc)::value;
    if constexpr (gi > 0) dump(Gc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < GROUP * PPS; ++q) {
      const int p = gi * GROUP * PPS + q;
      const int irow = (q / PPS) * 32 + (q % PPS) * RPP + rr;
      const long m = m_base + (p / PPS) * 32 + (p % PPS) * RPP + rr;
      const float4 v0 = *reinterpret_cast<const float4*>(img + irow * LD + cc), v1 = *reinterpret_cast<const float4*>(img + irow * LD + cc + 4);
      f32x2 x[4] = {pk_fma(f32x2{v0.x, v0.y}, alpha2, b2[0]), pk_fma(f32x2{v0.z, v0.w}, alpha2, b2[1]),
                    pk_fma(f32x2{v1.x, v1.y}, alpha2, b2[2]), pk_fma(f32x2{v1.z, v1.w}, alpha2, b2[3])};
      const bool ok = n_ok && m < M;
      f32x2 d[4];
      if constexpr (C2MODE == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = x[i];
      }
      if constexpr (MODE == VFM_EP_GELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x2 g, dg;
          gelu_pair<C2MODE == 2>(x[i], g, dg);
          x[i] = g;
          if constexpr (C2MODE == 2) d[i] = dg;
        }
      } else if constexpr (MODE == VFM_EP_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = f32x2{fmaxf(x[i].x, 0.f), fmaxf(x[i].y, 0.f)};
      } else if constexpr (MODE == VFM_EP_QGELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = f32x2{qgelu_f(x[i].x), qgelu_f(x[i].y)};
      } else if constexpr (AUX) {
        const uint32_t aw[4] = {aux[p].x, aux[p].y, aux[p].z, aux[p].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x2 a = unpack_bf16x2(aw[i]);
          if constexpr (MODE == VFM_EP_MUL_GELU_GRAD) a = f32x2{gelu_grad_fast(a.x), gelu_grad_fast(a.y)};
          if constexpr (MODE == VFM_EP_MUL_QGELU_GRAD) a = f32x2{qgelu_grad_f(a.x), qgelu_grad_f(a.y)};
          x[i] = x[i] * a;
        }
      }
      if constexpr (C2MODE != 0) {
        if (ok) st16((bf16_t*)e.C2 + zoff + m * e.ldc2 + n, pack_bf16x2(d[0]), pack_bf16x2(d[1]), pack_bf16x2(d[2]), pack_bf16x2(d[3]), e.nt);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = x[i] * s2[i];
      if constexpr (RES == 1) {
        x[0] += f32x2{res[2 * p].x, res[2 * p].y}, x[1] += f32x2{res[2 * p].z, res[2 * p].w};
        x[2] += f32x2{res[2 * p + 1].x, res[2 * p + 1].y}, x[3] += f32x2{res[2 * p + 1].z, res[2 * p + 1].w};
      }
      if constexpr (RES == 2) {
        x[0] += unpack_bf16x2(resh[p].x), x[1] += unpack_bf16x2(resh[p].y), x[2] += unpack_bf16x2(resh[p].z), x[3] += unpack_bf16x2(resh[p].w);
      }
      if (ok) {
        const long o = zoff + m * e.ldc + n;
        if constexpr (CDT == VFM_BF16) {
          st16((bf16_t*)e.C + o, pack_bf16x2(x[0]), pack_bf16x2(x[1]), pack_bf16x2(x[2]), pack_bf16x2(x[3]), e.nt);
        } else if constexpr (CDT == VFM_SPLIT3) {   // the split-bf16 image [hi | hi | lo] of the result: three 16-byte stores
          uint32_t hw[4], lw[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            hw[i] = pack_bf16x2(x[i]);
            lw[i] = pack_bf16x2(x[i] - unpack_bf16x2(hw[i]));
          }
          bf16_t* cp = (bf16_t*)e.C + o;
          st16(cp, hw[0], hw[1], hw[2], hw[3], 0);
          st16(cp + e.c_plane, hw[0], hw[1], hw[2], hw[3], 0);
          st16(cp + 2 * e.c_plane, lw[0], lw[1], lw[2], lw[3], 0);
        } else {
          float* cp = (float*)e.C + o;
          *reinterpret_cast<float4*>(cp) = make_float4(x[0].x, x[0].y, x[1].x, x[1].y);
          *reinterpret_cast<float4*>(cp + 4) = make_float4(x[2].x, x[2].y, x[3].x, x[3].y);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  group(IC<0>{});
  if constexpr (NG > 1) group(IC<1>{});
  if constexpr (NG > 2) group(IC<2>{});
  if constexpr (NG > 3) group(IC<3>{});
}

// any other combination: compact (rolled) pass loop around the generic epi_store4
template <int MI, int NI, int GROUP, class ACC>
__device__ __forceinline__ void epi_generic(const EpiParams& e, long zoff, const ACC& acc, float* img, int lane, long m_base,
                                            long n_base, long M, long N) {
  constexpr int WN = NI * 32, LD = WN + 4, LPR = WN / 4, RPP = 64 / LPR, PPS = 32 / RPP, NG = MI / GROUP;
  const int rr = lane / LPR, cc = (lane % LPR) * 4;
  const long n = n_base + cc;
  auto group = [&](auto Gc) {
    constexpr int gi = decltype(Gc)::value;
#pragma unroll
    for (int s = 0; s < GROUP; ++s)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc.dump(img, LD, s, gi * GROUP + s, j, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int q = 0; q < GROUP * PPS; ++q) {
      const int irow = (q / PPS) * 32 + (q % PPS) * RPP + rr;
      const long m = m_base + gi * GROUP * 32 + irow;
      const float4 v = *reinterpret_cast<const float4*>(img + irow * LD + cc);
      if (m < M && n < N) epi_store4(e, zoff, m, n, v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  group(IC<0>{});
  if constexpr (NG > 1) group(IC<1>{});
  if constexpr (NG > 2) group(IC<2>{});
  if constexpr (NG > 3) group(IC<3>{});
}

// wave-uniform selection of the epilogue instance
__device__ __forceinline__ bool epi_vec8_ok(const EpiParams& e, long zoff, long N) {
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool ok = (N % 8 == 0) && (e.ldc % 8 == 0) && (zoff % 8 == 0) && a16(e.C) && (!e.bias || e.bias_mod % 8 == 0);
  if (e.c_dt == VFM_SPLIT3) ok = ok && (e.c_plane % 8 == 0);
  if (e.C2) ok = ok && e.c2_dt == VFM_BF16 && a16(e.C2) && (e.ldc2 % 8 == 0);
  if (e.aux) ok = ok && e.aux_dt == VFM_BF16 && a16(e.aux) && (e.ld_aux % 8 == 0);
  if (e.residual) ok = ok && a16(e.residual) && (e.ldr % 8 == 0);
  return ok;
}

template <int MI, int NI, int GROUP, class ACC>
__device__ __forceinline__ void epi_wave_tile_acc(const EpiParams& e, long zoff, const ACC& acc, float* img, int lane, long m_base,
                                                  long n_base, long M, long N) {
  const int mode = e.ep_mode;
  const bool plain = !e.C2 && !e.residual;
  if (epi_vec8_ok(e, zoff, N)) {   // eight columns per lane, packed arithmetic: every shape of the four backbones
    if (mode == VFM_EP_GELU && plain && e.c_dt == VFM_SPLIT3)   // bf16x3 predictions: fc1's output straight into fc2's split A operand
      return epi_fast8<VFM_EP_GELU, VFM_SPLIT3, 0, 0, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
    if (mode == VFM_EP_NONE && plain && e.c_dt == VFM_SPLIT3)
      return epi_fast8<VFM_EP_NONE, VFM_SPLIT3, 0, 0, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
    if (mode == VFM_EP_NONE && plain && e.c_dt == VFM_BF16)
      return epi_fast8<VFM_EP_NONE, VFM_BF16, 0, 0, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
    if (mode == VFM_EP_NONE && !e.C2 && e.residual && e.r_dt == VFM_F32 && e.c_dt == VFM_F32)
      return epi_fast8<VFM_EP_NONE, VFM_F32, 1, 0, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
    if (mode == VFM_EP_GELU_DGELU && e.C2 && !e.residual && e.c_dt == VFM_BF16)
      return epi_fast8<VFM_EP_GELU, VFM_BF16, 0, 2, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
    if (mode == VFM_EP_GELU && e.C2 && !e.residual && e.c_dt == VFM_BF16)
      return epi_fast8<VFM_EP_GELU, VFM_BF16, 0, 1, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
    if (mode == VFM_EP_GELU && plain && e.c_dt == VFM_BF16)
      return epi_fast8<VFM_EP_GELU, VFM_BF16, 0, 0, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
    if (mode == VFM_EP_MUL && plain && e.c_dt == VFM_BF16)
      return epi_fast8<VFM_EP_MUL, VFM_BF16, 0, 0, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
    if (mode == VFM_EP_MUL_GELU_GRAD && plain && e.c_dt == VFM_BF16)
      return epi_fast8<VFM_EP_MUL_GELU_GRAD, VFM_BF16, 0, 0, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
    if (mode == VFM_EP_MUL && !e.C2 && e.residual && e.r_dt == VFM_BF16 && e.c_dt == VFM_BF16)
      return epi_fast8<VFM_EP_MUL, VFM_BF16, 2, 0, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);  // LoRA: dx += mask * (dT A)
    if (mode == VFM_EP_NONE && !e.C2 && e.residual && e.r_dt == VFM_BF16 && e.c_dt == VFM_BF16)
      return epi_fast8<VFM_EP_NONE, VFM_BF16, 2, 0, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
  }
  if (e.c_dt == VFM_SPLIT3)
    epi_generic<MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
  else if (mode == VFM_EP_NONE && plain && e.c_dt == VFM_BF16)
    epi_fast<VFM_EP_NONE, VFM_BF16, 0, false, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
  else if (mode == VFM_EP_NONE && plain && e.c_dt == VFM_F32)
    epi_fast<VFM_EP_NONE, VFM_F32, 0, false, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
  else if (mode == VFM_EP_NONE && !e.C2 && e.residual && e.r_dt == VFM_F32 && e.c_dt == VFM_F32)
    epi_fast<VFM_EP_NONE, VFM_F32, 1, false, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
  else if (mode == VFM_EP_QGELU && e.C2 && e.c2_dt == VFM_BF16 && !e.residual && e.c_dt == VFM_BF16)
    epi_fast<VFM_EP_QGELU, VFM_BF16, 0, true, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
  else if (mode == VFM_EP_MUL_QGELU_GRAD && e.aux_dt == VFM_BF16 && plain && e.c_dt == VFM_BF16)
    epi_fast<VFM_EP_MUL_QGELU_GRAD, VFM_BF16, 0, false, MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
  else
    epi_generic<MI, NI, GROUP>(e, zoff, acc, img, lane, m_base, n_base, M, N);
}

template <int MI, int NI, int GROUP>
__device__ __forceinline__ void epi_wave_tile(const EpiParams& e, long zoff, f32x16 (&acc)[MI][NI], float* img, int lane, long m_base,
                                              long n_base, long M, long N) {
  epi_wave_tile_acc<MI, NI, GROUP>(e, zoff, Acc32<MI, NI>{acc}, img, lane, m_base, n_base, M, N);
}

// unaligned outputs (N % 4 != 0, odd leading dims ...): same LDS image, one element per lane and step, rolled loop
template <int MI, int NI, int GROUP, class ACC>
__device__ __forceinline__ void epi_scalar_acc(const EpiParams& e, long zoff, const ACC& acc, float* img, int lane, long m_base,
                                               long n_base, long M, long N) {
  constexpr int WN = NI * 32, LD = WN + 4, NG = MI / GROUP;
  auto group = [&](auto Gc) {
    constexpr int gi = decltype(Gc)::value;
#pragma unroll
    for (int s = 0; s < GROUP; ++s)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc.dump(img, LD, s, gi * GROUP + s, j, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int idx = lane; idx < GROUP * 32 * WN; idx += 64) {
      const int irow = idx / WN, col = idx % WN;
      const long m = m_base + gi * GROUP * 32 + irow, n = n_base + col;
      if (m < M && n < N) epi_store(e, zoff, m, n, img[irow * LD + col]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  group(IC<0>{});
  if constexpr (NG > 1) group(IC<1>{});
  if constexpr (NG > 2) group(IC<2>{});
  if constexpr (NG > 3) group(IC<3>{});
}

template <int MI, int NI, int GROUP>
__device__ __forceinline__ void epi_scalar(const EpiParams& e, long zoff, f32x16 (&acc)[MI][NI], float* img, int lane, long m_base,
                                           long n_base, long M, long N) {
  epi_scalar_acc<MI, NI, GROUP>(e, zoff, Acc32<MI, NI>{acc}, img, lane, m_base, n_base, M, N);
}

// ------------------------------------------------------------------------------------------------ skinny tail
// M <= 32 rows (the [cls] rows that follow the 128-aligned patch-token rows): one 512-thread block = 32 rows x 32 columns,
// the 8 waves split K eight ways, B goes global -> registers -> a wave-private LDS image -> MFMA fragments (no block
// barriers in the loop: B is streamed exactly once), partial sums meet in LDS for the shared epilogue.  64 KiB of LDS.
// It is a device function so that the tile kernels can run it in extra blocks at the END of their grid (the tail rows of
// M = 4100 then cost no launch of their own; last linear ids: they fill in as CUs free up and displace nothing).
struct SkinnyTail {
  const bf16_t* A; long lda; long M; int nblk; EpiParams e;
};
// K is ALWAYS split eight ways and the eight partial sums are added in the same order, whatever the number of waves of the block (a
// 4-wave block runs slices w and w + 4 one after the other): the tail rows get the same bits from every tile configuration.
template <int NW = 8>
__device__ __forceinline__ void skinny_tile(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ Bz, long ldb, long M, long N,
                                            long K, long n0, const EpiParams& e, long zoff, char* smem) {
  constexpr int VW = 8;   // K slices
  static_assert(VW % NW == 0, "whole slices per wave");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  long am = fr;
  if (am > M - 1) am = M - 1;
  const bf16_t* ap = A + am * lda + fh * 8;
  // coalesced loader map: one wave-instruction = 4 rows x 256 B
  const int lrow = lane >> 4, lch = lane & 15;
  const bf16_t* bsrc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    long bn = n0 + 4 * i + lrow;
    if (bn > N - 1) bn = N - 1;
    bsrc[i] = Bz + bn * ldb + lch * 8;
  }
  const long nchunk = K / 128;                       // K % 128 may be 64: handled by the remainder chunk below
  const long per = (nchunk + VW - 1) / VW;   // eight slices of K
  f32x16 acc[VW / NW];
  char* W = smem + wave * 8192;  // [32 columns(n) x 128 k] slice of B: 256-B rows, chunk c of row r at c ^ (r & 15)
#pragma unroll
  for (int u = 0; u < VW / NW; ++u) {
    const int vw = wave + u * NW;   // this wave's u-th slice
    const long c_beg = vw * per, c_end = (c_beg + per < nchunk) ? c_beg + per : nchunk;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    for (long cidx = c_beg; cidx < c_end; ++cidx) {
      const long k0 = cidx * 128;
      uint4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const uint4*>(bsrc[i] + k0);
      bf16x8 af[8];
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) af[s2] = *reinterpret_cast<const bf16x8*>(ap + k0 + 16 * s2);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + lrow;
        *reinterpret_cast<uint4*>(W + row * 256 + ((lch ^ (row & 15)) << 4)) = v[i];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) {
        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(W + fr * 256 + (((2 * s2 + fh) ^ (fr & 15)) << 4));
        acc[u] = VFM_MFMA16(af[s2], bf, acc[u]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    if ((K % 128) != 0 && vw == VW - 1) {  // trailing 64-wide half chunk (K % 64 == 0 is guaranteed by the caller): the last slice's
      const long k0 = nchunk * 128;
      long bn = n0 + fr;
      if (bn > N - 1) bn = N - 1;
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + k0 + 16 * s2);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bz + bn * ldb + fh * 8 + k0 + 16 * s2);
        acc[u] = VFM_MFMA16(a, b, acc[u]);
      }
    }
  }
  __syncthreads();  // every wave is done with its staging slice: the partial sums reuse that memory
  float* part = reinterpret_cast<float*>(smem);  // [VW][32][33]
#pragma unroll
  for (int u = 0; u < VW / NW; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[((wave + u * NW) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * 33 + fr] = acc[u][r];
  __syncthreads();
  for (int i = tid; i < 32 * 32; i += NW * 64) {
    const int row = i >> 5, col = i & 31;
    const long m = row, n = n0 + col;
    if (m < M && n < N) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < VW; ++w) v += part[(w * 32 + row) * 33 + col];
      epi_store(e, zoff, m, n, v);
    }
  }
}
