// LayerNorm / GroupNorm / BatchNorm, forward and backward, on token-major fp32 maps. HBM-bound.
#include "common.h"

__device__ __forceinline__ float act_f(float v, int act) {
  return act == VFM_ACT_GELU ? gelu_f(v) : (act == VFM_ACT_RELU ? fmaxf(v, 0.f) : v);
}
__device__ __forceinline__ float act_grad_f(float v, int act) {
  return act == VFM_ACT_GELU ? gelu_grad_f(v) : (act == VFM_ACT_RELU ? (v > 0.f ? 1.f : 0.f) : 1.f);
}

// =============================================================================================== LayerNorm
// one wave per row; C <= 64*32; two-pass statistics in registers (matches torch's accuracy class)
template <typename TO, int MAXPL>
__global__ void k_ln_fwd(const float* __restrict__ x, long ld_x, const float* __restrict__ w, const float* __restrict__ b,
                         float eps, TO* __restrict__ y, long ld_y, float* __restrict__ stats, long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ld_x;
  float v[MAXPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int c = i * 64 + lane;
    v[i] = c < C ? xr[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int c = i * 64 + lane;
    const float d = c < C ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / C + eps);
  if (stats && lane == 0) {
    stats[row * 2] = mean;
    stats[row * 2 + 1] = rstd;
  }
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int c = i * 64 + lane;
    if (c < C) st_f32(y + row * ld_y + c, (v[i] - mean) * rstd * w[c] + b[c]);
  }
}


// ---- vectorised LayerNorm (C % 256 == 0): each lane owns float4 chunks -> 16-byte loads/stores (Guideline 13)
// DROP: also writes the dropout multiplier (0 or 1/(1-p), element index offset + row*C + c of the counter-based RNG, the
// values vfm_dropout_mask produces) and y_drop = y * multiplier - the LoRA branch input - in the same pass.
struct LnDrop {
  void* yd; long ld_yd; void* mask; long ld_mask; float p, keep_scale; uint64_t seed, offset;
};
// Split-bf16 image of an fp32 value (bf16x3 mode: what vfm_split3 pattern 0 writes for a GEMM's A operand): hi at column c, hi again at
// plane + c, lo at 2 plane + c.  A producer that writes it directly saves the consumer's vfm_split3 pass (4 B read + 6 B written per value).
struct Split3Out {
  bf16_t* y3; long ld3; long plane;
};
__device__ __forceinline__ void store_split4(const Split3Out& so, long row, int c0, float4 o) {
  const bf16_t h0 = f32_to_bf16(o.x), h1 = f32_to_bf16(o.y), h2 = f32_to_bf16(o.z), h3 = f32_to_bf16(o.w);
  const ushort4 hi = {h0, h1, h2, h3};
  const ushort4 lo = {f32_to_bf16(o.x - bf16_to_f32(h0)), f32_to_bf16(o.y - bf16_to_f32(h1)), f32_to_bf16(o.z - bf16_to_f32(h2)),
                      f32_to_bf16(o.w - bf16_to_f32(h3))};
  bf16_t* p = so.y3 + row * so.ld3 + c0;
  *reinterpret_cast<ushort4*>(p) = hi;
  *reinterpret_cast<ushort4*>(p + so.plane) = hi;
  *reinterpret_cast<ushort4*>(p + 2 * so.plane) = lo;
}
template <typename TO, int NV, bool DROP = false, bool SPLIT = false>
__global__ void k_ln_fwd_v4(const float* __restrict__ x, long ld_x, const float* __restrict__ w, const float* __restrict__ b,
                            float eps, TO* __restrict__ y, long ld_y, float* __restrict__ stats, long rows, int C, LnDrop dr = LnDrop(),
                            Split3Out so = Split3Out()) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ld_x;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = *reinterpret_cast<const float4*>(xr + (i * 64 + lane) * 4);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
    q += (a * a + bb * bb) + (c * c + d * d);
  }
  const float rstd = rsqrtf(wave_sum(q) / C + eps);
  if (stats && lane == 0) {
    stats[row * 2] = mean;
    stats[row * 2 + 1] = rstd;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c0 = (i * 64 + lane) * 4;
    const float4 ww = *reinterpret_cast<const float4*>(w + c0);
    const float4 bv = *reinterpret_cast<const float4*>(b + c0);
    float4 o;
    o.x = (v[i].x - mean) * rstd * ww.x + bv.x;
    o.y = (v[i].y - mean) * rstd * ww.y + bv.y;
    o.z = (v[i].z - mean) * rstd * ww.z + bv.z;
    o.w = (v[i].w - mean) * rstd * ww.w + bv.w;
    if constexpr (sizeof(TO) == 2) {
      ushort4 p = {f32_to_bf16(o.x), f32_to_bf16(o.y), f32_to_bf16(o.z), f32_to_bf16(o.w)};
      *reinterpret_cast<ushort4*>(y + row * ld_y + c0) = p;
      if constexpr (DROP) {
        const uint64_t e0 = dr.offset + (uint64_t)row * C + c0;
        const bf16_t ks = f32_to_bf16(dr.keep_scale);
        const float ksf = bf16_to_f32(ks);
        ushort4 mk, yd;
        const bool k0 = (hash_u32(dr.seed, e0) >> 8) * (1.0f / 16777216.0f) >= dr.p;
        const bool k1 = (hash_u32(dr.seed, e0 + 1) >> 8) * (1.0f / 16777216.0f) >= dr.p;
        const bool k2 = (hash_u32(dr.seed, e0 + 2) >> 8) * (1.0f / 16777216.0f) >= dr.p;
        const bool k3 = (hash_u32(dr.seed, e0 + 3) >> 8) * (1.0f / 16777216.0f) >= dr.p;
        mk.x = k0 ? ks : 0, mk.y = k1 ? ks : 0, mk.z = k2 ? ks : 0, mk.w = k3 ? ks : 0;
        yd.x = k0 ? f32_to_bf16(bf16_to_f32(p.x) * ksf) : f32_to_bf16(bf16_to_f32(p.x) * 0.f);
        yd.y = k1 ? f32_to_bf16(bf16_to_f32(p.y) * ksf) : f32_to_bf16(bf16_to_f32(p.y) * 0.f);
        yd.z = k2 ? f32_to_bf16(bf16_to_f32(p.z) * ksf) : f32_to_bf16(bf16_to_f32(p.z) * 0.f);
        yd.w = k3 ? f32_to_bf16(bf16_to_f32(p.w) * ksf) : f32_to_bf16(bf16_to_f32(p.w) * 0.f);
        *reinterpret_cast<ushort4*>((bf16_t*)dr.mask + row * dr.ld_mask + c0) = mk;
        *reinterpret_cast<ushort4*>((bf16_t*)dr.yd + row * dr.ld_yd + c0) = yd;
      }
    } else {
      if (!SPLIT || y != nullptr) *reinterpret_cast<float4*>(y + row * ld_y + c0) = o;
      if constexpr (SPLIT) store_split4(so, row, c0, o);
    }
  }
}

template <typename TD, int NV>
__global__ void k_ln_bwd_v4(const TD* __restrict__ dy, long ld_dy, const float* __restrict__ x, long ld_x,
                            const float* __restrict__ w, const float* __restrict__ stats, float* __restrict__ dx, long ld_dx,
                            int accumulate_dx, long rows, int C, bf16_t* __restrict__ t_out = nullptr, long ld_t = 0,
                            const float* __restrict__ t_scale = nullptr) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
  float4 g[NV], xh[NV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c0 = (i * 64 + lane) * 4;
    float4 d;
    if constexpr (sizeof(TD) == 2) {
      const ushort4 p = *reinterpret_cast<const ushort4*>(dy + row * ld_dy + c0);
      d = make_float4(bf16_to_f32(p.x), bf16_to_f32(p.y), bf16_to_f32(p.z), bf16_to_f32(p.w));
    } else {
      d = *reinterpret_cast<const float4*>(dy + row * ld_dy + c0);
    }
    const float4 xv = *reinterpret_cast<const float4*>(x + row * ld_x + c0);
    const float4 ww = *reinterpret_cast<const float4*>(w + c0);
    xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
    g[i] = make_float4(d.x * ww.x, d.y * ww.y, d.z * ww.z, d.w * ww.w);
    s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
    s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
  }
  s1 = wave_sum(s1) / C;
  s2 = wave_sum(s2) / C;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c0 = (i * 64 + lane) * 4;
    float4 o = make_float4(rstd * (g[i].x - s1 - xh[i].x * s2), rstd * (g[i].y - s1 - xh[i].y * s2),
                           rstd * (g[i].z - s1 - xh[i].z * s2), rstd * (g[i].w - s1 - xh[i].w * s2));
    float4* p = reinterpret_cast<float4*>(dx + row * ld_dx + c0);
    if (accumulate_dx) {
      const float4 old = *p;
      o.x += old.x, o.y += old.y, o.z += old.z, o.w += old.w;
    }
    *p = o;
    if (t_out) {  // bf16(dx * t_scale): the operand of the next (preceding-branch) dgrad GEMM, saves a cast pass
      const float4 ts = *reinterpret_cast<const float4*>(t_scale + c0);
      const ushort4 tv = {f32_to_bf16(o.x * ts.x), f32_to_bf16(o.y * ts.y), f32_to_bf16(o.z * ts.z), f32_to_bf16(o.w * ts.w)};
      *reinterpret_cast<ushort4*>(t_out + row * ld_t + c0) = tv;
    }
  }
}

template <typename TO, int MAXP2>
__global__ void k_ln_fwd_v2(const float* __restrict__ x, long ld_x, const float* __restrict__ w, const float* __restrict__ b, float eps,
                            TO* __restrict__ y, long ld_y, float* __restrict__ stats, long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ld_x;
  float2 v[MAXP2];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP2; ++i) {
    const int c = (i * 64 + lane) * 2;
    v[i] = c < C ? *reinterpret_cast<const float2*>(xr + c) : make_float2(0.f, 0.f);
    s += v[i].x + v[i].y;
  }
  const float mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP2; ++i) {
    const int c = (i * 64 + lane) * 2;
    if (c < C) {
      const float a = v[i].x - mean, bb = v[i].y - mean;
      q += a * a + bb * bb;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / C + eps);
  if (stats && lane == 0) {
    stats[row * 2] = mean;
    stats[row * 2 + 1] = rstd;
  }
#pragma unroll
  for (int i = 0; i < MAXP2; ++i) {
    const int c = (i * 64 + lane) * 2;
    if (c < C) {
      const float2 ww = *reinterpret_cast<const float2*>(w + c), bv = *reinterpret_cast<const float2*>(b + c);
      const float o0 = (v[i].x - mean) * rstd * ww.x + bv.x, o1 = (v[i].y - mean) * rstd * ww.y + bv.y;
      if constexpr (sizeof(TO) == 2) {
        const ushort2 pk = {f32_to_bf16(o0), f32_to_bf16(o1)};
        *reinterpret_cast<ushort2*>(y + row * ld_y + c) = pk;
      } else {
        *reinterpret_cast<float2*>(y + row * ld_y + c) = make_float2(o0, o1);
      }
    }
  }
}

// the same rows when their PITCH is a multiple of four elements (EVA02 keeps the 2730-wide SwiGLU hidden in a 2752-wide buffer): 16-byte
// pieces, the last piece of a row partly valid (read as a whole - the pitch covers it -, used and written element by element).
template <typename TO, int MAXP4>
__global__ void k_ln_fwd_v4t(const float* __restrict__ x, long ld_x, const float* __restrict__ w, const float* __restrict__ b, float eps,
                             TO* __restrict__ y, long ld_y, float* __restrict__ stats, long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ld_x;
  float v[MAXP4][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP4; ++i) {
    const int c = (i * 64 + lane) * 4;
    const float4 t = c < C ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    v[i][0] = t.x, v[i][1] = t.y, v[i][2] = t.z, v[i][3] = t.w;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < C) s += v[i][e];
  }
  const float mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP4; ++i) {
    const int c = (i * 64 + lane) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < C) q += (v[i][e] - mean) * (v[i][e] - mean);
  }
  const float rstd = rsqrtf(wave_sum(q) / C + eps);
  if (stats && lane == 0) {
    stats[row * 2] = mean;
    stats[row * 2 + 1] = rstd;
  }
#pragma unroll
  for (int i = 0; i < MAXP4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      if (c + 4 <= C) {
        const float4 ww = *reinterpret_cast<const float4*>(w + c), bv = *reinterpret_cast<const float4*>(b + c);
        const float o0 = (v[i][0] - mean) * rstd * ww.x + bv.x, o1 = (v[i][1] - mean) * rstd * ww.y + bv.y;
        const float o2 = (v[i][2] - mean) * rstd * ww.z + bv.z, o3 = (v[i][3] - mean) * rstd * ww.w + bv.w;
        if constexpr (sizeof(TO) == 2) {
          const ushort4 pk = {f32_to_bf16(o0), f32_to_bf16(o1), f32_to_bf16(o2), f32_to_bf16(o3)};
          *reinterpret_cast<ushort4*>(y + row * ld_y + c) = pk;
        } else {
          *reinterpret_cast<float4*>(y + row * ld_y + c) = make_float4(o0, o1, o2, o3);
        }
      } else {   // the partly valid last piece: element by element, nothing is written past column C
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < C) {
            const float o = (v[i][e] - mean) * rstd * w[c + e] + b[c + e];
            if constexpr (sizeof(TO) == 2) y[row * ld_y + c + e] = f32_to_bf16(o);
            else y[row * ld_y + c + e] = o;
          }
      }
    }
  }
}

extern "C" int vfm_layernorm_fwd_split3(const float* x, long ld_x, const float* w, const float* b, float eps, float* y, long ld_y, void* y3,
                                        long ld3, long plane, float* stats, long rows, long C, void* stream) {
  const int nv = (int)(C / 256);
  VFM_CHECK(C > 0 && C % 256 == 0 && (nv == 1 || nv == 2 || nv == 4 || nv == 5 || nv == 8) && ld_x >= C && ld_x % 4 == 0 && (uintptr_t)x % 16 == 0 &&
                (uintptr_t)w % 16 == 0 && (uintptr_t)b % 16 == 0,
            VFM_E_SHAPE, "vfm_layernorm_fwd_split3: C=%ld must be 256, 512, 1024, 1280 or 2048 with 16-byte aligned rows", C);
  VFM_CHECK(y3 && (uintptr_t)y3 % 8 == 0 && plane >= C && plane % 4 == 0 && ld3 >= 3 * plane && ld3 % 4 == 0 && (!y || (ld_y >= C && ld_y % 4 == 0 && (uintptr_t)y % 16 == 0)),
            VFM_E_ALIGN, "vfm_layernorm_fwd_split3: y3 [rows, >= 3 plane] bf16 with plane >= C, 8-byte aligned; y (optional fp32 copy) 16-byte aligned");
  if (rows == 0) return VFM_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(rows, 4)), blk(256);
  Split3Out so;
  so.y3 = (bf16_t*)y3, so.ld3 = ld3, so.plane = plane;
#define LVS(NV) hipLaunchKernelGGL((k_ln_fwd_v4<float, NV, false, true>), grid, blk, 0, s, x, ld_x, w, b, eps, y, ld_y, stats, rows, (int)C, LnDrop(), so)
  if (nv == 1) LVS(1); else if (nv == 2) LVS(2); else if (nv == 4) LVS(4); else if (nv == 5) LVS(5); else LVS(8);
#undef LVS
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

extern "C" int vfm_layernorm_fwd(const float* x, long ld_x, const float* w, const float* b, float eps, void* y, int y_dt,
                                 long ld_y, float* stats, long rows, long C, void* stream) {
  VFM_CHECK(C > 0 && C <= 3072 && ld_x >= C && ld_y >= C, VFM_E_SHAPE, "vfm_layernorm_fwd: C=%ld unsupported", C);
  if (rows == 0) return VFM_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(rows, 4)), blk(256);
  const int nvq = (int)(C / 256);
  const bool v4 = (C % 256 == 0) && (nvq == 1 || nvq == 2 || nvq == 4 || nvq == 5 || nvq == 8) && (ld_x % 4 == 0) && (ld_y % 4 == 0) && ((uintptr_t)x % 16 == 0) &&
                  ((uintptr_t)y % 8 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)b % 16 == 0);
  if (v4) {
#define LV(TO, NV) hipLaunchKernelGGL((k_ln_fwd_v4<TO, NV>), grid, blk, 0, s, x, ld_x, w, b, eps, (TO*)y, ld_y, stats, rows, (int)C)
    const int nv = (int)(C / 256);
    if (y_dt == VFM_BF16) { if (nv == 1) LV(bf16_t, 1); else if (nv == 4) LV(bf16_t, 4); else if (nv == 5) LV(bf16_t, 5); else if (nv == 2) LV(bf16_t, 2); else LV(bf16_t, 8); }
    else if (y_dt == VFM_F32) { if (nv == 1) LV(float, 1); else if (nv == 4) LV(float, 4); else if (nv == 5) LV(float, 5); else if (nv == 2) LV(float, 2); else LV(float, 8); }
    else VFM_FAIL(VFM_E_INVAL, "vfm_layernorm_fwd: dtype");
#undef LV
    if (!(nv == 1 || nv == 2 || nv == 4 || nv == 5 || nv == 8)) VFM_FAIL(VFM_E_SHAPE, "vfm_layernorm_fwd: C=%ld", C);
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  if (C > 1024 && C <= 2816 && ld_x % 4 == 0 && ld_y % 4 == 0 && ld_x >= (C + 3) / 4 * 4 && ld_y >= (C + 3) / 4 * 4 && (uintptr_t)x % 16 == 0 &&
      (uintptr_t)y % 16 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)b % 16 == 0) {   // padded pitch: 16-byte pieces, partly valid last piece (2.8 -> ~5 TB/s on EVA02's 2730-wide sub-LN)
    if (y_dt == VFM_BF16) hipLaunchKernelGGL((k_ln_fwd_v4t<bf16_t, 11>), grid, blk, 0, s, x, ld_x, w, b, eps, (bf16_t*)y, ld_y, stats, rows, (int)C);
    else if (y_dt == VFM_F32) hipLaunchKernelGGL((k_ln_fwd_v4t<float, 11>), grid, blk, 0, s, x, ld_x, w, b, eps, (float*)y, ld_y, stats, rows, (int)C);
    else VFM_FAIL(VFM_E_INVAL, "vfm_layernorm_fwd: dtype");
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  if (C > 1024 && C % 2 == 0 && ld_x % 2 == 0 && ld_y % 2 == 0 && (uintptr_t)x % 8 == 0 && (uintptr_t)y % 8 == 0 && (uintptr_t)w % 8 == 0 &&
      (uintptr_t)b % 8 == 0) {
    if (y_dt == VFM_BF16) hipLaunchKernelGGL((k_ln_fwd_v2<bf16_t, 24>), grid, blk, 0, s, x, ld_x, w, b, eps, (bf16_t*)y, ld_y, stats, rows, (int)C);
    else if (y_dt == VFM_F32) hipLaunchKernelGGL((k_ln_fwd_v2<float, 24>), grid, blk, 0, s, x, ld_x, w, b, eps, (float*)y, ld_y, stats, rows, (int)C);
    else VFM_FAIL(VFM_E_INVAL, "vfm_layernorm_fwd: dtype");
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
#define L(TO, PL) hipLaunchKernelGGL((k_ln_fwd<TO, PL>), grid, blk, 0, s, x, ld_x, w, b, eps, (TO*)y, ld_y, stats, rows, (int)C)
  const int pl = (int)((C + 63) / 64);
  if (y_dt == VFM_BF16) { if (pl <= 4) L(bf16_t, 4); else if (pl <= 16) L(bf16_t, 16); else if (pl <= 32) L(bf16_t, 32); else L(bf16_t, 48); }
  else if (y_dt == VFM_F32) { if (pl <= 4) L(float, 4); else if (pl <= 16) L(float, 16); else if (pl <= 32) L(float, 32); else L(float, 48); }
  else VFM_FAIL(VFM_E_INVAL, "vfm_layernorm_fwd: dtype");
#undef L
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

extern "C" int vfm_layernorm_dropout_fwd(const float* x, long ld_x, const float* w, const float* b, float eps, void* y, long ld_y,
                                         float* stats, void* y_drop, long ld_yd, void* mask, long ld_mask, float p, uint64_t seed,
                                         uint64_t offset, long rows, long C, void* stream) {
  VFM_CHECK(p >= 0.f && p < 1.f, VFM_E_INVAL, "vfm_layernorm_dropout_fwd: p");
  const long nv = C / 256;
  VFM_CHECK(C % 256 == 0 && (nv == 1 || nv == 2 || nv == 4 || nv == 5 || nv == 8), VFM_E_SHAPE, "vfm_layernorm_dropout_fwd: C=%ld", C);
  VFM_CHECK(ld_x % 4 == 0 && ld_y % 4 == 0 && ld_yd % 4 == 0 && ld_mask % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 8 == 0 &&
                (uintptr_t)y_drop % 8 == 0 && (uintptr_t)mask % 8 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)b % 16 == 0,
            VFM_E_ALIGN, "vfm_layernorm_dropout_fwd: alignment");
  if (rows == 0) return VFM_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 blk(256), grid(cdiv(rows, 4));
  LnDrop dr{y_drop, ld_yd, mask, ld_mask, p, 1.0f / (1.0f - p), seed, offset};
#define LVD(NV) hipLaunchKernelGGL((k_ln_fwd_v4<bf16_t, NV, true>), grid, blk, 0, s, x, ld_x, w, b, eps, (bf16_t*)y, ld_y, stats, rows, (int)C, dr)
  if (nv == 1) LVD(1); else if (nv == 2) LVD(2); else if (nv == 4) LVD(4); else if (nv == 5) LVD(5); else LVD(8);
#undef LVD
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// backward: one wave per row for dx; per-block partial dw/db in LDS -> ws[block][2][C]
template <typename TD, int MAXPL, bool NEED_W>
__global__ void k_ln_bwd(const TD* __restrict__ dy, long ld_dy, const float* __restrict__ x, long ld_x,
                         const float* __restrict__ w, const float* __restrict__ stats, float* __restrict__ dx, long ld_dx,
                         int accumulate_dx, float* __restrict__ ws, long rows, int C) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float pdw[NEED_W ? MAXPL : 1], pdb[NEED_W ? MAXPL : 1];
#pragma unroll
  for (int i = 0; i < (NEED_W ? MAXPL : 1); ++i) pdw[i] = pdb[i] = 0.f;
  for (long row = (long)blockIdx.x * nw + wv; row < rows; row += (long)gridDim.x * nw) {
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float g[MAXPL], xh[MAXPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
      const int c = i * 64 + lane;
      if (c < C) {
        const float d = ld_f32(dy + row * ld_dy + c);
        xh[i] = (x[row * ld_x + c] - mean) * rstd;
        g[i] = d * w[c];
        s1 += g[i];
        s2 += g[i] * xh[i];
        if constexpr (NEED_W) {
          pdw[i] += d * xh[i];
          pdb[i] += d;
        }
      } else {
        g[i] = xh[i] = 0.f;
      }
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
      const int c = i * 64 + lane;
      if (c < C) {
        const float v = rstd * (g[i] - s1 - xh[i] * s2);
        float* p = dx + row * ld_dx + c;
        *p = accumulate_dx ? *p + v : v;
      }
    }
  }
  if constexpr (NEED_W) {
    extern __shared__ float sh[];  // [nw][2][C]
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
      const int c = i * 64 + lane;
      if (c < C) {
        sh[(wv * 2 + 0) * C + c] = pdw[i];
        sh[(wv * 2 + 1) * C + c] = pdb[i];
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) {
      const int which = c / C, cc = c - which * C;
      float a = 0.f;
      for (int k = 0; k < nw; ++k) a += sh[(k * 2 + which) * C + cc];
      ws[((long)blockIdx.x * 2 + which) * C + cc] = a;
    }
  }
}
// block = 64 channels x 4 part-groups (a thread per channel looping over all 128 partials took 32 us)
__global__ void k_ln_bwd_fin(const float* __restrict__ ws, int parts, int C, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float sh[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float a = 0.f, b = 0.f;
  if (c < C) {
#pragma unroll 8   // independent loads in flight: a rolled loop is a chain of round trips (10 us per call)
    for (int p = ty; p < parts; p += 4) {
      a += ws[((long)p * 2 + 0) * C + c];
      b += ws[((long)p * 2 + 1) * C + c];
    }
  }
  sh[0][ty][tx] = a, sh[1][ty][tx] = b;
  __syncthreads();
  if (ty == 0 && c < C) {
    if (dw) dw[c] += (sh[0][0][tx] + sh[0][1][tx]) + (sh[0][2][tx] + sh[0][3][tx]);
    if (db) db[c] += (sh[1][0][tx] + sh[1][1][tx]) + (sh[1][2][tx] + sh[1][3][tx]);
  }
}

// padded-pitch form of the above (see k_ln_fwd_v4t): 16-byte pieces of x / dx, 8-byte pieces of a bf16 dy
template <typename TD, int MAXP4>
__global__ void k_ln_bwd_v4t(const TD* __restrict__ dy, long ld_dy, const float* __restrict__ x, long ld_x, const float* __restrict__ w,
                             const float* __restrict__ stats, float* __restrict__ dx, long ld_dx, int accumulate_dx, long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
  float g[MAXP4][4], xh[MAXP4][4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP4; ++i) {
    const int c = (i * 64 + lane) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[i][e] = xh[i][e] = 0.f;
    if (c < C) {
      float d[4];
      if constexpr (sizeof(TD) == 2) {
        const ushort4 pp = *reinterpret_cast<const ushort4*>(dy + row * ld_dy + c);
        d[0] = bf16_to_f32(pp.x), d[1] = bf16_to_f32(pp.y), d[2] = bf16_to_f32(pp.z), d[3] = bf16_to_f32(pp.w);
      } else {
        const float4 t = *reinterpret_cast<const float4*>(dy + row * ld_dy + c);
        d[0] = t.x, d[1] = t.y, d[2] = t.z, d[3] = t.w;
      }
      const float4 xv4 = *reinterpret_cast<const float4*>(x + row * ld_x + c);
      const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
      float wv[4] = {0.f, 0.f, 0.f, 0.f};
      if (c + 4 <= C) {
        const float4 ww = *reinterpret_cast<const float4*>(w + c);
        wv[0] = ww.x, wv[1] = ww.y, wv[2] = ww.z, wv[3] = ww.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < C) wv[e] = w[c + e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < C) {
          xh[i][e] = (xv[e] - mean) * rstd;
          g[i][e] = d[e] * wv[e];
          s1 += g[i][e];
          s2 += g[i][e] * xh[i][e];
        }
    }
  }
  s1 = wave_sum(s1) / C;
  s2 = wave_sum(s2) / C;
#pragma unroll
  for (int i = 0; i < MAXP4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rstd * (g[i][e] - s1 - xh[i][e] * s2);
      if (c + 4 <= C) {
        float4* pp = reinterpret_cast<float4*>(dx + row * ld_dx + c);
        if (accumulate_dx) {
          const float4 old = *pp;
          o[0] += old.x, o[1] += old.y, o[2] += old.z, o[3] += old.w;
        }
        *pp = make_float4(o[0], o[1], o[2], o[3]);
      } else {   // partly valid last piece: nothing is written past column C
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < C) dx[row * ld_dx + c + e] = accumulate_dx ? dx[row * ld_dx + c + e] + o[e] : o[e];
      }
    }
  }
}

static int ln_bwd_impl(const void* dy, int dy_dt, long ld_dy, const float* x, long ld_x, const float* w, const float* stats,
                       float* dx, long ld_dx, int accumulate_dx, float* dw, float* db, float* ws, long rows, long C, void* t_out,
                       long ld_t, const float* t_scale, void* stream);
extern "C" int vfm_layernorm_bwd(const void* dy, int dy_dt, long ld_dy, const float* x, long ld_x, const float* w,
                                 const float* stats, float* dx, long ld_dx, int accumulate_dx, float* dw, float* db,
                                 float* ws, long rows, long C, void* stream) {
  return ln_bwd_impl(dy, dy_dt, ld_dy, x, ld_x, w, stats, dx, ld_dx, accumulate_dx, dw, db, ws, rows, C, nullptr, 0, nullptr, stream);
}
extern "C" int vfm_layernorm_bwd_scaled(const void* dy, int dy_dt, long ld_dy, const float* x, long ld_x, const float* w,
                                        const float* stats, float* dx, long ld_dx, int accumulate_dx, void* t_out, long ld_t,
                                        const float* t_scale, long rows, long C, void* stream) {
  VFM_CHECK(t_out && t_scale && ld_t % 4 == 0 && (uintptr_t)t_out % 8 == 0 && (uintptr_t)t_scale % 16 == 0, VFM_E_INVAL,
            "vfm_layernorm_bwd_scaled: t_out / t_scale");
  return ln_bwd_impl(dy, dy_dt, ld_dy, x, ld_x, w, stats, dx, ld_dx, accumulate_dx, nullptr, nullptr, nullptr, rows, C, t_out, ld_t,
                     t_scale, stream);
}
// wide rows whose width is not a multiple of 256 (EVA02's SwiGLU sub-LN: C = 2730): each lane owns PAIRS of columns, so the row
// moves in 8-byte (fp32) / 4-byte (bf16) pieces instead of the scalar pieces of k_ln_bwd / k_ln_fwd.  C % 2 == 0.
template <typename TD, int MAXP2>
__global__ void k_ln_bwd_v2(const TD* __restrict__ dy, long ld_dy, const float* __restrict__ x, long ld_x, const float* __restrict__ w,
                            const float* __restrict__ stats, float* __restrict__ dx, long ld_dx, int accumulate_dx, long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
  float2 g[MAXP2], xh[MAXP2];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP2; ++i) {
    const int c = (i * 64 + lane) * 2;
    g[i] = xh[i] = make_float2(0.f, 0.f);
    if (c < C) {
      float2 d;
      if constexpr (sizeof(TD) == 2) {
        const ushort2 pp = *reinterpret_cast<const ushort2*>(dy + row * ld_dy + c);
        d = make_float2(bf16_to_f32(pp.x), bf16_to_f32(pp.y));
      } else {
        d = *reinterpret_cast<const float2*>(dy + row * ld_dy + c);
      }
      const float2 xv = *reinterpret_cast<const float2*>(x + row * ld_x + c);
      const float2 ww = *reinterpret_cast<const float2*>(w + c);
      xh[i] = make_float2((xv.x - mean) * rstd, (xv.y - mean) * rstd);
      g[i] = make_float2(d.x * ww.x, d.y * ww.y);
      s1 += g[i].x + g[i].y;
      s2 += g[i].x * xh[i].x + g[i].y * xh[i].y;
    }
  }
  s1 = wave_sum(s1) / C;
  s2 = wave_sum(s2) / C;
#pragma unroll
  for (int i = 0; i < MAXP2; ++i) {
    const int c = (i * 64 + lane) * 2;
    if (c < C) {
      float2 o = make_float2(rstd * (g[i].x - s1 - xh[i].x * s2), rstd * (g[i].y - s1 - xh[i].y * s2));
      float2* pp = reinterpret_cast<float2*>(dx + row * ld_dx + c);
      if (accumulate_dx) {
        const float2 old = *pp;
        o.x += old.x, o.y += old.y;
      }
      *pp = o;
    }
  }
}

static int ln_bwd_impl(const void* dy, int dy_dt, long ld_dy, const float* x, long ld_x, const float* w, const float* stats,
                       float* dx, long ld_dx, int accumulate_dx, float* dw, float* db, float* ws, long rows, long C, void* t_out,
                       long ld_t, const float* t_scale, void* stream) {
  VFM_CHECK(C > 0 && (C <= 1024 || (C <= 3072 && !(dw || db))), VFM_E_SHAPE, "vfm_layernorm_bwd: C=%ld unsupported", C);
  VFM_CHECK(!(dw || db) || ws, VFM_E_INVAL, "vfm_layernorm_bwd: ws required for dw/db");
  if (rows == 0) return VFM_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool need_w = (dw || db);
  const int nv = (int)(C / 256);
  const bool v4 = !need_w && (C % 256 == 0) && (nv == 1 || nv == 2 || nv == 4 || nv == 5) && (ld_x % 4 == 0) && (ld_dx % 4 == 0) &&
                  (ld_dy % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dx % 16 == 0) && ((uintptr_t)dy % 8 == 0) &&
                  ((uintptr_t)w % 16 == 0);
  if (v4) {
    dim3 grid(cdiv(rows, 4)), blk(256);
#define LV(TD, NV) hipLaunchKernelGGL((k_ln_bwd_v4<TD, NV>), grid, blk, 0, s, (const TD*)dy, ld_dy, x, ld_x, w, stats, dx, ld_dx, accumulate_dx, rows, (int)C, (bf16_t*)t_out, ld_t, t_scale)
    if (dy_dt == VFM_BF16) { if (nv == 1) LV(bf16_t, 1); else if (nv == 2) LV(bf16_t, 2); else if (nv == 4) LV(bf16_t, 4); else LV(bf16_t, 5); }
    else if (dy_dt == VFM_F32) { if (nv == 1) LV(float, 1); else if (nv == 2) LV(float, 2); else if (nv == 4) LV(float, 4); else LV(float, 5); }
    else VFM_FAIL(VFM_E_INVAL, "vfm_layernorm_bwd: dtype");
#undef LV
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  VFM_CHECK(!t_out, VFM_E_UNSUPPORTED, "vfm_layernorm_bwd_scaled: needs the vectorised path (C %% 256 == 0, aligned operands, no dw/db)");
  {
    const long cp = (C + 3) / 4 * 4;
    if (!need_w && C > 1024 && C <= 2816 && ld_x % 4 == 0 && ld_dx % 4 == 0 && ld_dy % 4 == 0 && ld_x >= cp && ld_dx >= cp && ld_dy >= cp &&
        (uintptr_t)x % 16 == 0 && (uintptr_t)dx % 16 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)w % 16 == 0) {
      dim3 grid(cdiv(rows, 4)), blk(256);
      if (dy_dt == VFM_BF16)
        hipLaunchKernelGGL((k_ln_bwd_v4t<bf16_t, 11>), grid, blk, 0, s, (const bf16_t*)dy, ld_dy, x, ld_x, w, stats, dx, ld_dx, accumulate_dx, rows, (int)C);
      else if (dy_dt == VFM_F32)
        hipLaunchKernelGGL((k_ln_bwd_v4t<float, 11>), grid, blk, 0, s, (const float*)dy, ld_dy, x, ld_x, w, stats, dx, ld_dx, accumulate_dx, rows, (int)C);
      else VFM_FAIL(VFM_E_INVAL, "vfm_layernorm_bwd: dtype");
      VFM_LAUNCH_CHECK();
      return VFM_OK;
    }
  }
  if (!need_w && C > 1024 && C % 2 == 0 && ld_x % 2 == 0 && ld_dx % 2 == 0 && ld_dy % 2 == 0 && (uintptr_t)x % 8 == 0 &&
      (uintptr_t)dx % 8 == 0 && (uintptr_t)dy % 8 == 0 && (uintptr_t)w % 8 == 0) {
    dim3 grid(cdiv(rows, 4)), blk(256);
    if (dy_dt == VFM_BF16)
      hipLaunchKernelGGL((k_ln_bwd_v2<bf16_t, 24>), grid, blk, 0, s, (const bf16_t*)dy, ld_dy, x, ld_x, w, stats, dx, ld_dx, accumulate_dx, rows, (int)C);
    else if (dy_dt == VFM_F32)
      hipLaunchKernelGGL((k_ln_bwd_v2<float, 24>), grid, blk, 0, s, (const float*)dy, ld_dy, x, ld_x, w, stats, dx, ld_dx, accumulate_dx, rows, (int)C);
    else VFM_FAIL(VFM_E_INVAL, "vfm_layernorm_bwd: dtype");
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  const int parts = need_w ? 128 : cdiv(rows, 4);
  const size_t shm = need_w ? (size_t)4 * 2 * C * sizeof(float) : 0;
  float* wsp = need_w ? ws : nullptr;
#define L(TD, PL, NW) hipLaunchKernelGGL((k_ln_bwd<TD, PL, NW>), dim3(parts), dim3(256), shm, s, (const TD*)dy, ld_dy, x, ld_x, w, stats, dx, ld_dx, accumulate_dx, wsp, rows, (int)C)
  const int pl = (int)((C + 63) / 64);
  if (need_w) {
    if (dy_dt == VFM_BF16) { if (pl <= 4) L(bf16_t, 4, true); else L(bf16_t, 16, true); }
    else if (dy_dt == VFM_F32) { if (pl <= 4) L(float, 4, true); else L(float, 16, true); }
    else VFM_FAIL(VFM_E_INVAL, "vfm_layernorm_bwd: dtype");
  } else {
    if (dy_dt == VFM_BF16) { if (pl <= 4) L(bf16_t, 4, false); else if (pl <= 16) L(bf16_t, 16, false); else L(bf16_t, 48, false); }
    else if (dy_dt == VFM_F32) { if (pl <= 4) L(float, 4, false); else if (pl <= 16) L(float, 16, false); else L(float, 48, false); }
    else VFM_FAIL(VFM_E_INVAL, "vfm_layernorm_bwd: dtype");
  }
#undef L
  if (need_w) hipLaunchKernelGGL(k_ln_bwd_fin, dim3(cdiv(C, 64)), dim3(256), 0, s, ws, parts, (int)C, dw, db);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// =============================================================================================== per-channel partial moments
// Shared by GroupNorm and BatchNorm.  x viewed as [B][P][C]; block (bx: 64-channel strip, by: row chunk, bz: image)
// writes ws[(b*nchunk + chunk)][2][C] = (sum_rows v1, sum_rows v2).
// MODE 0: v1 = x, v2 = x*x.  MODE 1 (backward): v1 = dz, v2 = dz*xhat, where dz = dy*act'(xhat*w+b) and
// (mean,rstd) come per (b, group) [GN: stats[b][g][2]] or per channel [BN: stats[2][C] = (mean, var), G==0].
template <int MODE, typename TD>
__global__ void k_chan_moments(const float* __restrict__ x, const TD* __restrict__ dy, const float* __restrict__ w,
                               const float* __restrict__ bb, const float* __restrict__ stats, int G, float eps, int act,
                               float* __restrict__ ws, long P, int C, int rows_per_chunk) {
  __shared__ float sh[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const long b = blockIdx.z;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  long r1 = r0 + rows_per_chunk;
  if (r1 > P) r1 = P;
  float a1 = 0.f, a2 = 0.f;
  if (c < C) {
    float mean = 0.f, rstd = 1.f, wc = 1.f, bc = 0.f;
    if (MODE == 1) {
      if (G > 0) {
        const int g = c / (C / G);
        mean = stats[(b * G + g) * 2];
        rstd = stats[(b * G + g) * 2 + 1];
      } else {
        mean = stats[c];
        rstd = rsqrtf(stats[C + c] + eps);
      }
      wc = w[c];
      bc = bb[c];
    }
    const float* xp = x + (b * P) * C + c;
    for (long r = r0 + ty; r < r1; r += 4) {
      const float xv = xp[r * C];
      if (MODE == 0) {
        a1 += xv;
        a2 += xv * xv;
      } else {
        const float xh = (xv - mean) * rstd;
        float d = ld_f32(dy + (b * P + r) * C + c);
        if (act != VFM_ACT_NONE) d *= act_grad_f(xh * wc + bc, act);
        a1 += d;
        a2 += d * xh;
      }
    }
  }
  sh[0][ty][tx] = a1;
  sh[1][ty][tx] = a2;
  __syncthreads();
  if (ty == 0 && c < C) {
    const long slot = b * gridDim.y + blockIdx.y;
    ws[(slot * 2 + 0) * C + c] = sh[0][0][tx] + sh[0][1][tx] + sh[0][2][tx] + sh[0][3][tx];
    ws[(slot * 2 + 1) * C + c] = sh[1][0][tx] + sh[1][1][tx] + sh[1][2][tx] + sh[1][3][tx];
  }
}

static inline int pick_chunks(long P) {
  int n = (int)((P + 63) / 64);
  if (n > 64) n = 64;
  if (n < 1) n = 1;
  return n;
}

// =============================================================================================== GroupNorm
// finalize forward: one thread per (b,g): combine channel partials in double
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// one wave per (image, group): the lanes split the nchunk x (C/G) partial sums (a single thread looping over them took 25 us)
__global__ void k_gn_fin_fwd(const float* __restrict__ ws, int nchunk, int C, int G, long P, float eps,
                             float* __restrict__ stats, int BG) {
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= BG) return;
  const int b = i / G, g = i - b * G, cg = C / G;
  double s = 0.0, q = 0.0;
  for (int e = lane; e < nchunk * cg; e += 64) {
    const int k = e / cg, j = e - k * cg;
    const long slot = (long)b * nchunk + k;
    s += ws[(slot * 2 + 0) * C + g * cg + j];
    q += ws[(slot * 2 + 1) * C + g * cg + j];
  }
  s = wave_sum_f64(s), q = wave_sum_f64(q);
  if (lane) return;
  const double n = (double)P * cg;
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0) var = 0;
  stats[i * 2] = (float)mean;
  stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}
template <typename TO>
__global__ void k_gn_apply(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                           const float* __restrict__ stats, int G, int act, TO* __restrict__ y, long B, long P, int C) {
  const long total = B * P * C;
  const int cg = C / G;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long bi = i / ((long)P * C);
    const int g = c / cg;
    const float mean = stats[(bi * G + g) * 2], rstd = stats[(bi * G + g) * 2 + 1];
    st_f32(y + i, act_f((x[i] - mean) * rstd * w[c] + b[c], act));
  }
}
extern "C" int vfm_groupnorm_fwd(const float* x, const float* w, const float* b, float eps, int G, int act, void* y, int y_dt,
                                 float* stats, float* ws_in, long B, long P, long C, void* stream) {
  VFM_CHECK(G > 0 && C % G == 0 && ws_in, VFM_E_SHAPE, "vfm_groupnorm_fwd: C %% G / ws");
  if (B * P * C == 0) return VFM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = pick_chunks(P);
  const int rpc = (int)((P + nchunk - 1) / nchunk);
  float* ws = ws_in + B * G * 2;
  hipLaunchKernelGGL((k_chan_moments<0, float>), dim3(cdiv(C, 64), nchunk, (unsigned)B), dim3(256), 0, s, x, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, 0.f, 0, ws, P, (int)C, rpc);
  hipLaunchKernelGGL(k_gn_fin_fwd, dim3(cdiv(B * G, 4)), dim3(256), 0, s, ws, nchunk, (int)C, G, P, eps, stats, (int)(B * G));
  const long total = B * P * C;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  if (y_dt == VFM_BF16) hipLaunchKernelGGL(k_gn_apply<bf16_t>, dim3(grid), dim3(256), 0, s, x, w, b, stats, G, act, (bf16_t*)y, B, P, (int)C);
  else if (y_dt == VFM_F32) hipLaunchKernelGGL(k_gn_apply<float>, dim3(grid), dim3(256), 0, s, x, w, b, stats, G, act, (float*)y, B, P, (int)C);
  else VFM_FAIL(VFM_E_INVAL, "vfm_groupnorm_fwd: dtype");
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// backward finalize: per (b,g): s1 = sum_c w*sum dz, s2 = sum_c w*sum dz*xhat -> gs[b][g][2]; and dw/db accumulate
__global__ void k_gn_fin_bwd(const float* __restrict__ ws, int nchunk, int C, int G, const float* __restrict__ w,
                             float* __restrict__ gs, int BG) {
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= BG) return;
  const int b = i / G, g = i - b * G, cg = C / G;
  double s1 = 0.0, s2 = 0.0;
  for (int e = lane; e < nchunk * cg; e += 64) {
    const int k = e / cg, j = e - k * cg;
    const long slot = (long)b * nchunk + k;
    const int c = g * cg + j;
    s1 += (double)w[c] * ws[(slot * 2 + 0) * C + c];
    s2 += (double)w[c] * ws[(slot * 2 + 1) * C + c];
  }
  s1 = wave_sum_f64(s1), s2 = wave_sum_f64(s2);
  if (lane) return;
  gs[i * 2] = (float)s1;
  gs[i * 2 + 1] = (float)s2;
}
__global__ void k_chan_fin_wb(const float* __restrict__ ws, int slots, int C, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float sh[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float a = 0.f, b = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int k = ty; k < slots; k += 4) {
      b += ws[((long)k * 2 + 0) * C + c];
      a += ws[((long)k * 2 + 1) * C + c];
    }
  }
  sh[0][ty][tx] = a, sh[1][ty][tx] = b;
  __syncthreads();
  if (ty == 0 && c < C) {
    if (dw) dw[c] += (sh[0][0][tx] + sh[0][1][tx]) + (sh[0][2][tx] + sh[0][3][tx]);
    if (db) db[c] += (sh[1][0][tx] + sh[1][1][tx]) + (sh[1][2][tx] + sh[1][3][tx]);
  }
}
template <typename TD>
__global__ void k_gn_bwd_apply(const TD* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w,
                               const float* __restrict__ b, const float* __restrict__ stats, const float* __restrict__ gs,
                               int G, int act, float* __restrict__ dx, long B, long P, int C) {
  const long total = B * P * C;
  const int cg = C / G;
  const float inv_n = 1.0f / ((float)P * cg);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long bi = i / ((long)P * C);
    const int g = c / cg;
    const float mean = stats[(bi * G + g) * 2], rstd = stats[(bi * G + g) * 2 + 1];
    const float xh = (x[i] - mean) * rstd;
    float d = ld_f32(dy + i);
    if (act != VFM_ACT_NONE) d *= act_grad_f(xh * w[c] + b[c], act);
    dx[i] = rstd * (d * w[c] - (gs[(bi * G + g) * 2] + xh * gs[(bi * G + g) * 2 + 1]) * inv_n);
  }
}
extern "C" int vfm_groupnorm_bwd(const void* dy, int dy_dt, const float* x, const float* w, const float* b,
                                 const float* stats, int G, int act, float* dx, float* dw, float* db, float* ws, long B,
                                 long P, long C, void* stream) {
  VFM_CHECK(G > 0 && C % G == 0 && ws, VFM_E_SHAPE, "vfm_groupnorm_bwd: args");
  if (B * P * C == 0) return VFM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = pick_chunks(P);
  const int rpc = (int)((P + nchunk - 1) / nchunk);
  float* gs = ws;                 // [B,G,2]
  float* part = ws + B * G * 2;   // [B*nchunk][2][C]
  dim3 grid(cdiv(C, 64), nchunk, (unsigned)B);
  if (dy_dt == VFM_BF16)
    hipLaunchKernelGGL((k_chan_moments<1, bf16_t>), grid, dim3(256), 0, s, x, (const bf16_t*)dy, w, b, stats, G, 0.f, act, part, P, (int)C, rpc);
  else if (dy_dt == VFM_F32)
    hipLaunchKernelGGL((k_chan_moments<1, float>), grid, dim3(256), 0, s, x, (const float*)dy, w, b, stats, G, 0.f, act, part, P, (int)C, rpc);
  else VFM_FAIL(VFM_E_INVAL, "vfm_groupnorm_bwd: dtype");
  hipLaunchKernelGGL(k_gn_fin_bwd, dim3(cdiv(B * G, 4)), dim3(256), 0, s, part, nchunk, (int)C, G, w, gs, (int)(B * G));
  if (dw || db) hipLaunchKernelGGL(k_chan_fin_wb, dim3(cdiv(C, 64)), dim3(256), 0, s, part, (int)(B * nchunk), (int)C, dw, db);
  const long total = B * P * C;
  const int g2 = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  if (dy_dt == VFM_BF16) hipLaunchKernelGGL(k_gn_bwd_apply<bf16_t>, dim3(g2), dim3(256), 0, s, (const bf16_t*)dy, x, w, b, stats, gs, G, act, dx, B, P, (int)C);
  else hipLaunchKernelGGL(k_gn_bwd_apply<float>, dim3(g2), dim3(256), 0, s, (const float*)dy, x, w, b, stats, gs, G, act, dx, B, P, (int)C);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// =============================================================================================== BatchNorm
__global__ void __launch_bounds__(256) k_bn_fin_sums(const float* __restrict__ ws, int slots, int C, float* __restrict__ sums) {
  // 64 channels x 4 slot groups per block, fixed combination order (the one-thread-per-channel walk over all slots was a chain of
  // dependent loads: 17 us per call)
  __shared__ double sh[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  double a = 0.0, b = 0.0;
  if (c < C) {
#pragma unroll 8
    for (int k = ty; k < slots; k += 4) {
      a += ws[((long)k * 2 + 0) * C + c];
      b += ws[((long)k * 2 + 1) * C + c];
    }
  }
  sh[0][ty][tx] = a, sh[1][ty][tx] = b;
  __syncthreads();
  if (ty == 0 && c < C) {
    sums[c] = (float)((sh[0][0][tx] + sh[0][1][tx]) + (sh[0][2][tx] + sh[0][3][tx]));
    sums[C + c] = (float)((sh[1][0][tx] + sh[1][1][tx]) + (sh[1][2][tx] + sh[1][3][tx]));
  }
}
extern "C" int vfm_bn_moments(const float* x, long rows, long C, float* sums, float* ws, void* stream) {
  VFM_CHECK(ws && sums, VFM_E_INVAL, "vfm_bn_moments: args");
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = pick_chunks(rows);
  const int rpc = (int)((rows + nchunk - 1) / nchunk);
  hipLaunchKernelGGL((k_chan_moments<0, float>), dim3(cdiv(C, 64), nchunk, 1), dim3(256), 0, s, x, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, 0.f, 0, ws, rows, (int)C, rpc);
  hipLaunchKernelGGL(k_bn_fin_sums, dim3(cdiv(C, 64)), dim3(256), 0, s, ws, nchunk, (int)C, sums);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
// (sum, sumsq, count) -> mean_var (biased) and running-stat update (momentum, unbiased var) as nn.SyncBatchNorm does
__global__ void k_bn_finalize(const float* __restrict__ sums, float count, float* __restrict__ mean_var,
                              float* __restrict__ rmean, float* __restrict__ rvar, float momentum, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = (double)sums[c] / count;
  double var = (double)sums[C + c] / count - mean * mean;
  if (var < 0) var = 0;
  mean_var[c] = (float)mean;
  mean_var[C + c] = (float)var;
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * (count / fmaxf(count - 1.f, 1.f)));
}
extern "C" int vfm_bn_finalize(const float* sums, float count, float* mean_var, float* running_mean, float* running_var,
                               float momentum, long C, void* stream) {
  hipLaunchKernelGGL(k_bn_finalize, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, count, mean_var, running_mean,
                     running_var, momentum, (int)C);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
template <typename TO>
__global__ void k_bn_apply(const float* __restrict__ x, const float* __restrict__ mv, const float* __restrict__ w,
                           const float* __restrict__ b, float eps, int act, TO* __restrict__ y, long rows, int C) {
  const long total = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    st_f32(y + i, act_f((x[i] - mv[c]) * rsqrtf(mv[C + c] + eps) * w[c] + b[c], act));
  }
}
extern "C" int vfm_bn_apply(const float* x, const float* mean_var, const float* w, const float* b, float eps, int act, void* y,
                            int y_dt, long rows, long C, void* stream) {
  const long total = rows * C;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (y_dt == VFM_BF16) hipLaunchKernelGGL(k_bn_apply<bf16_t>, dim3(grid), dim3(256), 0, s, x, mean_var, w, b, eps, act, (bf16_t*)y, rows, (int)C);
  else if (y_dt == VFM_F32) hipLaunchKernelGGL(k_bn_apply<float>, dim3(grid), dim3(256), 0, s, x, mean_var, w, b, eps, act, (float*)y, rows, (int)C);
  else VFM_FAIL(VFM_E_INVAL, "vfm_bn_apply: dtype");
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
extern "C" int vfm_bn_bwd_reduce(const void* dy, int dy_dt, const float* x, const float* mean_var, const float* w,
                                 const float* b, float eps, int act, float* sums_dy, float* ws, long rows, long C,
                                 void* stream) {
  VFM_CHECK(ws && sums_dy, VFM_E_INVAL, "vfm_bn_bwd_reduce: args");
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = pick_chunks(rows);
  const int rpc = (int)((rows + nchunk - 1) / nchunk);
  dim3 grid(cdiv(C, 64), nchunk, 1);
  if (dy_dt == VFM_BF16)
    hipLaunchKernelGGL((k_chan_moments<1, bf16_t>), grid, dim3(256), 0, s, x, (const bf16_t*)dy, w, b, mean_var, 0, eps, act, ws, rows, (int)C, rpc);
  else if (dy_dt == VFM_F32)
    hipLaunchKernelGGL((k_chan_moments<1, float>), grid, dim3(256), 0, s, x, (const float*)dy, w, b, mean_var, 0, eps, act, ws, rows, (int)C, rpc);
  else VFM_FAIL(VFM_E_INVAL, "vfm_bn_bwd_reduce: dtype");
  hipLaunchKernelGGL(k_bn_fin_sums, dim3(cdiv(C, 64)), dim3(256), 0, s, ws, nchunk, (int)C, sums_dy);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
template <typename TD>
__global__ void k_bn_bwd_apply(const TD* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mv,
                               const float* __restrict__ w, const float* __restrict__ b, float eps, int act,
                               const float* __restrict__ sums, float inv_n, float* __restrict__ dx, long rows, int C) {
  const long total = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const float rstd = rsqrtf(mv[C + c] + eps);
    const float xh = (x[i] - mv[c]) * rstd;
    float d = ld_f32(dy + i);
    if (act != VFM_ACT_NONE) d *= act_grad_f(xh * w[c] + b[c], act);
    dx[i] = rstd * w[c] * (d - (sums[c] + xh * sums[C + c]) * inv_n);
  }
}
extern "C" int vfm_bn_bwd_apply(const void* dy, int dy_dt, const float* x, const float* mean_var, const float* w,
                                const float* b, float eps, int act, const float* sums_dy, float total_rows, float* dx,
                                long rows, long C, void* stream) {
  const long total = rows * C;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  const float inv_n = 1.0f / total_rows;
  if (dy_dt == VFM_BF16) hipLaunchKernelGGL(k_bn_bwd_apply<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, x, mean_var, w, b, eps, act, sums_dy, inv_n, dx, rows, (int)C);
  else if (dy_dt == VFM_F32) hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, x, mean_var, w, b, eps, act, sums_dy, inv_n, dx, rows, (int)C);
  else VFM_FAIL(VFM_E_INVAL, "vfm_bn_bwd_apply: dtype");
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
