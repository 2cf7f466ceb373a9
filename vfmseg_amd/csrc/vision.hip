// Image-side kernels: patchify, token assembly, bilinear/nearest resize, fused upsample+CE, sliding-window
// inference helpers, fused AdamW.  All HBM-bound; fp32 arithmetic follows ATen's formulas so results match the
// reference's F.interpolate / F.cross_entropy to rounding.
#include "common.h"

// ---------------------------------------------------------------------------------------------------- patchify
template <typename TO>
__global__ void k_patchify(const float* __restrict__ img, long sb, long sc, long sy, int y0, int x0, int nh, int nw, int P,
                           TO* __restrict__ out, long ld_out, int B) {
  const long K = 3L * P * P;
  const long total = (long)B * nh * nw * K;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / K;
    const int k = (int)(i - row * K);
    const int c = k / (P * P), r = k - c * P * P, iy = r / P, ix = r - iy * P;
    const int px = (int)(row % nw);
    const long t = row / nw;
    const int py = (int)(t % nh);
    const long b = t / nh;
    st_f32(out + row * ld_out + k, img[b * sb + c * sc + (long)(y0 + py * P + iy) * sy + (x0 + px * P + ix)]);
  }
}
extern "C" int vfm_patchify(const float* img, long stride_b, long stride_c, long stride_y, int y0, int x0, int h, int w, int P,
                            void* out, int out_dt, long ld_out, int B, void* stream) {
  VFM_CHECK(P > 0 && h % P == 0 && w % P == 0, VFM_E_SHAPE, "vfm_patchify: %dx%d not a multiple of patch %d", h, w, P);
  VFM_CHECK(ld_out >= 3L * P * P, VFM_E_SHAPE, "vfm_patchify: ld_out");
  const int nh = h / P, nw = w / P;
  const long total = (long)B * nh * nw * 3 * P * P;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (out_dt == VFM_BF16) hipLaunchKernelGGL(k_patchify<bf16_t>, dim3(grid), dim3(256), 0, s, img, stride_b, stride_c, stride_y, y0, x0, nh, nw, P, (bf16_t*)out, ld_out, B);
  else if (out_dt == VFM_F32) hipLaunchKernelGGL(k_patchify<float>, dim3(grid), dim3(256), 0, s, img, stride_b, stride_c, stride_y, y0, x0, nh, nw, P, (float*)out, ld_out, B);
  else VFM_FAIL(VFM_E_INVAL, "vfm_patchify: dtype");
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ---------------------------------------------------------------------------------------------------- tokens
__global__ void k_assemble_tokens(const float* __restrict__ ptok, const float* __restrict__ cls, const float* __restrict__ pos,
                                  float* __restrict__ x, int B, int np, int C) {
  const long total = ((long)B * np + B) * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C;
    const int c = (int)(i - row * C);
    if (row < (long)B * np) {
      const int t = (int)(row % np);
      x[i] = ptok[i] + pos[(long)(1 + t) * C + c];
    } else {
      x[i] = cls[c] + pos[c];
    }
  }
}
extern "C" int vfm_assemble_tokens(const float* patch_tok, const float* cls, const float* pos, float* x, int B, int np, int C,
                                   void* stream) {
  const long total = ((long)B * np + B) * C;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_assemble_tokens, dim3(grid), dim3(256), 0, (hipStream_t)stream, patch_tok, cls, pos, x, B, np, C);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ---------------------------------------------------------------------------------------------------- bilinear
struct Lerp {
  int i0, i1;
  float l0, l1;
};
// ATen area_pixel_compute_source_index (align_corners=False, cubic=false) + guard
__device__ __forceinline__ Lerp lerp_idx(int dst, float scale, int in_size) {
  float src = scale * (dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Lerp r;
  r.i0 = (int)src;
  if (r.i0 > in_size - 1) r.i0 = in_size - 1;
  r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
  r.l1 = src - r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}
__device__ __forceinline__ int blocked2(int y, int x, int wc) {
  // 2-level 2x2 blocked pixel order: (y/4, x/4, (y/2)%2, (x/2)%2, y%2, x%2)
  return ((((y >> 2) * (wc >> 2) + (x >> 2)) * 2 + ((y >> 1) & 1)) * 2 + ((x >> 1) & 1)) * 4 + ((y & 1) << 1) + (x & 1);
}

__global__ void k_resize_bilinear(const void* __restrict__ in, int in_dt, int in_nchw, int B, int Hi, int Wi, int C,
                                  long in_ld, void* __restrict__ out, int out_dt, int out_mode, long out_ld, float sy,
                                  float sx, int y0, int x0, int hc, int wc) {
  const long total = (long)B * hc * wc * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c, x, y;
    long b;
    if (out_mode == 1) {  // NCHW: x fastest
      x = (int)(i % wc);
      long t = i / wc;
      y = (int)(t % hc);
      t /= hc;
      c = (int)(t % C);
      b = t / C;
    } else {  // NHWC: c fastest
      c = (int)(i % C);
      long t = i / C;
      x = (int)(t % wc);
      t /= wc;
      y = (int)(t % hc);
      b = t / hc;
    }
    const Lerp ly = lerp_idx(y + y0, sy, Hi), lx = lerp_idx(x + x0, sx, Wi);
    float v00, v01, v10, v11;
    if (in_nchw) {
      const long base = (b * C + c) * (long)Hi * Wi;
      v00 = ld_any(in, base + (long)ly.i0 * Wi + lx.i0, in_dt);
      v01 = ld_any(in, base + (long)ly.i0 * Wi + lx.i1, in_dt);
      v10 = ld_any(in, base + (long)ly.i1 * Wi + lx.i0, in_dt);
      v11 = ld_any(in, base + (long)ly.i1 * Wi + lx.i1, in_dt);
    } else {
      const long base = b * (long)Hi * Wi;
      v00 = ld_any(in, (base + (long)ly.i0 * Wi + lx.i0) * in_ld + c, in_dt);
      v01 = ld_any(in, (base + (long)ly.i0 * Wi + lx.i1) * in_ld + c, in_dt);
      v10 = ld_any(in, (base + (long)ly.i1 * Wi + lx.i0) * in_ld + c, in_dt);
      v11 = ld_any(in, (base + (long)ly.i1 * Wi + lx.i1) * in_ld + c, in_dt);
    }
    const float v = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
    long o;
    if (out_mode == 1) o = i;
    else if (out_mode == 0) o = ((b * hc + y) * wc + x) * out_ld + c;
    else {
      const int p = blocked2(y, x, wc);
      o = (b * ((long)hc * wc / 4) + (p >> 2)) * out_ld + (long)(p & 3) * C + c;
    }
    st_any(out, o, out_dt, v);
  }
}
extern "C" int vfm_resize_bilinear(const void* in, int in_dt, int in_nchw, int B, int Hi, int Wi, int C, long in_ld_c,
                                   void* out, int out_dt, int out_mode, long out_ld_c, int Hv, int Wv, int y0, int x0, int hc,
                                   int wc, void* stream) {
  VFM_CHECK(Hv > 0 && Wv > 0 && y0 >= 0 && x0 >= 0 && y0 + hc <= Hv && x0 + wc <= Wv, VFM_E_SHAPE, "vfm_resize_bilinear: window");
  VFM_CHECK(out_mode != 2 || (hc % 4 == 0 && wc % 4 == 0 && out_ld_c >= 4L * C), VFM_E_SHAPE, "vfm_resize_bilinear: blocked mode needs hc,wc %% 4");
  const long total = (long)B * hc * wc * C;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipLaunchKernelGGL(k_resize_bilinear, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, in_dt, in_nchw, B, Hi, Wi, C, in_ld_c,
                     out, out_dt, out_mode, out_ld_c, (float)Hi / (float)Hv, (float)Wi / (float)Wv, y0, x0, hc, wc);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ---------------------------------------------------------------------------------------------------- bicubic
// ATen upsample_bicubic2d (align_corners=False, A=-0.75): src = scale*(dst+0.5)-0.5 (not clamped), 4x4 taps with indices
// clamped to the border.  Used once per token grid to re-interpolate the frozen DINOv2 pos-embed (dino_v2.py:184-215).
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
  w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}
__global__ void k_resize_bicubic(const float* __restrict__ in, int Hi, int Wi, int C, float* __restrict__ out, int Ho, int Wo,
                                 float sy, float sx) {
  const long total = (long)Ho * Wo * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int x = (int)(t % Wo);
    const int y = (int)(t / Wo);
    const float fy = sy * (y + 0.5f) - 0.5f, fx = sx * (x + 0.5f) - 0.5f;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float wy[4], wx[4];
    cubic_coeffs(fy - iy, wy);
    cubic_coeffs(fx - ix, wx);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int yy = iy - 1 + a;
      yy = yy < 0 ? 0 : (yy > Hi - 1 ? Hi - 1 : yy);
      float row = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        int xx = ix - 1 + b;
        xx = xx < 0 ? 0 : (xx > Wi - 1 ? Wi - 1 : xx);
        row += wx[b] * in[((long)yy * Wi + xx) * C + c];
      }
      acc += wy[a] * row;
    }
    out[i] = acc;
  }
}
extern "C" int vfm_resize_bicubic(const float* in, int Hi, int Wi, int C, float* out, int Ho, int Wo, float scale_y,
                                  float scale_x, void* stream) {
  const long total = (long)Ho * Wo * C;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_resize_bicubic, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, Hi, Wi, C, out, Ho, Wo, scale_y, scale_x);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ---------------------------------------------------------------------------------------------------- labels
__global__ void k_label_resize(const int64_t* __restrict__ in, int B, int Hi, int Wi, int64_t* __restrict__ out, float sy,
                               float sx, int y0, int x0, int hc, int wc) {
  const long total = (long)B * hc * wc;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % wc);
    long t = i / wc;
    const int y = (int)(t % hc);
    const long b = t / hc;
    int iy = (int)floorf((y + y0) * sy), ix = (int)floorf((x + x0) * sx);  // nearest_neighbor_compute_source_index
    if (iy > Hi - 1) iy = Hi - 1;
    if (ix > Wi - 1) ix = Wi - 1;
    out[i] = in[(b * Hi + iy) * (long)Wi + ix];
  }
}
extern "C" int vfm_label_resize(const int64_t* in, int B, int Hi, int Wi, int64_t* out, int Hv, int Wv, int y0, int x0, int hc,
                                int wc, void* stream) {
  VFM_CHECK(y0 >= 0 && x0 >= 0 && y0 + hc <= Hv && x0 + wc <= Wv, VFM_E_SHAPE, "vfm_label_resize: window");
  const long total = (long)B * hc * wc;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_label_resize, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, B, Hi, Wi, out, (float)Hi / (float)Hv,
                     (float)Wi / (float)Wv, y0, x0, hc, wc);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ---------------------------------------------------------------------------------------------------- unblock
__device__ __forceinline__ long blocked_index(int y, int x, int H, int W, int levels) {
  // levels of 2x2 blocking; coarsest cell is (y >> levels, x >> levels) in raster order
  long p = (long)(y >> levels) * (W >> levels) + (x >> levels);
  for (int l = levels - 1; l >= 0; --l) p = (p * 2 + ((y >> l) & 1)) * 2 + ((x >> l) & 1);
  return p;
}
__global__ void k_unblock(const float* __restrict__ xin, float* __restrict__ yout, int B, int H, int W, int C, int levels,
                          int inverse) {
  const long total = (long)B * H * W * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const long b = t / H;
    const long pb = (b * (long)H * W + blocked_index(y, x, H, W, levels)) * C + c;
    if (inverse) yout[pb] = xin[i];
    else yout[i] = xin[pb];
  }
}
extern "C" int vfm_unblock(const float* x, float* y, int B, int H, int W, int C, int levels, int inverse, void* stream) {
  VFM_CHECK(levels >= 0 && H % (1 << levels) == 0 && W % (1 << levels) == 0, VFM_E_SHAPE, "vfm_unblock: H,W vs levels");
  const long total = (long)B * H * W * C;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_unblock, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C, levels, inverse);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ---------------------------------------------------------------------------------------------------- upsample + CE
// One wave per LOW-resolution pixel. It visits every high-resolution pixel whose bilinear footprint touches it,
// recomputes that pixel's interpolated logits / softmax in registers, and gathers its own share of the gradient
// (no atomics on floats, bitwise reproducible).  The owner (y0==y && x0==x) also accounts loss / accuracy.
#define CE_CMAX 32
// WPP = waves per low-res pixel: 1 (small footprints, 4 pixels per block) or 4 (large footprints: the block's four waves
// split one pixel's candidates and meet in LDS)
template <int WPP>
__global__ void __launch_bounds__(256) k_upsample_ce(const float* __restrict__ lg, const int64_t* __restrict__ label, int B, int h,
                                                      int w, int C, int H, int W, int ignore, float sy, float sx,
                                                      float inv_total, float* __restrict__ loss_parts,
                                                      int32_t* __restrict__ counts, float* __restrict__ dlogits) {
  __shared__ float red[4][CE_CMAX + 4];
  constexpr int PS = CE_CMAX + 1;                // odd stride: the four corner vectors of a lane sit in different banks
  __shared__ float nb[WPP == 1 ? 4 : 1][9 * PS];  // the 3x3 low-res logit vectors a pixel's members interpolate from
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long pix_raw = WPP == 1 ? (long)blockIdx.x * 4 + wv : (long)blockIdx.x;
  const bool active = pix_raw < (long)B * h * w;  // tail waves of the WPP==1 form compute on the last pixel, write nothing
  const long pix = active ? pix_raw : (long)B * h * w - 1;
  const int x = (int)(pix % w);
  const long t = pix / w;
  const int y = (int)(t % h);
  const long b = t / h;
  // conservative candidate window in the high-res grid
  int hy0 = (int)floorf((y - 1 + 0.5f) / sy - 0.5f) - 1, hy1 = (int)ceilf((y + 1 + 0.5f) / sy - 0.5f) + 1;
  int hx0 = (int)floorf((x - 1 + 0.5f) / sx - 0.5f) - 1, hx1 = (int)ceilf((x + 1 + 0.5f) / sx - 0.5f) + 1;
  if (y == 0) hy0 = 0;  // clamped sources
  if (x == 0) hx0 = 0;
  hy0 = max(hy0, 0), hx0 = max(hx0, 0), hy1 = min(hy1, H - 1), hx1 = min(hx1, W - 1);
  const int nx = hx1 - hx0 + 1, ny = hy1 - hy0 + 1;
  float acc[CE_CMAX];
#pragma unroll
  for (int c = 0; c < CE_CMAX; ++c) acc[c] = 0.f;
  float loss = 0.f;
  int hits = 0, valid = 0;
  const float* lb = lg + b * (long)h * w * C;
  float* Ls = nb[WPP == 1 ? wv : 0];
  for (int i = (WPP == 1 ? lane : (int)threadIdx.x); i < 9 * C; i += 64 * WPP) {
    const int c = i % C, p = i / C;
    const int yy = min(max(y - 1 + p / 3, 0), h - 1), xx = min(max(x - 1 + p % 3, 0), w - 1);
    Ls[p * PS + c] = lb[((long)yy * w + xx) * C + c];
  }
  __syncthreads();
  for (int i = (WPP == 1 ? lane : (int)threadIdx.x); i < nx * ny; i += 64 * WPP) {
    const int hy = hy0 + i / nx, hx = hx0 + i % nx;
    const Lerp ly = lerp_idx(hy, sy, h), lx = lerp_idx(hx, sx, w);
    const float wy = (ly.i0 == y ? ly.l0 : 0.f) + (ly.i1 == y ? ly.l1 : 0.f);
    const float wx = (lx.i0 == x ? lx.l0 : 0.f) + (lx.i1 == x ? lx.l1 : 0.f);
    const bool member = (ly.i0 == y || ly.i1 == y) && (lx.i0 == x || lx.i1 == x);
    if (!member) continue;
    const int64_t lab = label[(b * H + hy) * (long)W + hx];
    const bool is_valid = lab != ignore;
    const bool owner = (ly.i0 == y) && (lx.i0 == x);
    if (!is_valid && !owner) continue;
    // members have i0 in {y-1, y} and i1 in {y, y+1}
    const float* p00 = Ls + ((ly.i0 - y + 1) * 3 + (lx.i0 - x + 1)) * PS;
    const float* p01 = Ls + ((ly.i0 - y + 1) * 3 + (lx.i1 - x + 1)) * PS;
    const float* p10 = Ls + ((ly.i1 - y + 1) * 3 + (lx.i0 - x + 1)) * PS;
    const float* p11 = Ls + ((ly.i1 - y + 1) * 3 + (lx.i1 - x + 1)) * PS;
    float v[CE_CMAX];
    float m = -INFINITY;
    int am = 0;
#pragma unroll
    for (int c = 0; c < CE_CMAX; ++c) {
      if (c < C) {
        v[c] = ly.l0 * (lx.l0 * p00[c] + lx.l1 * p01[c]) + ly.l1 * (lx.l0 * p10[c] + lx.l1 * p11[c]);
        if (v[c] > m) {
          m = v[c];
          am = c;
        }
      } else {
        v[c] = -INFINITY;
      }
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < CE_CMAX; ++c)
      if (c < C) se += __expf(v[c] - m);
    const float lse = m + __logf(se);
    if (is_valid) {
      const float g = wy * wx * inv_total;
      const float inv_se = 1.f / se;
#pragma unroll
      for (int c = 0; c < CE_CMAX; ++c)
        if (c < C) acc[c] += g * (__expf(v[c] - m) * inv_se - (c == (int)lab ? 1.f : 0.f));
      if (owner) {
        float vl = 0.f;
#pragma unroll
        for (int c = 0; c < CE_CMAX; ++c)
          if (c == (int)lab) vl = v[c];
        loss += lse - vl;
        valid += 1;
        hits += (am == (int)lab) ? 1 : 0;
      }
    }
  }
  loss = wave_sum(loss);
  for (int o = 32; o > 0; o >>= 1) {
    hits += __shfl_xor(hits, o, 64);
    valid += __shfl_xor(valid, o, 64);
  }
  if constexpr (WPP == 1) {
    if (dlogits) {
#pragma unroll
      for (int c = 0; c < CE_CMAX; ++c) {
        if (c < C) {
          const float s = wave_sum(acc[c]);
          if (lane == 0 && active) dlogits[pix * C + c] = s;
        }
      }
    }
    if (lane == 0 && active) {
      loss_parts[pix] = loss;
      if (valid) {
        atomicAdd(&counts[0], hits);
        atomicAdd(&counts[1], valid);
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < CE_CMAX; ++c) {
      if (c < C) {
        const float s = wave_sum(acc[c]);
        if (lane == 0) red[wv][c] = s;
      }
    }
    if (lane == 0) {
      red[wv][CE_CMAX] = loss;
      red[wv][CE_CMAX + 1] = __int_as_float(hits);
      red[wv][CE_CMAX + 2] = __int_as_float(valid);
    }
    __syncthreads();
    if (threadIdx.x < C && dlogits)
      dlogits[pix * C + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (threadIdx.x == 0) {
      loss_parts[pix] = (red[0][CE_CMAX] + red[1][CE_CMAX]) + (red[2][CE_CMAX] + red[3][CE_CMAX]);
      int hsum = 0, vsum = 0;
      for (int k = 0; k < 4; ++k) {
        hsum += __float_as_int(red[k][CE_CMAX + 1]);
        vsum += __float_as_int(red[k][CE_CMAX + 2]);
      }
      if (vsum) {
        atomicAdd(&counts[0], hsum);
        atomicAdd(&counts[1], vsum);
      }
    }
  }
}
// Tiled form for integer scale factors S and a compile-time class count (H = S*h, W = S*w; the x4 LinearHead and x16 VFMHead
// losses with 19 classes): a block owns T x T low-res pixels, i.e. a (T+1)S x (T+1)S high-res region, walked in strips of RS rows.
//   1. the (T+2)^2 low-res logit vectors it touches and the 2S bilinear weights of each owned row / column go to LDS; every
//      thread fetches the labels of all its high-res pixels up front (one exposed global latency per block),
//   2. per strip, every high-res pixel gets its interpolated logits / softmax ONCE per block (operands from LDS) and leaves
//      d = softmax - onehot in LDS (0 outside the image / ignored); the thread also accounts loss / accuracy of the pixels whose
//      owner (i0y, i0x) lies in the tile,
//   3. the gather is separable: a horizontal pass per strip (RS x T x C sums of 2S terms) and one vertical pass (T x T x C sums
//      of 2S terms), each in a fixed order (no float atomics, bitwise reproducible).
// LDS per block is 25-40 KB and the kernel stays under 128 VGPRs, so four or more blocks share a CU; the redundancy is
// ((T+1)/T)^2 softmax evaluations per high-res pixel, which is cheap next to the exposed latencies of an under-occupied CU.
// The block's loss goes to loss_parts[first owned pixel]; its other owned pixels get 0.
template <int S, int T, int C, int RS>
__global__ void __launch_bounds__(256) k_upsample_ce_tile(const float* __restrict__ lg, const int64_t* __restrict__ label, int B, int h,
                                                           int w, int H, int W, int ignore, float inv_total,
                                                           float* __restrict__ loss_parts, int32_t* __restrict__ counts,
                                                           float* __restrict__ dlogits) {
  constexpr int R = (T + 1) * S, LT = T + 2, PS = C | 1;  // odd LDS stride: conflict-free per-pixel vectors
  constexpr int NSTRIP = R / RS, PPT = (RS * R + 255) / 256;
  static_assert(R % RS == 0, "strips tile the region");
  __shared__ float Ls[LT * LT * PS];  // low-res logits
  __shared__ float Ds[RS * R * PS];   // softmax - onehot per high-res pixel of the strip
  __shared__ float Hs[R * T * PS];    // horizontal pass
  __shared__ float Wy[T][2 * S], Wx[T][2 * S];
  __shared__ float red_l[4];
  __shared__ int red_h[4], red_v[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tiles_x = w / T, tiles_y = h / T;
  const int bx = blockIdx.x % tiles_x, by = (blockIdx.x / tiles_x) % tiles_y;
  const long b = blockIdx.x / (tiles_x * tiles_y);
  const int y0 = by * T, x0 = bx * T;
  const float sc = 1.0f / S;
  const int hy_base = S * y0 - S / 2, hx_base = S * x0 - S / 2;
  const float* lb = lg + b * (long)h * w * C;
  int labs[NSTRIP * PPT];  // -1: outside the image / ignored / no pixel
#pragma unroll
  for (int k = 0; k < NSTRIP * PPT; ++k) {
    const int i = tid + (k % PPT) * 256;
    const int hy = hy_base + (k / PPT) * RS + i / R, hx = hx_base + i % R;
    int lab = -1;
    if (i < RS * R && hy >= 0 && hy < H && hx >= 0 && hx < W) {
      const int64_t l64 = label[(b * H + hy) * (long)W + hx];
      if (l64 != ignore) lab = (int)l64;
    }
    labs[k] = lab;
  }
  for (int i = tid; i < LT * LT * C; i += 256) {
    const int c = i % C, p = i / C;
    const int yy = min(max(y0 - 1 + p / LT, 0), h - 1), xx = min(max(x0 - 1 + p % LT, 0), w - 1);
    Ls[p * PS + c] = lb[((long)yy * w + xx) * C + c];
  }
  for (int i = tid; i < 2 * T * 2 * S; i += 256) {  // the share of owned row / column t in member d (0 outside the image)
    const int d = i % (2 * S), t = (i / (2 * S)) % T, isx = i / (2 * S * T);
    const int hp = (isx ? hx_base : hy_base) + t * S + d, own = (isx ? x0 : y0) + t, lim = isx ? W : H;
    float wgt = 0.f;
    if (hp >= 0 && hp < lim) {
      const Lerp l = lerp_idx(hp, sc, isx ? w : h);
      wgt = (l.i0 == own ? l.l0 : 0.f) + (l.i1 == own ? l.l1 : 0.f);
    }
    if (isx) Wx[t][d] = wgt * inv_total;
    else Wy[t][d] = wgt;
  }
  float loss = 0.f;
  int hits = 0, valid = 0;
#pragma unroll
  for (int st = 0; st < NSTRIP; ++st) {
    __syncthreads();  // Ls / weights ready (st == 0); previous strip's horizontal pass done with Ds
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int i = tid + j * 256;
      if (i >= RS * R) break;
      const int lab = labs[st * PPT + j];
      float v[C];
      if (lab >= 0) {
        const int hy = hy_base + st * RS + i / R, hx = hx_base + i % R;
        const Lerp ly = lerp_idx(hy, sc, h), lx = lerp_idx(hx, sc, w);
        const float* p00 = Ls + ((ly.i0 - y0 + 1) * LT + (lx.i0 - x0 + 1)) * PS;
        const float* p01 = Ls + ((ly.i0 - y0 + 1) * LT + (lx.i1 - x0 + 1)) * PS;
        const float* p10 = Ls + ((ly.i1 - y0 + 1) * LT + (lx.i0 - x0 + 1)) * PS;
        const float* p11 = Ls + ((ly.i1 - y0 + 1) * LT + (lx.i1 - x0 + 1)) * PS;
        float m = -INFINITY, vl = 0.f;
        int am = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          v[c] = ly.l0 * (lx.l0 * p00[c] + lx.l1 * p01[c]) + ly.l1 * (lx.l0 * p10[c] + lx.l1 * p11[c]);
          if (v[c] > m) m = v[c], am = c;
          if (c == lab) vl = v[c];
        }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = __expf(v[c] - m), se += v[c];
        const float inv_se = 1.f / se;
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = v[c] * inv_se - (c == lab ? 1.f : 0.f);
        if (ly.i0 >= y0 && ly.i0 < y0 + T && lx.i0 >= x0 && lx.i0 < x0 + T) {
          loss += m + __logf(se) - vl;
          valid += 1;
          hits += am == lab ? 1 : 0;
        }
      }
#pragma unroll
      for (int c = 0; c < C; ++c) Ds[i * PS + c] = lab >= 0 ? v[c] : 0.f;
    }
    __syncthreads();
    for (int task = tid; task < RS * T * C; task += 256) {
      const int c = task % C, tx = (task / C) % T, ry = task / (C * T);
      const float* dp = Ds + (ry * R + tx * S) * PS + c;
      float acc = 0.f;
#pragma unroll
      for (int dx = 0; dx < 2 * S; ++dx) acc += Wx[tx][dx] * dp[dx * PS];
      Hs[((st * RS + ry) * T + tx) * PS + c] = acc;
    }
  }
  loss = wave_sum(loss);
  for (int o = 32; o > 0; o >>= 1) {
    hits += __shfl_xor(hits, o, 64);
    valid += __shfl_xor(valid, o, 64);
  }
  if (lane == 0) red_l[wv] = loss, red_h[wv] = hits, red_v[wv] = valid;
  __syncthreads();
  for (int task = tid; task < T * T * C; task += 256) {
    const int c = task % C, tx = (task / C) % T, ty = task / (C * T);
    const float* hp = Hs + (ty * S * T + tx) * PS + c;
    float g = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2 * S; ++dy) g += Wy[ty][dy] * hp[dy * T * PS];
    if (dlogits) dlogits[((b * h + y0 + ty) * (long)w + x0 + tx) * C + c] = g;
  }
  if (tid < T * T) loss_parts[(b * h + y0 + tid / T) * (long)w + x0 + tid % T] = tid ? 0.f : (red_l[0] + red_l[1]) + (red_l[2] + red_l[3]);
  if (tid < 2) {  // one global tally per block: same-address global atomics serialise
    const int n = tid ? (red_v[0] + red_v[1]) + (red_v[2] + red_v[3]) : (red_h[0] + red_h[1]) + (red_h[2] + red_h[3]);
    if (n) atomicAdd(&counts[tid], n);
  }
}

extern "C" int vfm_upsample_ce(const float* logits_low, const int64_t* label, int B, int h, int w, int C, int H, int W,
                               int ignore_index, float* loss_parts, int32_t* counts, float* dlogits, void* stream) {
  VFM_CHECK(C > 0 && C <= CE_CMAX, VFM_E_SHAPE, "vfm_upsample_ce: C=%d > %d", C, CE_CMAX);
  VFM_CHECK(H >= h && W >= w, VFM_E_SHAPE, "vfm_upsample_ce: only up-sampling is supported");
  const long npix = (long)B * h * w;
  if (npix == 0) return VFM_OK;
  const bool big = ((long)H * W) >= 64L * h * w;  // footprint (2*scale)^2 >= 256 candidates per low-res pixel
  if (H == 4 * h && W == 4 * w && h % 4 == 0 && w % 4 == 0 && C == 19) {
    hipLaunchKernelGGL((k_upsample_ce_tile<4, 4, 19, 20>), dim3((unsigned)(B * (h / 4) * (w / 4))), dim3(256), 0, (hipStream_t)stream,
                       logits_low, label, B, h, w, H, W, ignore_index, 1.0f / ((float)B * H * W), loss_parts, counts, dlogits);
  } else if (H == 16 * h && W == 16 * w && C == 19) {
    hipLaunchKernelGGL((k_upsample_ce_tile<16, 1, 19, 8>), dim3((unsigned)npix), dim3(256), 0, (hipStream_t)stream, logits_low, label, B,
                       h, w, H, W, ignore_index, 1.0f / ((float)B * H * W), loss_parts, counts, dlogits);
  } else if (big)
    hipLaunchKernelGGL(k_upsample_ce<4>, dim3((unsigned)npix), dim3(256), 0, (hipStream_t)stream, logits_low, label, B, h, w, C, H, W,
                       ignore_index, (float)h / (float)H, (float)w / (float)W, 1.0f / ((float)B * H * W), loss_parts, counts,
                       dlogits);
  else
    hipLaunchKernelGGL(k_upsample_ce<1>, dim3(cdiv(npix, 4)), dim3(256), 0, (hipStream_t)stream, logits_low, label, B, h, w, C, H, W,
                       ignore_index, (float)h / (float)H, (float)w / (float)W, 1.0f / ((float)B * H * W), loss_parts, counts,
                       dlogits);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ---------------------------------------------------------------------------------------------------- input preprocessing
// mmseg SegDataPreProcessor on a decoded uint8 CHW image: optional BGR->RGB, (x - mean[c]) / std[c], right/bottom padding to
// [Hp, Wp] with pad_val (applied AFTER normalisation, as stack_batch does).  One sample of the batch per call.
__global__ void k_preprocess_u8(const uint8_t* __restrict__ in, int H, int W, float* __restrict__ out, int Hp, int Wp, float m0,
                                float m1, float m2, float s0, float s1, float s2, int swap_rb, float pad_val) {
  const long total = 3L * Hp * Wp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const long t = i / Wp;
    const int y = (int)(t % Hp), c = (int)(t / Hp);
    float v = pad_val;
    if (y < H && x < W) {
      const int cs = swap_rb ? 2 - c : c;
      const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
      v = ((float)in[((long)cs * H + y) * W + x] - mean) / sd;
    }
    out[i] = v;
  }
}
extern "C" int vfm_preprocess_u8(const uint8_t* img, int H, int W, float* out, int Hp, int Wp, const float* mean3, const float* std3,
                                 int bgr_to_rgb, float pad_val, void* stream) {
  VFM_CHECK(img && out && mean3 && std3 && Hp >= H && Wp >= W, VFM_E_SHAPE, "vfm_preprocess_u8: shape (mean3/std3 are HOST arrays)");
  const long total = 3L * Hp * Wp;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_preprocess_u8, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, H, W, out, Hp, Wp, mean3[0], mean3[1], mean3[2],
                     std3[0], std3[1], std3[2], bgr_to_rgb, pad_val);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ---------------------------------------------------------------------------------------------------- inference helpers
__global__ void k_conf_gate(const float* __restrict__ lg, int B, int C, int H, int W, int y0, int x0, int hc, int wc, float thr,
                            int32_t* __restrict__ count) {
  const long total = (long)B * hc * wc;
  int local = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % wc);
    long t = i / wc;
    const int y = (int)(t % hc);
    const long b = t / hc;
    const float* p = lg + (b * C) * (long)H * W + (long)(y + y0) * W + (x + x0);
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, p[(long)c * H * W]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(p[(long)c * H * W] - m);
    local += (1.0f / se > thr) ? 1 : 0;  // max softmax = exp(0)/se
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}
extern "C" int vfm_conf_gate(const float* logits, int B, int C, int H, int W, int y0, int x0, int hc, int wc, float thr,
                             int32_t* count, void* stream) {
  VFM_CHECK(y0 >= 0 && x0 >= 0 && y0 + hc <= H && x0 + wc <= W, VFM_E_SHAPE, "vfm_conf_gate: window");
  const long total = (long)B * hc * wc;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(k_conf_gate, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, B, C, H, W, y0, x0, hc, wc, thr, count);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

__global__ void k_slide_acc(const float* __restrict__ crop, int crop_nchw, int B, int h, int w, int C, float* __restrict__ preds,
                            float* __restrict__ count, int H, int W, int y0, int x0, int hc, int wc, float sy, float sx) {
  const long total = (long)B * C * hc * wc;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % wc);
    long t = i / wc;
    const int y = (int)(t % hc);
    t /= hc;
    const int c = (int)(t % C);
    const long b = t / C;
    const Lerp ly = lerp_idx(y, sy, h), lx = lerp_idx(x, sx, w);
    float v00, v01, v10, v11;
    if (crop_nchw) {
      const float* p = crop + (b * C + c) * (long)h * w;
      v00 = p[ly.i0 * w + lx.i0], v01 = p[ly.i0 * w + lx.i1], v10 = p[ly.i1 * w + lx.i0], v11 = p[ly.i1 * w + lx.i1];
    } else {
      const float* p = crop + b * (long)h * w * C + c;
      v00 = p[((long)ly.i0 * w + lx.i0) * C], v01 = p[((long)ly.i0 * w + lx.i1) * C];
      v10 = p[((long)ly.i1 * w + lx.i0) * C], v11 = p[((long)ly.i1 * w + lx.i1) * C];
    }
    const float v = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
    preds[((b * C + c) * (long)H + (y + y0)) * W + (x + x0)] += v;
    if (c == 0) count[(b * (long)H + (y + y0)) * W + (x + x0)] += 1.f;
  }
}
extern "C" int vfm_slide_accumulate(const float* crop, int crop_nchw, int B, int h, int w, int C, float* preds, float* count,
                                    int H, int W, int y0, int x0, int hc, int wc, void* stream) {
  VFM_CHECK(y0 >= 0 && x0 >= 0 && y0 + hc <= H && x0 + wc <= W, VFM_E_SHAPE, "vfm_slide_accumulate: window");
  const long total = (long)B * C * hc * wc;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipLaunchKernelGGL(k_slide_acc, dim3(grid), dim3(256), 0, (hipStream_t)stream, crop, crop_nchw, B, h, w, C, preds, count, H, W,
                     y0, x0, hc, wc, (float)h / (float)hc, (float)w / (float)wc);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ---- the sliding-window merge in ONE pass (gather form): every output pixel sums, in window order, the bilinear samples of the windows
// that cover it and divides by their number - what vfm_slide_accumulate (one read-modify-write pass over preds per window) followed by
// vfm_slide_finalize computes, with preds written once and never read: for a 1024^2 image and nine 512^2 windows 80 MB of stores
// instead of nine RMW passes of 40 MB, two fills and a 160-MB finalize.  Same per-sample arithmetic, same summation order.
struct SlideTab {
  vfm_slide_win w[16];
  int n;
};
template <int CMAX>
__global__ void __launch_bounds__(256) k_slide_gather(SlideTab tab, int B, int C, float* __restrict__ preds, int H, int W) {
  const long total = (long)B * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const long t = i / W;
    const int y = (int)(t % H);
    const long b = t / H;
    float acc[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) acc[c] = 0.f;
    float cnt = 0.f;
    for (int j = 0; j < tab.n; ++j) {
      const vfm_slide_win wd = tab.w[j];
      const int yy = y - wd.y0, xx = x - wd.x0;
      if (yy < 0 || yy >= wd.hc || xx < 0 || xx >= wd.wc) continue;
      cnt += 1.f;
      const Lerp ly = lerp_idx(yy, (float)wd.h / (float)wd.hc, wd.h), lx = lerp_idx(xx, (float)wd.w / (float)wd.wc, wd.w);
      if (wd.nchw) {
        const float* p = wd.crop + (b * C) * (long)wd.h * wd.w;
        const long plane = (long)wd.h * wd.w;
        const long o00 = ly.i0 * wd.w + lx.i0, o01 = ly.i0 * wd.w + lx.i1, o10 = ly.i1 * wd.w + lx.i0, o11 = ly.i1 * wd.w + lx.i1;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) {
            const float* q = p + c * plane;
            acc[c] += ly.l0 * (lx.l0 * q[o00] + lx.l1 * q[o01]) + ly.l1 * (lx.l0 * q[o10] + lx.l1 * q[o11]);
          }
      } else {
        const float* p = wd.crop + b * (long)wd.h * wd.w * C;
        const float* q00 = p + ((long)ly.i0 * wd.w + lx.i0) * C;
        const float* q01 = p + ((long)ly.i0 * wd.w + lx.i1) * C;
        const float* q10 = p + ((long)ly.i1 * wd.w + lx.i0) * C;
        const float* q11 = p + ((long)ly.i1 * wd.w + lx.i1) * C;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) acc[c] += ly.l0 * (lx.l0 * q00[c] + lx.l1 * q01[c]) + ly.l1 * (lx.l0 * q10[c] + lx.l1 * q11[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) preds[((b * C + c) * (long)H + y) * W + x] = acc[c] / cnt;
  }
}
extern "C" int vfm_slide_gather(const vfm_slide_win* wins, int nwin, int B, int C, float* preds, int H, int W, void* stream) {
  VFM_CHECK(wins && nwin >= 1 && nwin <= 16 && C >= 1 && C <= 32, VFM_E_UNSUPPORTED, "vfm_slide_gather: 1..16 windows, 1..32 channels");
  SlideTab tab;
  tab.n = nwin;
  for (int j = 0; j < nwin; ++j) {
    tab.w[j] = wins[j];
    VFM_CHECK(wins[j].crop && wins[j].y0 >= 0 && wins[j].x0 >= 0 && wins[j].y0 + wins[j].hc <= H && wins[j].x0 + wins[j].wc <= W &&
                  wins[j].h >= 1 && wins[j].w >= 1, VFM_E_SHAPE, "vfm_slide_gather: window %d", j);
  }
  const long total = (long)B * H * W;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipLaunchKernelGGL(k_slide_gather<32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, tab, B, C, preds, H, W);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// the confidence gates of ALL windows in one pass: max softmax > thr is evaluated once per pixel (the windows overlap: nine 512^2 windows of
// a 1024^2 map cover every pixel 2.25 times on average) and counted for every window that holds the pixel.  Integer atomics: exact.
struct GateTab {
  int y0[16], x0[16], hc[16], wc[16];
  int n;
};
__global__ void __launch_bounds__(256) k_conf_gate_windows(const float* __restrict__ lg, int B, int C, int H, int W, GateTab tab, float thr,
                                                          int32_t* __restrict__ counts) {
  const long total = (long)B * H * W;
  int local[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) local[j] = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const long t = i / W;
    const int y = (int)(t % H);
    const long b = t / H;
    const float* p = lg + (b * C) * (long)H * W + (long)y * W + x;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, p[(long)c * H * W]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(p[(long)c * H * W] - m);
    const int hit = (1.0f / se > thr) ? 1 : 0;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < tab.n && y >= tab.y0[j] && y < tab.y0[j] + tab.hc[j] && x >= tab.x0[j] && x < tab.x0[j] + tab.wc[j]) local[j] += hit;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j >= tab.n) break;
    int v = local[j];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(counts + j, v);
  }
}
extern "C" int vfm_conf_gate_windows(const float* logits, int B, int C, int H, int W, const int* boxes, int nwin, float thr, int32_t* counts,
                                     void* stream) {
  VFM_CHECK(logits && boxes && counts && nwin >= 1 && nwin <= 16, VFM_E_UNSUPPORTED, "vfm_conf_gate_windows: 1..16 windows");
  GateTab tab;
  tab.n = nwin;
  for (int j = 0; j < nwin; ++j) {
    tab.y0[j] = boxes[4 * j], tab.x0[j] = boxes[4 * j + 1], tab.hc[j] = boxes[4 * j + 2], tab.wc[j] = boxes[4 * j + 3];
    VFM_CHECK(tab.y0[j] >= 0 && tab.x0[j] >= 0 && tab.y0[j] + tab.hc[j] <= H && tab.x0[j] + tab.wc[j] <= W, VFM_E_SHAPE, "vfm_conf_gate_windows: window %d", j);
  }
  for (int j = nwin; j < 16; ++j) tab.y0[j] = tab.x0[j] = tab.hc[j] = tab.wc[j] = 0;
  const long total = (long)B * H * W;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(k_conf_gate_windows, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, B, C, H, W, tab, thr, counts);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

__global__ void k_slide_fin(float* __restrict__ preds, const float* __restrict__ count, uint8_t* __restrict__ am, int B, int C,
                            int H, int W) {
  const long total = (long)B * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / ((long)H * W), p = i - b * (long)H * W;
    const float cnt = count ? count[i] : 1.f;
    float best = -INFINITY;
    int bi = 0;
    for (int c = 0; c < C; ++c) {
      float* q = preds + (b * C + c) * (long)H * W + p;
      const float v = count ? *q / cnt : *q;
      if (count) *q = v;
      if (v > best) {
        best = v;
        bi = c;
      }
    }
    if (am) am[i] = (uint8_t)bi;
  }
}
extern "C" int vfm_slide_finalize(float* preds, const float* count, uint8_t* argmax, int B, int C, int H, int W, void* stream) {
  const long total = (long)B * H * W;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_slide_fin, dim3(grid), dim3(256), 0, (hipStream_t)stream, preds, count, argmax, B, C, H, W);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ---------------------------------------------------------------------------------------------------- AdamW
// float4 per lane (every segment starts on a 16-float boundary of the flat buffers when seg_start % 4 == 0: a vector never
// straddles two parameter groups), the segment table searched in LDS once per vector; 28 B/param of traffic (+4 B when the
// gradient is cleared in the same pass: zero_grad fused, no separate 66 MB fill launch).
template <bool VEC>
__global__ void __launch_bounds__(256) k_adamw(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                               const long* __restrict__ seg_start, const float* __restrict__ seg_lr,
                                               const float* __restrict__ seg_wd, int n_seg, float lr, float b1, float b2, float eps, float bc1,
                                               float bc2_sqrt, float gscale, int zero_grad, const int* __restrict__ skip,
                                               const float* __restrict__ amp_state) {
  extern __shared__ long s_start[];
  // amp_state (device, AmpOptimWrapper): [0] loss scale, [2] optimiser steps taken so far.  With it the un-scale factor and the bias
  // corrections are formed here, so neither the scale nor the step count has to be known on the host when the launch is enqueued.
  if (amp_state != nullptr) {
    gscale = gscale / amp_state[0];
    const float st = amp_state[2] + 1.f;
    bc1 = 1.f - powf(b1, st);
    bc2_sqrt = sqrtf(1.f - powf(b2, st));
  }
  // loss-scaled training (AmpOptimWrapper): a step whose gradients hold an inf / NaN leaves parameters and moments alone.  The flag is
  // read on the device, so the host does not have to wait for the backward pass before it can enqueue this launch.
  const bool skipped = skip != nullptr && *skip != 0;
  if (skipped && !zero_grad) return;
  for (int i = threadIdx.x; i < n_seg; i += 256) s_start[i] = seg_start[i];
  __syncthreads();
  constexpr int W = VEC ? 4 : 1;
  const long nv = VEC ? n / 4 : n;
  for (long iv = blockIdx.x * 256L + threadIdx.x; iv < nv; iv += (long)gridDim.x * 256L) {
    const long i = iv * W;
    int lo = 0, hi = n_seg - 1;  // largest s with seg_start[s] <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_start[mid] <= i) lo = mid;
      else hi = mid - 1;
    }
    if (skipped) {   // only the gradient clear of the fused zero_grad
      if constexpr (VEC) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
      else g[i] = 0.f;
      continue;
    }
    const float lr_i = lr * seg_lr[lo], decay = 1.f - lr_i * seg_wd[lo], step = lr_i / bc1;
    float gg[W], pp[W], mm[W], vv[W];
    if constexpr (VEC) {
      const float4 g4 = *reinterpret_cast<const float4*>(g + i), p4 = *reinterpret_cast<const float4*>(p + i);
      const float4 m4 = *reinterpret_cast<const float4*>(m + i), v4 = *reinterpret_cast<const float4*>(v + i);
      gg[0] = g4.x, gg[1] = g4.y, gg[2] = g4.z, gg[3] = g4.w, pp[0] = p4.x, pp[1] = p4.y, pp[2] = p4.z, pp[3] = p4.w;
      mm[0] = m4.x, mm[1] = m4.y, mm[2] = m4.z, mm[3] = m4.w, vv[0] = v4.x, vv[1] = v4.y, vv[2] = v4.z, vv[3] = v4.w;
    } else {
      gg[0] = g[i], pp[0] = p[i], mm[0] = m[i], vv[0] = v[i];
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const float gi = gg[k] * gscale;
      const float mi = b1 * mm[k] + (1.f - b1) * gi;
      const float vi = b2 * vv[k] + (1.f - b2) * gi * gi;
      const float denom = sqrtf(vi) / bc2_sqrt + eps;
      pp[k] = pp[k] * decay - step * (mi / denom);
      mm[k] = mi, vv[k] = vi;
    }
    if constexpr (VEC) {
      *reinterpret_cast<float4*>(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
      *reinterpret_cast<float4*>(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
      *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      if (zero_grad) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      p[i] = pp[0], m[i] = mm[0], v[i] = vv[0];
      if (zero_grad) g[i] = 0.f;
    }
  }
}
extern "C" int vfm_adamw(float* p, float* g, float* m, float* v, long n, const long* seg_start, const float* seg_lr_mult,
                         const float* seg_wd, int n_seg, float lr, float beta1, float beta2, float eps, int step,
                         float grad_scale, int zero_grad, int vec4, void* stream) {
  return vfm_adamw_guarded(p, g, m, v, n, seg_start, seg_lr_mult, seg_wd, n_seg, lr, beta1, beta2, eps, step, grad_scale, zero_grad, vec4, nullptr, nullptr,
                           stream);
}
// one thread: torch GradScaler.update() + the optimiser's step count, on the device
__global__ void k_amp_update(const int* __restrict__ flag, float* __restrict__ st, float growth, float backoff, int interval, int dynamic) {
  if (*flag != 0) {
    st[3] += 1.f;                       // skipped steps
    if (dynamic) st[0] *= backoff, st[1] = 0.f;
  } else {
    st[2] += 1.f;                       // optimiser steps taken
    if (dynamic) {
      st[1] += 1.f;
      if ((int)st[1] == interval) st[0] *= growth, st[1] = 0.f;
    }
  }
}
extern "C" int vfm_amp_update(const int* flag, float* amp_state, float growth, float backoff, int interval, int dynamic, void* stream) {
  VFM_CHECK(flag && amp_state && interval >= 1, VFM_E_INVAL, "vfm_amp_update: arguments");
  hipLaunchKernelGGL(k_amp_update, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, amp_state, growth, backoff, interval, dynamic);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
extern "C" int vfm_adamw_guarded(float* p, float* g, float* m, float* v, long n, const long* seg_start, const float* seg_lr_mult,
                                 const float* seg_wd, int n_seg, float lr, float beta1, float beta2, float eps, int step,
                                 float grad_scale, int zero_grad, int vec4, const int* skip, const float* amp_state, void* stream) {
  VFM_CHECK(n_seg >= 1 && n_seg <= 8192 && step >= 1, VFM_E_INVAL, "vfm_adamw: n_seg/step");
  if (n == 0) return VFM_OK;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  const bool vec = vec4 && n % 4 == 0 && a16(p) && a16(g) && a16(m) && a16(v);   // vec4: the caller vouches that seg_start % 4 == 0
  const long work = vec ? n / 4 : n;
  const int grid = (int)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
  const size_t lds = (size_t)n_seg * sizeof(long);
  if (vec)
    hipLaunchKernelGGL(k_adamw<true>, dim3(grid), dim3(256), lds, (hipStream_t)stream, p, g, m, v, n, seg_start, seg_lr_mult, seg_wd, n_seg,
                       lr, beta1, beta2, eps, bc1, bc2s, grad_scale, zero_grad, skip, amp_state);
  else
    hipLaunchKernelGGL(k_adamw<false>, dim3(grid), dim3(256), lds, (hipStream_t)stream, p, g, m, v, n, seg_start, seg_lr_mult, seg_wd, n_seg,
                       lr, beta1, beta2, eps, bc1, bc2s, grad_scale, zero_grad, skip, amp_state);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
