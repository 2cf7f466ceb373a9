// Elementwise / layout kernels (HBM-bound): casts, transposes, masks, GEGLU, mask-token, reductions.
#include "common.h"

thread_local char g_vfm_err[512] = {0};

extern "C" const char* vfm_last_error(void) { return g_vfm_err; }
extern "C" int vfm_abi_version(void) { return 3; }   // 2: launch plans (vfm_run_plan, vfm_prof_*); 3: vfm_gemm_desc.c_plane (c_dt VFM_SPLIT3)
extern "C" int vfm_half_kind(void) { return VFM_HALF_KIND; }

// ------------------------------------------------------------------------------------------------ cast
template <typename TI, typename TO>
__global__ void k_cast(const TI* __restrict__ src, long ld_src, TO* __restrict__ dst, long ld_dst, long rows, long cols,
                       const float* __restrict__ colscale) {
  const long cols4 = cols >> 2;
  const long total = rows * cols4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols4, c = (i - r * cols4) << 2;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ld_f32(src + r * ld_src + c + j);
    if (colscale) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= colscale[c + j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) st_f32(dst + r * ld_dst + c + j, v[j]);
  }
  // tail columns (cols % 4)
  const long tail0 = cols4 << 2;
  const long ntail = cols - tail0;
  if (ntail) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < rows * ntail; i += (long)gridDim.x * blockDim.x) {
      const long r = i / ntail, c = tail0 + (i - r * ntail);
      float v = ld_f32(src + r * ld_src + c);
      if (colscale) v *= colscale[c];
      st_f32(dst + r * ld_dst + c, v);
    }
  }
}

extern "C" int vfm_cast(const void* src, int src_dt, long ld_src, void* dst, int dst_dt, long ld_dst, long rows, long cols,
                        const float* colscale, void* stream) {
  VFM_CHECK(rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= cols, VFM_E_SHAPE, "vfm_cast: bad shape");
  if (rows == 0 || cols == 0) return VFM_OK;
  hipStream_t s = (hipStream_t)stream;
  const long total = rows * ((cols + 3) / 4);
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
#define L(TI, TO) hipLaunchKernelGGL((k_cast<TI, TO>), dim3(grid), dim3(256), 0, s, (const TI*)src, ld_src, (TO*)dst, ld_dst, rows, cols, colscale)
  if (src_dt == VFM_F32 && dst_dt == VFM_F32) L(float, float);
  else if (src_dt == VFM_F32 && dst_dt == VFM_BF16) L(float, bf16_t);
  else if (src_dt == VFM_BF16 && dst_dt == VFM_F32) L(bf16_t, float);
  else if (src_dt == VFM_BF16 && dst_dt == VFM_BF16) L(bf16_t, bf16_t);
  else VFM_FAIL(VFM_E_INVAL, "vfm_cast: dtype");
#undef L
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ split-bf16 operands
// An fp32 GEMM operand X[rows, K] (any strides) as three bf16 K-segments of Kp = ceil64(K) columns each, so that ONE bf16 MFMA GEMM over
// K' = 3 Kp computes  sum_k (a_hi b_hi + a_hi b_lo + a_lo b_hi)  with fp32 accumulation - the "bf16 x 3" form of an fp32 product
// (hi = bf16(x), lo = bf16(x - hi): 16 significant bits between them; the dropped a_lo b_lo term is 2^-18 relative):
//     pattern 0 (the A side): [hi | hi | lo]        pattern 1 (the B side): [hi | lo | hi]
// Columns K .. Kp-1 of every segment are zero.  Rounding is RNE on both halves (plain casts: v_cvt_pk_bf16_f32).
__global__ void k_split3(const float* __restrict__ src, long sr, long sc, bf16_t* __restrict__ dst, long ld_dst, long rows, long K, long Kp,
                         int pattern) {
  const long per_row = Kp / 4;
  const long total = rows * per_row;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / per_row, c = (i % per_row) * 4;
    float v[4];
    if (sc == 1 && c + 3 < K && ((reinterpret_cast<uintptr_t>(src + r * sr + c) & 15) == 0)) {
      const float4 f = *reinterpret_cast<const float4*>(src + r * sr + c);
      v[0] = f.x, v[1] = f.y, v[2] = f.z, v[3] = f.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (c + e < K) ? src[r * sr + (c + e) * sc] : 0.f;
    }
    ushort4 hi, lo;
    bf16_t* ph = &hi.x;
    bf16_t* pl = &lo.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ph[e] = f32_to_bf16(v[e]);
      pl[e] = f32_to_bf16(v[e] - bf16_to_f32(ph[e]));
    }
    bf16_t* d = dst + r * ld_dst + c;
    *reinterpret_cast<ushort4*>(d) = hi;
    *reinterpret_cast<ushort4*>(d + Kp) = pattern == 0 ? hi : lo;
    *reinterpret_cast<ushort4*>(d + 2 * Kp) = pattern == 0 ? lo : hi;
  }
}

extern "C" int vfm_split3(const float* src, long stride_r, long stride_c, void* dst, long ld_dst, long rows, long K, int pattern, void* stream) {
  const long Kp = (K + 63) / 64 * 64;
  VFM_CHECK(rows >= 0 && K > 0 && ld_dst >= 3 * Kp && ld_dst % 4 == 0 && ((uintptr_t)dst % 8) == 0 && (pattern == 0 || pattern == 1), VFM_E_SHAPE,
            "vfm_split3: dst must hold rows x 3 ceil64(K) bf16 (ld %% 4 == 0, 8-byte aligned)");
  if (rows == 0) return VFM_OK;
  const long total = rows * (Kp / 4);
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_split3, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, stride_r, stride_c, (bf16_t*)dst, ld_dst, rows, K, Kp, pattern);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ transpose
// 64x64 tiles through LDS (+1 pad): coalesced reads along cols, coalesced writes along rows.
template <typename TI, typename TO>
__global__ void k_transpose(const TI* __restrict__ src, long ld_src, TO* __restrict__ dst, long ld_dst, long rows, long cols,
                            long pad_rows) {
  __shared__ float tile[64][65];
  const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 4 row-groups
  for (int i = ty; i < 64; i += 4) {
    const long r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? ld_f32(src + r * ld_src + c) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const long c = c0 + i, r = r0 + tx;  // dst[c, r]
    if (c < cols && r < pad_rows) st_f32(dst + c * ld_dst + r, tile[tx][i]);
  }
}

extern "C" int vfm_transpose(const void* src, int src_dt, long ld_src, void* dst, int dst_dt, long ld_dst, long rows, long cols,
                             long pad_rows, void* stream) {
  VFM_CHECK(pad_rows >= rows && ld_dst >= pad_rows && ld_src >= cols, VFM_E_SHAPE, "vfm_transpose: bad shape");
  if (cols == 0 || pad_rows == 0) return VFM_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(cols, 64), cdiv(pad_rows, 64));
#define L(TI, TO) hipLaunchKernelGGL((k_transpose<TI, TO>), grid, dim3(256), 0, s, (const TI*)src, ld_src, (TO*)dst, ld_dst, rows, cols, pad_rows)
  if (src_dt == VFM_F32 && dst_dt == VFM_F32) L(float, float);
  else if (src_dt == VFM_F32 && dst_dt == VFM_BF16) L(float, bf16_t);
  else if (src_dt == VFM_BF16 && dst_dt == VFM_F32) L(bf16_t, float);
  else if (src_dt == VFM_BF16 && dst_dt == VFM_BF16) L(bf16_t, bf16_t);
  else VFM_FAIL(VFM_E_INVAL, "vfm_transpose: dtype");
#undef L
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ strided copy
template <typename TI, typename TO>
__global__ void k_strided_copy(const TI* __restrict__ src, TO* __restrict__ dst, long n0, long n1, long n2, long n3, long s0,
                               long s1, long s2, long s3, long d0, long d1, long d2, long d3, int accumulate) {
  const long total = n0 * n1 * n2 * n3;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long t = i;
    const long i3 = t % n3; t /= n3;
    const long i2 = t % n2; t /= n2;
    const long i1 = t % n1; t /= n1;
    const float v = ld_f32(src + t * s0 + i1 * s1 + i2 * s2 + i3 * s3);
    TO* q = dst + t * d0 + i1 * d1 + i2 * d2 + i3 * d3;
    st_f32(q, accumulate ? ld_f32(q) + v : v);
  }
}

extern "C" int vfm_strided_copy(const void* src, int src_dt, void* dst, int dst_dt, long n0, long n1, long n2, long n3,
                                long s0, long s1, long s2, long s3, long d0, long d1, long d2, long d3, int accumulate,
                                void* stream) {
  const long total = n0 * n1 * n2 * n3;
  if (total == 0) return VFM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
#define L(TI, TO) hipLaunchKernelGGL((k_strided_copy<TI, TO>), dim3(grid), dim3(256), 0, s, (const TI*)src, (TO*)dst, n0, n1, n2, n3, s0, s1, s2, s3, d0, d1, d2, d3, accumulate)
  if (src_dt == VFM_F32 && dst_dt == VFM_F32) L(float, float);
  else if (src_dt == VFM_F32 && dst_dt == VFM_BF16) L(float, bf16_t);
  else if (src_dt == VFM_BF16 && dst_dt == VFM_F32) L(bf16_t, float);
  else if (src_dt == VFM_BF16 && dst_dt == VFM_BF16) L(bf16_t, bf16_t);
  else VFM_FAIL(VFM_E_INVAL, "vfm_strided_copy: dtype");
#undef L
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ axpby / scale
__global__ void k_axpby(const float* __restrict__ x, float a, float* __restrict__ y, float b, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = (b == 0.f) ? a * x[i] : a * x[i] + b * y[i];
}
extern "C" int vfm_axpby(const float* x, float a, float* y, float b, long n, void* stream) {
  if (n == 0) return VFM_OK;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(k_axpby, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, a, y, b, n);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
__global__ void k_scale_dev(float* __restrict__ y, const float* __restrict__ sc, long n) {
  const float s = *sc;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] *= s;
}
extern "C" int vfm_scale_by_device_scalar(float* y, const float* scalar, long n, void* stream) {
  if (n == 0) return VFM_OK;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(k_scale_dev, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, scalar, n);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ column sums
// stage 1: block = CT columns x (256/CT) row lanes; grid (cdiv(cols,CT), parts): ws[part, c] = sum over the part's rows.
// CT adapts to narrow matrices (e.g. 19 logit columns) so that all 256 threads stay busy.
template <typename T>
__global__ void k_colsum1(const T* __restrict__ x, long ld, long rows, long cols, float* __restrict__ ws, int CT) {
  __shared__ float sh[256];
  const int rl = 256 / CT;
  const int tx = threadIdx.x % CT, ty = threadIdx.x / CT;
  const long c = (long)blockIdx.x * CT + tx;
  float acc = 0.f;
  if (c < cols)
    for (long r = (long)blockIdx.y * rl + ty; r < rows; r += (long)gridDim.y * rl) acc += ld_f32(x + r * ld + c);
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (ty == 0 && c < cols) {
    float a = 0.f;
    for (int k = 0; k < rl; ++k) a += sh[k * CT + tx];
    ws[(long)blockIdx.y * cols + c] = a;
  }
}
// second stage: 64 columns x 4 part groups per block; every thread's loads are independent (unrolled), the four group sums meet in
// LDS in a fixed order (deterministic).  The first form - one thread per column walking all parts - was a chain of up to 64
// dependent loads: 16 us per call, ~30 calls per train step.
__global__ void __launch_bounds__(256) k_colsum2(const float* __restrict__ ws, long cols, int parts, float* __restrict__ out, int accumulate) {
  __shared__ float sh[4][64];
  const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long c = blockIdx.x * 64L + tx;
  float a = 0.f;
  if (c < cols) {
    int p = g;
    for (; p + 12 < parts; p += 16) {
      const float v0 = ws[(long)p * cols + c], v1 = ws[(long)(p + 4) * cols + c], v2 = ws[(long)(p + 8) * cols + c], v3 = ws[(long)(p + 12) * cols + c];
      a += (v0 + v1) + (v2 + v3);
    }
    for (; p < parts; p += 4) a += ws[(long)p * cols + c];
  }
  sh[g][tx] = a;
  __syncthreads();
  if (g == 0 && c < cols) {
    const float t = (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
    out[c] = accumulate ? out[c] + t : t;
  }
}
extern "C" int vfm_colsum(const void* x, int dt, long ld, long rows, long cols, float* out, int accumulate, float* ws,
                          void* stream) {
  VFM_CHECK(ld >= cols && ws, VFM_E_SHAPE, "vfm_colsum: bad args");
  if (cols == 0) return VFM_OK;
  hipStream_t s = (hipStream_t)stream;
  int CT = 64;
  while (CT > 8 && CT / 2 >= cols) CT /= 2;
  const int rl = 256 / CT;
  long parts = (rows + rl * 8 - 1) / (rl * 8);  // >= 8 rows per thread
  if (parts > 64) parts = 64;
  if (parts < 1) parts = 1;
  dim3 grid(cdiv(cols, CT), (unsigned)parts);
  if (dt == VFM_F32) hipLaunchKernelGGL(k_colsum1<float>, grid, dim3(256), 0, s, (const float*)x, ld, rows, cols, ws, CT);
  else if (dt == VFM_BF16) hipLaunchKernelGGL(k_colsum1<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, ld, rows, cols, ws, CT);
  else VFM_FAIL(VFM_E_INVAL, "vfm_colsum: dtype");
  hipLaunchKernelGGL(k_colsum2, dim3(cdiv(cols, 64)), dim3(256), 0, s, ws, cols, (int)parts, out, accumulate);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ split-K slab reduction
// dst[p*sp + q*sq] (+)= alpha * sum_k slabs[k][p][q]  for p < rows_used: the combine step of the split-K weight-gradient
// GEMMs, with the (possibly transposing) scatter into the parameter's gradient layout and the accumulation fused in -
// one launch instead of colsum (2) + strided copy (1).  Fixed summation order: bitwise reproducible.
__global__ void k_slab_reduce(const float* __restrict__ slabs, int kch, long P, long Q, long rows_used, float alpha,
                              float* __restrict__ dst, long sp, long sq, int accumulate) {
  const long total = rows_used * Q;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i / Q, q = i - p * Q;
    float a = 0.f;
    for (int k = 0; k < kch; ++k) a += slabs[((long)k * P + p) * Q + q];
    a *= alpha;
    float* d = dst + p * sp + q * sq;
    *d = accumulate ? *d + a : a;
  }
}
extern "C" int vfm_slab_reduce(const float* slabs, int kch, long P, long Q, long rows_used, float alpha, float* dst, long sp,
                               long sq, int accumulate, void* stream) {
  VFM_CHECK(kch >= 1 && rows_used >= 0 && rows_used <= P, VFM_E_SHAPE, "vfm_slab_reduce: bad shape");
  const long total = rows_used * Q;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(k_slab_reduce, dim3(grid), dim3(256), 0, (hipStream_t)stream, slabs, kch, P, Q, rows_used, alpha, dst, sp, sq,
                     accumulate);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ batched strided copy
// One launch for a whole TABLE of 4-D strided fp32 -> (bf16 | fp32) copies (blockIdx.y = entry): the per-step re-pack of every
// trainable decoder weight into its GEMM operand layout (39 single launches + zero fills per train step before).
__global__ void __launch_bounds__(256) k_strided_copy_batch(const vfm_copy_job* __restrict__ tab) {
  const vfm_copy_job t = tab[blockIdx.y];
  const unsigned n1 = (unsigned)t.n[1], n2 = (unsigned)t.n[2], n3 = (unsigned)t.n[3];
  const unsigned total = (unsigned)t.n[0] * n1 * n2 * n3;   // < 2^31 elements per job (checked on the host side of the table)
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned r = i;
    const unsigned i3 = r % n3; r /= n3;
    const unsigned i2 = r % n2; r /= n2;
    const unsigned i1 = r % n1; r /= n1;
    const float* sp = t.src + ((long)r * t.s[0] + (long)i1 * t.s[1] + (long)i2 * t.s[2] + (long)i3 * t.s[3]);
    float v = sp[0];
    for (long c = 1; c < t.nsum; ++c) v += sp[c * t.sum_stride];   // partial results of one reduction, summed in a fixed order
    const long o = (long)r * t.d[0] + (long)i1 * t.d[1] + (long)i2 * t.d[2] + (long)i3 * t.d[3];
    if (t.accumulate) ((float*)t.dst)[o] += v;
    else st_any(t.dst, o, (int)t.dst_dt, v);
  }
}
extern "C" int vfm_strided_copy_batch(const vfm_copy_job* table_dev, int njobs, long max_elems, void* stream) {
  VFM_CHECK(table_dev && njobs >= 0, VFM_E_INVAL, "vfm_strided_copy_batch: bad args");
  if (njobs == 0 || max_elems <= 0) return VFM_OK;
  VFM_CHECK(max_elems < (1l << 31), VFM_E_SHAPE, "vfm_strided_copy_batch: jobs are limited to 2^31 elements");
  int gx = (int)((max_elems + 2047) / 2048);   // ~8 elements per thread of the largest job; smaller jobs' surplus blocks exit at once
  if (gx > 2048) gx = 2048;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(k_strided_copy_batch, dim3(gx, njobs), dim3(256), 0, (hipStream_t)stream, table_dev);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ LoRA factor re-pack
// After every optimiser step the fp32 LoRA factors are re-packed into the K-concatenated GEMM operands of their layer:
//   a[:r, :K] = A            at[:K, :r] = A^T          w[:N, Kw:Kw+r] = B          wt[Kw:Kw+r, :N] = B^T   (wt may be null)
// One launch for ALL adapter sites of the model (blockIdx.y = site), instead of four small kernels per site and step.
__global__ void k_lora_pack(const vfm_lora_site* __restrict__ tab, int dt) {
  const vfm_lora_site t = tab[blockIdx.y];
  const long na = t.r * t.K, nb = t.N * t.r;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < na + nb; i += (long)gridDim.x * blockDim.x) {
    if (i < na) {
      const long rr = i / t.K, k = i - rr * t.K;
      const float v = t.A[i];
      st_any(t.a, rr * t.ld_a + k, dt, v);
      st_any(t.at, k * t.ld_at + rr, dt, v);
    } else {
      const long j = i - na;
      const long n = j / t.r, rr = j - n * t.r;
      const float v = t.B[j];
      st_any(t.w, n * t.ld_w + t.Kw + rr, dt, v);
      if (t.wt) st_any(t.wt, (t.Kw + rr) * t.ld_wt + n, dt, v);
    }
  }
}
extern "C" int vfm_lora_pack(const vfm_lora_site* table_dev, int nsites, long max_elems, int dt, void* stream) {
  VFM_CHECK(table_dev && nsites >= 0 && (dt == VFM_F32 || dt == VFM_BF16), VFM_E_INVAL, "vfm_lora_pack: bad args");
  if (nsites == 0) return VFM_OK;
  int gx = (int)((max_elems + 255) / 256);
  if (gx > 256) gx = 256;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(k_lora_pack, dim3(gx, nsites), dim3(256), 0, (hipStream_t)stream, table_dev, dt);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ dropout
// ------------------------------------------------------------------------------------------------ bf16 x8 fast paths
// The dtype-erased element-per-thread kernels below move 2 bytes per lane and instruction and spend a 64-bit division per element:
// on [4100 x 4096] bf16 operands they run at 40-45 % of the HBM rate.  When every operand is bf16 (or the named fp32 ones), the
// row length and leading dimensions are multiples of 8 and the bases 16-byte aligned, one lane moves 8 elements (16 bytes).
struct F8 { float v[8]; };
__device__ __forceinline__ F8 ld8_bf16(const bf16_t* p) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  F8 r;
  r.v[0] = h16_lo(u.x), r.v[1] = h16_hi(u.x);
  r.v[2] = h16_lo(u.y), r.v[3] = h16_hi(u.y);
  r.v[4] = h16_lo(u.z), r.v[5] = h16_hi(u.z);
  r.v[6] = h16_lo(u.w), r.v[7] = h16_hi(u.w);
  return r;
}
__device__ __forceinline__ void st8_bf16(bf16_t* p, const F8& r) {
  uint4 u;
  u.x = (uint32_t)f32_to_bf16(r.v[0]) | ((uint32_t)f32_to_bf16(r.v[1]) << 16);
  u.y = (uint32_t)f32_to_bf16(r.v[2]) | ((uint32_t)f32_to_bf16(r.v[3]) << 16);
  u.z = (uint32_t)f32_to_bf16(r.v[4]) | ((uint32_t)f32_to_bf16(r.v[5]) << 16);
  u.w = (uint32_t)f32_to_bf16(r.v[6]) | ((uint32_t)f32_to_bf16(r.v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ F8 ld8_f32(const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  F8 r;
  r.v[0] = a.x, r.v[1] = a.y, r.v[2] = a.z, r.v[3] = a.w, r.v[4] = b.x, r.v[5] = b.y, r.v[6] = b.z, r.v[7] = b.w;
  return r;
}
__device__ __forceinline__ void st8_f32(float* p, const F8& r) {
  *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static inline int grid8(long total8) { return (int)((total8 + 255) / 256 > 16384 ? 16384 : (total8 + 255) / 256); }

__global__ void k_dropout_mask_bf16x8(bf16_t* __restrict__ out, long n8, float p, float keep_scale, uint64_t seed, uint64_t offset) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    F8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float u = (hash_u32(seed, offset + (uint64_t)(i * 8 + e)) >> 8) * (1.0f / 16777216.0f);
      r.v[e] = u >= p ? keep_scale : 0.f;
    }
    st8_bf16(out + i * 8, r);
  }
}
__global__ void k_mul_mask_bf16x8(const bf16_t* __restrict__ src, long ld_src, const bf16_t* __restrict__ mask, long mask_ld, long rpg,
                                  bf16_t* __restrict__ dst, long ld_dst, long rows, int cols8) {
  const long total = rows * cols8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols8;
    const int c = (int)(i - r * cols8) * 8;
    F8 a = ld8_bf16(src + r * ld_src + c);
    const F8 m = ld8_bf16(mask + (r / rpg) * mask_ld + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) a.v[e] *= m.v[e];
    st8_bf16(dst + r * ld_dst + c, a);
  }
}
__global__ void k_act_grad_mul_bf16x8(const bf16_t* __restrict__ dy, long ld_dy, const bf16_t* __restrict__ pre, long ld_pre,
                                      bf16_t* __restrict__ out, long ld_out, long rows, int cols8, int act) {
  const long total = rows * cols8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols8;
    const int c = (int)(i - r * cols8) * 8;
    F8 d = ld8_bf16(dy + r * ld_dy + c);
    const F8 p = ld8_bf16(pre + r * ld_pre + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = p.v[e];
      const float g = act == VFM_ACT_GELU ? gelu_grad_f(x) : (act == VFM_ACT_RELU ? (x > 0.f ? 1.f : 0.f) : (act == VFM_ACT_QGELU ? qgelu_grad_f(x) : 1.f));
      d.v[e] *= g;
    }
    st8_bf16(out + r * ld_out + c, d);
  }
}
// SwiGLU with bf16 h = [a | g] and fp32 or bf16 products (EVA02: the sub-LN input / its gradient are fp32)
template <bool OUT_BF16>
__global__ void k_swiglu_fwd_x8(const bf16_t* __restrict__ h, long ld_h, void* __restrict__ out, long ld_out, long rows, long C, int c8) {
  const long total = rows * c8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c8;
    const int c = (int)(i - r * c8) * 8;
    const F8 a = ld8_bf16(h + r * ld_h + c), g = ld8_bf16(h + r * ld_h + C + c);
    F8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = a.v[e] / (1.f + __expf(-a.v[e])) * g.v[e];
    if constexpr (OUT_BF16) st8_bf16((bf16_t*)out + r * ld_out + c, o);
    else st8_f32((float*)out + r * ld_out + c, o);
  }
}
template <bool DO_BF16>
__global__ void k_swiglu_bwd_x8(const bf16_t* __restrict__ h, long ld_h, const void* __restrict__ dout, long ld_do, bf16_t* __restrict__ dh,
                                long ld_dh, long rows, long C, int c8) {
  const long total = rows * c8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c8;
    const int c = (int)(i - r * c8) * 8;
    const F8 a = ld8_bf16(h + r * ld_h + c), g = ld8_bf16(h + r * ld_h + C + c);
    F8 d;
    if constexpr (DO_BF16) d = ld8_bf16((const bf16_t*)dout + r * ld_do + c);
    else d = ld8_f32((const float*)dout + r * ld_do + c);
    F8 da, dg;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float sg = 1.f / (1.f + __expf(-a.v[e]));
      da.v[e] = d.v[e] * g.v[e] * sg * (1.f + a.v[e] * (1.f - sg));
      dg.v[e] = d.v[e] * a.v[e] * sg;
    }
    st8_bf16(dh + r * ld_dh + c, da);
    st8_bf16(dh + r * ld_dh + C + c, dg);
  }
}

template <typename T>
__global__ void k_dropout_mask(T* __restrict__ out, long n, float p, float keep_scale, uint64_t seed, uint64_t offset) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float u = (hash_u32(seed, offset + (uint64_t)i) >> 8) * (1.0f / 16777216.0f);
    st_f32(out + i, u >= p ? keep_scale : 0.f);
  }
}
extern "C" int vfm_dropout_mask(void* out, int dt, long n, float p, uint64_t seed, uint64_t offset, void* stream) {
  VFM_CHECK(p >= 0.f && p < 1.f, VFM_E_INVAL, "vfm_dropout_mask: p");
  if (n == 0) return VFM_OK;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  const float ks = 1.0f / (1.0f - p);
  if (dt == VFM_F32) hipLaunchKernelGGL(k_dropout_mask<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float*)out, n, p, ks, seed, offset);
  else if (dt == VFM_BF16 && n % 8 == 0 && al16(out))
    hipLaunchKernelGGL(k_dropout_mask_bf16x8, dim3(grid8(n / 8)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)out, n / 8, p, ks, seed, offset);
  else if (dt == VFM_BF16) hipLaunchKernelGGL(k_dropout_mask<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (bf16_t*)out, n, p, ks, seed, offset);
  else VFM_FAIL(VFM_E_INVAL, "vfm_dropout_mask: dtype");
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

__global__ void k_mul_mask(const void* __restrict__ src, int src_dt, long ld_src, const void* __restrict__ mask, int mask_dt,
                           long mask_ld, long rpg, void* __restrict__ dst, int dst_dt, long ld_dst, long rows, long cols) {
  const long total = rows * cols;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols, c = i - r * cols;
    const float m = ld_any(mask, (r / rpg) * mask_ld + c, mask_dt);
    st_any(dst, r * ld_dst + c, dst_dt, ld_any(src, r * ld_src + c, src_dt) * m);
  }
}
// eight columns per thread for any mix of the 16-bit type and fp32 (the decoder's Dropout draws fp32 multipliers for bf16 activations: the
// element-per-thread kernel above moved those 16 MB at 1.7 TB/s, 16 launches per train step)
template <typename TS, typename TM, typename TD>
__global__ void k_mul_mask_x8(const TS* __restrict__ src, long ld_src, const TM* __restrict__ mask, long mask_ld, long rpg, TD* __restrict__ dst,
                              long ld_dst, long rows, int cols8) {
  const long total = rows * cols8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols8;
    const int c = (int)(i - r * cols8) * 8;
    F8 a, m;
    if constexpr (sizeof(TS) == 2) a = ld8_bf16(src + r * ld_src + c); else a = ld8_f32(src + r * ld_src + c);
    if constexpr (sizeof(TM) == 2) m = ld8_bf16(mask + (r / rpg) * mask_ld + c); else m = ld8_f32(mask + (r / rpg) * mask_ld + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) a.v[e] *= m.v[e];
    if constexpr (sizeof(TD) == 2) st8_bf16(dst + r * ld_dst + c, a); else st8_f32(dst + r * ld_dst + c, a);
  }
}
template <typename TS, typename TM, typename TD>
static void launch_mul_mask_x8(const void* src, long ld_src, const void* mask, long mask_ld, long rpg, void* dst, long ld_dst, long rows, long cols,
                               hipStream_t s) {
  hipLaunchKernelGGL((k_mul_mask_x8<TS, TM, TD>), dim3(grid8(rows * cols / 8)), dim3(256), 0, s, (const TS*)src, ld_src, (const TM*)mask, mask_ld, rpg,
                     (TD*)dst, ld_dst, rows, (int)(cols / 8));
}
extern "C" int vfm_mul_mask(const void* src, int src_dt, long ld_src, const void* mask, int mask_dt, long mask_ld,
                            long rows_per_group, void* dst, int dst_dt, long ld_dst, long rows, long cols, void* stream) {
  VFM_CHECK(rows_per_group >= 1, VFM_E_INVAL, "vfm_mul_mask: rows_per_group");
  const long total = rows * cols;
  if (total == 0) return VFM_OK;
  auto h16 = [](int dt) { return dt == VFM_BF16; };
  auto okdt = [](int dt) { return dt == VFM_BF16 || dt == VFM_F32; };
  if (okdt(src_dt) && okdt(mask_dt) && okdt(dst_dt) && !(h16(src_dt) && h16(mask_dt) && h16(dst_dt)) && cols % 8 == 0 && ld_src % 8 == 0 &&
      mask_ld % 8 == 0 && ld_dst % 8 == 0 && al16(src) && al16(mask) && al16(dst)) {
    hipStream_t s = (hipStream_t)stream;
    const int key = (h16(src_dt) ? 4 : 0) | (h16(mask_dt) ? 2 : 0) | (h16(dst_dt) ? 1 : 0);
#define MM(TS, TM, TD) launch_mul_mask_x8<TS, TM, TD>(src, ld_src, mask, mask_ld, rows_per_group, dst, ld_dst, rows, cols, s)
    switch (key) {
      case 0: MM(float, float, float); break;
      case 1: MM(float, float, bf16_t); break;
      case 2: MM(float, bf16_t, float); break;
      case 3: MM(float, bf16_t, bf16_t); break;
      case 4: MM(bf16_t, float, float); break;
      case 5: MM(bf16_t, float, bf16_t); break;
      default: MM(bf16_t, bf16_t, float); break;
    }
#undef MM
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  if (src_dt == VFM_BF16 && mask_dt == VFM_BF16 && dst_dt == VFM_BF16 && cols % 8 == 0 && ld_src % 8 == 0 && mask_ld % 8 == 0 && ld_dst % 8 == 0 &&
      al16(src) && al16(mask) && al16(dst)) {
    hipLaunchKernelGGL(k_mul_mask_bf16x8, dim3(grid8(total / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, (const bf16_t*)mask,
                       mask_ld, rows_per_group, (bf16_t*)dst, ld_dst, rows, (int)(cols / 8));
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_mul_mask, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, src_dt, ld_src, mask, mask_dt, mask_ld,
                     rows_per_group, dst, dst_dt, ld_dst, rows, cols);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ GEGLU
__global__ void k_geglu_fwd(const void* __restrict__ h, int h_dt, long ld_h, void* __restrict__ out, int out_dt, long ld_out,
                            long rows, long C) {
  const long total = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C, c = i - r * C;
    const float a = ld_any(h, r * ld_h + c, h_dt), g = ld_any(h, r * ld_h + C + c, h_dt);
    st_any(out, r * ld_out + c, out_dt, a * gelu_f(g));
  }
}
__global__ void k_geglu_bwd(const void* __restrict__ h, int h_dt, long ld_h, const void* __restrict__ dout, int do_dt,
                            long ld_do, void* __restrict__ dh, int dh_dt, long ld_dh, long rows, long C) {
  const long total = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C, c = i - r * C;
    const float a = ld_any(h, r * ld_h + c, h_dt), g = ld_any(h, r * ld_h + C + c, h_dt);
    const float d = ld_any(dout, r * ld_do + c, do_dt);
    st_any(dh, r * ld_dh + c, dh_dt, d * gelu_f(g));
    st_any(dh, r * ld_dh + C + c, dh_dt, d * a * gelu_grad_f(g));
  }
}
// eight columns per thread, 16-bit h / dout / dh (the decoder's three GEGLU feed-forwards: the element-per-thread kernels took 9 + 12 us each)
__global__ void k_geglu_fwd_x8(const bf16_t* __restrict__ h, long ld_h, bf16_t* __restrict__ out, long ld_out, long rows, long C, int c8) {
  const long total = rows * c8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c8;
    const int c = (int)(i - r * c8) * 8;
    F8 a = ld8_bf16(h + r * ld_h + c);
    const F8 g = ld8_bf16(h + r * ld_h + C + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) a.v[e] *= gelu_f(g.v[e]);
    st8_bf16(out + r * ld_out + c, a);
  }
}
__global__ void k_geglu_bwd_x8(const bf16_t* __restrict__ h, long ld_h, const bf16_t* __restrict__ dout, long ld_do, bf16_t* __restrict__ dh,
                               long ld_dh, long rows, long C, int c8) {
  const long total = rows * c8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c8;
    const int c = (int)(i - r * c8) * 8;
    const F8 a = ld8_bf16(h + r * ld_h + c), g = ld8_bf16(h + r * ld_h + C + c), d = ld8_bf16(dout + r * ld_do + c);
    F8 da, dg;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      da.v[e] = d.v[e] * gelu_f(g.v[e]);
      dg.v[e] = d.v[e] * a.v[e] * gelu_grad_f(g.v[e]);
    }
    st8_bf16(dh + r * ld_dh + c, da);
    st8_bf16(dh + r * ld_dh + C + c, dg);
  }
}
extern "C" int vfm_geglu_fwd(const void* h, int h_dt, long ld_h, void* out, int out_dt, long ld_out, long rows, long C,
                             void* stream) {
  const long total = rows * C;
  if (total == 0) return VFM_OK;
  if (h_dt == VFM_BF16 && out_dt == VFM_BF16 && C % 8 == 0 && ld_h % 8 == 0 && ld_out % 8 == 0 && al16(h) && al16(out)) {
    hipLaunchKernelGGL(k_geglu_fwd_x8, dim3(grid8(total / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, ld_h, (bf16_t*)out, ld_out, rows, C,
                       (int)(C / 8));
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_geglu_fwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, h_dt, ld_h, out, out_dt, ld_out, rows, C);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
extern "C" int vfm_geglu_bwd(const void* h, int h_dt, long ld_h, const void* dout, int do_dt, long ld_do, void* dh, int dh_dt,
                             long ld_dh, long rows, long C, void* stream) {
  const long total = rows * C;
  if (total == 0) return VFM_OK;
  if (h_dt == VFM_BF16 && do_dt == VFM_BF16 && dh_dt == VFM_BF16 && C % 8 == 0 && ld_h % 8 == 0 && ld_do % 8 == 0 && ld_dh % 8 == 0 && al16(h) &&
      al16(dout) && al16(dh)) {
    hipLaunchKernelGGL(k_geglu_bwd_x8, dim3(grid8(total / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, ld_h, (const bf16_t*)dout, ld_do,
                       (bf16_t*)dh, ld_dh, rows, C, (int)(C / 8));
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_geglu_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, h_dt, ld_h, dout, do_dt, ld_do, dh, dh_dt,
                     ld_dh, rows, C);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ SwiGLU
// eva_02.py:235-242: hidden = silu(x1) * x2 with h = [x1 | x2] ([rows, 2C])
__global__ void k_swiglu_fwd(const void* __restrict__ h, int h_dt, long ld_h, void* __restrict__ out, int out_dt, long ld_out,
                             long rows, long C) {
  const long total = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C, c = i - r * C;
    const float a = ld_any(h, r * ld_h + c, h_dt), g = ld_any(h, r * ld_h + C + c, h_dt);
    st_any(out, r * ld_out + c, out_dt, a / (1.f + __expf(-a)) * g);
  }
}
__global__ void k_swiglu_bwd(const void* __restrict__ h, int h_dt, long ld_h, const void* __restrict__ dout, int do_dt,
                             long ld_do, void* __restrict__ dh, int dh_dt, long ld_dh, long rows, long C) {
  const long total = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C, c = i - r * C;
    const float a = ld_any(h, r * ld_h + c, h_dt), g = ld_any(h, r * ld_h + C + c, h_dt);
    const float d = ld_any(dout, r * ld_do + c, do_dt);
    const float sg = 1.f / (1.f + __expf(-a));
    st_any(dh, r * ld_dh + c, dh_dt, d * g * sg * (1.f + a * (1.f - sg)));
    st_any(dh, r * ld_dh + C + c, dh_dt, d * a * sg);
  }
}
extern "C" int vfm_swiglu_fwd(const void* h, int h_dt, long ld_h, void* out, int out_dt, long ld_out, long rows, long C,
                              void* stream) {
  const long total = rows * C;
  if (total == 0) return VFM_OK;
  if (h_dt == VFM_BF16 && (out_dt == VFM_BF16 || out_dt == VFM_F32) && C % 8 == 0 && ld_h % 8 == 0 && ld_out % 8 == 0 && al16(h) && al16(out)) {
    if (out_dt == VFM_BF16)
      hipLaunchKernelGGL(k_swiglu_fwd_x8<true>, dim3(grid8(total / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, ld_h, out, ld_out, rows, C, (int)(C / 8));
    else
      hipLaunchKernelGGL(k_swiglu_fwd_x8<false>, dim3(grid8(total / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, ld_h, out, ld_out, rows, C, (int)(C / 8));
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_swiglu_fwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, h_dt, ld_h, out, out_dt, ld_out, rows, C);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
extern "C" int vfm_swiglu_bwd(const void* h, int h_dt, long ld_h, const void* dout, int do_dt, long ld_do, void* dh, int dh_dt,
                              long ld_dh, long rows, long C, void* stream) {
  const long total = rows * C;
  if (total == 0) return VFM_OK;
  if (h_dt == VFM_BF16 && dh_dt == VFM_BF16 && (do_dt == VFM_BF16 || do_dt == VFM_F32) && C % 8 == 0 && ld_h % 8 == 0 && ld_do % 8 == 0 &&
      ld_dh % 8 == 0 && al16(h) && al16(dout) && al16(dh)) {
    if (do_dt == VFM_BF16)
      hipLaunchKernelGGL(k_swiglu_bwd_x8<true>, dim3(grid8(total / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, ld_h, dout, ld_do, (bf16_t*)dh, ld_dh, rows, C, (int)(C / 8));
    else
      hipLaunchKernelGGL(k_swiglu_bwd_x8<false>, dim3(grid8(total / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, ld_h, dout, ld_do, (bf16_t*)dh, ld_dh, rows, C, (int)(C / 8));
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_swiglu_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, h_dt, ld_h, dout, do_dt, ld_do, dh, dh_dt,
                     ld_dh, rows, C);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ RoPE
// eva_02.py:119-160 / 362-369: y = x*cos + rotate_half(x)*sin on the q and k heads of the patch tokens, in place.
// x: [rows, ld], columns [0, ncols) are consecutive heads of width d; token t = row % np indexes the [np, d] tables.
// inverse=1 applies the transpose (the gradient of the rotation).
__global__ void k_rope(void* __restrict__ x, int dt, long ld, long rows, int np, int ncols, int d, const float* __restrict__ cs,
                       const float* __restrict__ sn, int inverse) {
  const long half = ncols / 2;
  const long total = rows * half;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / half;
    const int c0 = (int)(i - r * half) * 2;
    const int t = (int)(r % np), dc = c0 % d;
    const float x0 = ld_any(x, r * ld + c0, dt), x1 = ld_any(x, r * ld + c0 + 1, dt);
    const float c_0 = cs[(long)t * d + dc], c_1 = cs[(long)t * d + dc + 1];
    const float s_0 = sn[(long)t * d + dc], s_1 = sn[(long)t * d + dc + 1];
    float y0, y1;
    if (!inverse) {
      y0 = x0 * c_0 - x1 * s_0;
      y1 = x1 * c_1 + x0 * s_1;
    } else {
      y0 = x0 * c_0 + x1 * s_1;
      y1 = x1 * c_1 - x0 * s_0;
    }
    st_any(x, r * ld + c0, dt, y0);
    st_any(x, r * ld + c0 + 1, dt, y1);
  }
}
__global__ void k_rope_bf16x8(bf16_t* __restrict__ x, long ld, long rows, int np, int c8n, int d, const float* __restrict__ cs,
                              const float* __restrict__ sn, int inverse) {
  const long total = rows * c8n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c8n;
    const int c0 = (int)(i - r * c8n) * 8;
    const int t = (int)(r % np), dc = c0 % d;
    F8 v = ld8_bf16(x + r * ld + c0);
    const F8 c = ld8_f32(cs + (long)t * d + dc), sv = ld8_f32(sn + (long)t * d + dc);
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const float x0 = v.v[e], x1 = v.v[e + 1];
      if (!inverse) {
        v.v[e] = x0 * c.v[e] - x1 * sv.v[e];
        v.v[e + 1] = x1 * c.v[e + 1] + x0 * sv.v[e + 1];
      } else {
        v.v[e] = x0 * c.v[e] + x1 * sv.v[e + 1];
        v.v[e + 1] = x1 * c.v[e + 1] - x0 * sv.v[e];
      }
    }
    st8_bf16(x + r * ld + c0, v);
  }
}
extern "C" int vfm_rope(void* x, int dt, long ld, long rows, int np, int ncols, int d, const float* cos_t, const float* sin_t,
                        int inverse, void* stream) {
  VFM_CHECK(d % 2 == 0 && ncols % d == 0 && np > 0, VFM_E_SHAPE, "vfm_rope: shape");
  const long total = rows * (ncols / 2);
  if (total == 0) return VFM_OK;
  if (dt == VFM_BF16 && d % 8 == 0 && ld % 8 == 0 && al16(x) && al16(cos_t) && al16(sin_t)) {
    hipLaunchKernelGGL(k_rope_bf16x8, dim3(grid8(rows * (ncols / 8))), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, ld, rows, np, ncols / 8, d, cos_t,
                       sin_t, inverse);
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_rope, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, dt, ld, rows, np, ncols, d, cos_t, sin_t, inverse);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ act grad
__global__ void k_act_grad_mul(const void* __restrict__ dy, int dy_dt, long ld_dy, const void* __restrict__ pre, int pre_dt,
                               long ld_pre, void* __restrict__ out, int out_dt, long ld_out, long rows, long cols, int act) {
  const long total = rows * cols;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols, c = i - r * cols;
    const float p = ld_any(pre, r * ld_pre + c, pre_dt);
    const float g = act == VFM_ACT_GELU ? gelu_grad_f(p) : (act == VFM_ACT_RELU ? (p > 0.f ? 1.f : 0.f) : (act == VFM_ACT_QGELU ? qgelu_grad_f(p) : 1.f));
    st_any(out, r * ld_out + c, out_dt, ld_any(dy, r * ld_dy + c, dy_dt) * g);
  }
}
extern "C" int vfm_act_grad_mul(const void* dy, int dy_dt, long ld_dy, const void* pre, int pre_dt, long ld_pre, void* out,
                                int out_dt, long ld_out, long rows, long cols, int act, void* stream) {
  const long total = rows * cols;
  if (total == 0) return VFM_OK;
  if (dy_dt == VFM_BF16 && pre_dt == VFM_BF16 && out_dt == VFM_BF16 && cols % 8 == 0 && ld_dy % 8 == 0 && ld_pre % 8 == 0 && ld_out % 8 == 0 &&
      al16(dy) && al16(pre) && al16(out)) {
    hipLaunchKernelGGL(k_act_grad_mul_bf16x8, dim3(grid8(total / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, ld_dy, (const bf16_t*)pre,
                       ld_pre, (bf16_t*)out, ld_out, rows, (int)(cols / 8), act);
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_act_grad_mul, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, dy_dt, ld_dy, pre, pre_dt, ld_pre, out,
                     out_dt, ld_out, rows, cols, act);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ mask token
__global__ void k_mask_token_fwd(const float* __restrict__ x, const uint8_t* __restrict__ keep, const float* __restrict__ tok,
                                 float* __restrict__ out, long rows, long C) {
  const long total = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C, c = i - r * C;
    out[i] = keep[r] ? x[i] : tok[c];
  }
}
// dx = keep ? dout : 0 ; dtoken[c] = sum over masked rows of dout[r,c].  Block (64 columns, one row chunk) -> partial sums
// ws[chunk][C]; k_colsum2 adds the chunks in a fixed order (deterministic; 4 blocks looping over every row took 136 us).
__global__ void k_mask_token_bwd(const float* __restrict__ dout, const uint8_t* __restrict__ keep, float* __restrict__ dx,
                                 float* __restrict__ ws, long rows, long C, long rows_per_chunk) {
  __shared__ float sh[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long c = (long)blockIdx.x * 64 + tx;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  long r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  float acc = 0.f;
  if (c < C) {
    for (long r = r0 + ty; r < r1; r += 4) {
      const float d = dout[r * C + c];
      const bool k = keep[r] != 0;
      dx[r * C + c] = k ? d : 0.f;
      if (!k) acc += d;
    }
  }
  sh[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && c < C) ws[(long)blockIdx.y * C + c] = (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
}
extern "C" int vfm_mask_token_fwd(const float* x, const uint8_t* keep, const float* token, float* out, long rows, long C,
                                  void* stream) {
  const long total = rows * C;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_mask_token_fwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, keep, token, out, rows, C);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
extern "C" int vfm_mask_token_bwd(const float* dout, const uint8_t* keep, float* dx, float* dtoken, float* ws, long rows, long C,
                                  void* stream) {
  VFM_CHECK(ws, VFM_E_INVAL, "vfm_mask_token_bwd: ws (>= 64*C floats) required");
  if (rows * C == 0) return VFM_OK;
  long nchunk = (rows + 31) / 32;
  if (nchunk > 64) nchunk = 64;
  const long rpc = (rows + nchunk - 1) / nchunk;
  hipLaunchKernelGGL(k_mask_token_bwd, dim3(cdiv(C, 64), (unsigned)nchunk), dim3(256), 0, (hipStream_t)stream, dout, keep, dx, ws, rows,
                     C, rpc);
  hipLaunchKernelGGL(k_colsum2, dim3(cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, ws, C, (int)nchunk, dtoken, 0);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------ reduce_sum
__global__ void k_reduce_sum(const float* __restrict__ x, long n, float scale, float* __restrict__ out) {
  __shared__ float sh[16];
  float a = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) a += x[i];
  a = block_sum(a, sh);
  if (threadIdx.x == 0) out[0] = a * scale;
}
// loss = scale * sum(parts) and acc = 100 * hits / (valid + eps) in one launch; counts = (hits, valid) is then reset to zero
// for the next call (the head keeps one persistent counter pair: no fill launch, no five-kernel ATen chain per head and step)
__global__ void k_ce_finish(const float* __restrict__ x, long n, float scale, int32_t* __restrict__ counts, float eps,
                            float* __restrict__ loss, float* __restrict__ acc) {
  __shared__ float sh[16];
  float a = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) a += x[i];
  a = block_sum(a, sh);
  if (threadIdx.x == 0) {
    loss[0] = a * scale;
    acc[0] = (float)counts[0] * (100.0f / ((float)counts[1] + eps));
    counts[0] = 0, counts[1] = 0;
  }
}
extern "C" int vfm_ce_finish(const float* parts, long n, float scale, int32_t* counts, float eps, float* loss, float* acc,
                             void* stream) {
  VFM_CHECK(parts && counts && loss && acc, VFM_E_INVAL, "vfm_ce_finish: null pointer");
  hipLaunchKernelGGL(k_ce_finish, dim3(1), dim3(1024), 0, (hipStream_t)stream, parts, n, scale, counts, eps, loss, acc);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
extern "C" int vfm_reduce_sum(const float* x, long n, float scale, float* out, void* stream) {
  hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, scale, out);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
