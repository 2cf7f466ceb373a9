// vfm_gemm dispatch + the exact-fp32 kernel (v_mfma_f32_32x32x2_f32, arbitrary strides - the parity path).
// The bf16 MFMA kernel lives in gemm_bf16.hip.
#include "gemm_epilogue.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ====================================================================================================== fp32 path
// 64x64 tile, BK=16, 256 threads = 4 waves in 2x2, each wave one 32x32 MFMA tile.
#define F32_BM 64
#define F32_BN 64
#define F32_BK 16
__global__ void __launch_bounds__(256) k_gemm_f32(const float* __restrict__ A, const float* __restrict__ B, long M, long N, long K,
                                                  long sa_m, long sa_k, long sb_n, long sb_k, long stride_a, long stride_b,
                                                  long stride_c, EpiParams e) {
  __shared__ float As[F32_BM][F32_BK + 1];
  __shared__ float Bs[F32_BN][F32_BK + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long m0 = (long)blockIdx.y * F32_BM, n0 = (long)blockIdx.x * F32_BN;
  const long z = blockIdx.z;
  const float* Ab = A + z * stride_a;
  const float* Bb = B + z * stride_b;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // loader mapping: choose the fastest-varying index along the unit-stride dimension
  const bool a_kfast = (sa_k == 1) || (sa_m != 1);
  const bool b_kfast = (sb_k == 1) || (sb_n != 1);
  for (long k0 = 0; k0 < K; k0 += F32_BK) {
#pragma unroll
    for (int j = 0; j < (F32_BM * F32_BK) / 256; ++j) {
      const int eidx = tid + j * 256;
      int mm, kk;
      if (a_kfast) { mm = eidx / F32_BK; kk = eidx % F32_BK; } else { kk = eidx / F32_BM; mm = eidx % F32_BM; }
      const long gm = m0 + mm, gk = k0 + kk;
      As[mm][kk] = (gm < M && gk < K) ? Ab[gm * sa_m + gk * sa_k] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < (F32_BN * F32_BK) / 256; ++j) {
      const int eidx = tid + j * 256;
      int nn, kk;
      if (b_kfast) { nn = eidx / F32_BK; kk = eidx % F32_BK; } else { kk = eidx / F32_BN; nn = eidx % F32_BN; }
      const long gn = n0 + nn, gk = k0 + kk;
      Bs[nn][kk] = (gn < N && gk < K) ? Bb[gn * sb_n + gk * sb_k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < F32_BK; kk += 2) {
      const float a = As[wm * 32 + (lane & 31)][kk + (lane >> 5)];
      const float b = Bs[wn * 32 + (lane & 31)][kk + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const long zoff = z * stride_c;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const long n = n0 + wn * 32 + (lane & 31);
    if (m < M && n < N) epi_store(e, zoff, m, n, acc[r]);
  }
}

int vfm_gemm_bf16_impl(const vfm_gemm_desc* d, hipStream_t s);

extern "C" int vfm_gemm(const vfm_gemm_desc* d, void* stream) {
  VFM_CHECK(d && d->A && d->B && d->C, VFM_E_INVAL, "vfm_gemm: null operand");
  VFM_CHECK(d->M >= 0 && d->N >= 0 && d->K >= 0, VFM_E_SHAPE, "vfm_gemm: negative dim");
  VFM_CHECK(d->ldc >= d->N, VFM_E_SHAPE, "vfm_gemm: ldc < N");
  if (d->c_dt == VFM_SPLIT3) {
    VFM_CHECK(d->in_dt == VFM_BF16 && !d->C2 && !d->residual && d->c_plane >= d->N && d->c_plane % 8 == 0 && d->ldc >= 3 * d->c_plane && d->ldc % 8 == 0 &&
                  (uintptr_t)d->C % 16 == 0 && d->batch <= 1,
              VFM_E_UNSUPPORTED, "vfm_gemm: c_dt VFM_SPLIT3 needs bf16 inputs, no C2 / residual / batch, c_plane >= N (multiple of 8), ldc >= 3 c_plane, a 16-byte aligned C");
  }
  VFM_CHECK(!(d->ep_mode == VFM_EP_MUL_GELU_GRAD || d->ep_mode == VFM_EP_MUL || d->ep_mode == VFM_EP_MUL_QGELU_GRAD) || d->aux, VFM_E_INVAL, "vfm_gemm: aux missing");
  VFM_CHECK(d->ep_mode != VFM_EP_GELU_DGELU || d->C2, VFM_E_INVAL, "vfm_gemm: VFM_EP_GELU_DGELU needs the second output C2");
  if (d->M == 0 || d->N == 0) return VFM_OK;
  hipStream_t s = (hipStream_t)stream;
  const long batch = d->batch > 0 ? d->batch : 1;
  if (d->in_dt == VFM_F32) {
    dim3 grid(cdiv(d->N, F32_BN), cdiv(d->M, F32_BM), (unsigned)batch);
    hipLaunchKernelGGL(k_gemm_f32, grid, dim3(256), 0, s, (const float*)d->A, (const float*)d->B, d->M, d->N, d->K, d->sa_m,
                       d->sa_k, d->sb_n, d->sb_k, d->stride_a, d->stride_b, d->stride_c, make_epi(d));
  } else if (d->in_dt == VFM_BF16) {
    const bool bt = d->sb_n == 1 && d->sb_k != 1, at = d->sa_m == 1 && d->sa_k != 1;
    VFM_CHECK((d->sa_k == 1 || (at && bt && d->M % 8 == 0)) && (d->sb_k == 1 || (bt && d->N % 8 == 0)), VFM_E_UNSUPPORTED,
              "vfm_gemm(bf16): A must be K-contiguous (or [K,M] together with a [K,N] B, M %% 8 == 0); B either [N,K] K-contiguous or "
              "[K,N] N-contiguous with N %% 8 == 0");
    VFM_CHECK(d->K % 64 == 0 && d->K > 0, VFM_E_ALIGN, "vfm_gemm(bf16): K=%ld must be a positive multiple of 64 (zero-pad)", d->K);
    VFM_CHECK((at ? d->sa_k : d->sa_m) % 8 == 0 && (d->sb_k == 1 ? d->sb_n % 8 == 0 : d->sb_k % 8 == 0) && ((uintptr_t)d->A % 16 == 0) &&
                  ((uintptr_t)d->B % 16 == 0) && d->stride_a % 8 == 0 && d->stride_b % 8 == 0,
              VFM_E_ALIGN, "vfm_gemm(bf16): operands must be 16-byte aligned with lda/ldb %% 8 == 0");
    const int rc = vfm_gemm_bf16_impl(d, s);
    if (rc) return rc;
  } else {
    VFM_FAIL(VFM_E_INVAL, "vfm_gemm: in_dt");
  }
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
