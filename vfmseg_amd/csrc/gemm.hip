// GEMM kernels: C[M,N] = epilogue(alpha * A[M,K] * B[N,K]^T)
//   k_gemm_f32  : exact fp32 (v_mfma_f32_32x32x2_f32), arbitrary strides - the parity path
//   k_gemm_bf16 : bf16 MFMA (v_mfma_f32_32x32x16_bf16), LDS-DMA staged, XOR-swizzled LDS, double buffered
#include "gemm_epilogue.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

// ====================================================================================================== bf16 path
// Tile BM x BN x 64, 256 threads = 4 waves (2 x 2), wave tile (BM/2) x (BN/2) built from 32x32x16 MFMAs.
// LDS image per operand per stage: [rows][64 k] bf16 = 128 B per row, 16-B chunk c of row r stored at chunk
// c ^ ((r >> 1) & 7)  -> conflict-free ds_read_b128 for the MFMA fragment reads (16 lanes x 16 B cover all 64 banks).
// Global -> LDS goes through global_load_lds_dwordx4 (LDS-DMA): the destination is lane-linear, so the swizzle is
// applied to the per-lane SOURCE address and again on the read (cdna_hip_programming.md 5.4 rule 21).
#define BK 64

template <int BM, int BN>
struct GemmCfg {
  static constexpr int A_BYTES = BM * BK * 2;
  static constexpr int B_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int A_INSTR = BM / 8 / 4;  // 1 KiB (8 rows) per wave-instruction, 4 waves
  static constexpr int B_INSTR = BN / 8 / 4;
  static constexpr int WM = BM / 2, WN = BN / 2;
  static constexpr int MI = WM / 32, NI = WN / 32;
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int BM, int BN>
__global__ void __launch_bounds__(256, 2)
    k_gemm_bf16(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb, long M, long N, long K,
                long stride_a, long stride_b, long stride_c, int tiles_m, int tiles_n, EpiParams e) {
  using Cfg = GemmCfg<BM, BN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware tile mapping: hardware deals blocks round-robin over 8 XCDs; give each XCD a contiguous run of
  // logical tiles, ordered in groups of 8 tile-rows so neighbours share A/B panels in that XCD's L2.
  const int ntiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective for any ntiles
  }
  const int GM = 8;
  const int group = bid / (GM * tiles_n);
  const int first_m = group * GM;
  const int gsz = min(tiles_m - first_m, GM);
  const int tm = first_m + (bid % (GM * tiles_n)) % gsz;
  const int tn = (bid % (GM * tiles_n)) / gsz;
  const long m0 = (long)tm * BM, n0 = (long)tn * BN;
  const long z = blockIdx.y;
  const bf16_t* Ab = A + z * stride_a;
  const bf16_t* Bb = B + z * stride_b;

  // ---- per-lane source pointers for the LDS-DMA pieces (row clamp keeps every load in bounds)
  const int lr = lane >> 3;  // row within the 8-row piece
  const int pc = lane & 7;   // physical chunk
  const bf16_t* a_src[Cfg::A_INSTR];
  const bf16_t* b_src[Cfg::B_INSTR];
#pragma unroll
  for (int j = 0; j < Cfg::A_INSTR; ++j) {
    const int r = (wave * Cfg::A_INSTR + j) * 8 + lr;
    const int c = pc ^ ((r >> 1) & 7);
    long gm = m0 + r;
    if (gm > M - 1) gm = M - 1;
    a_src[j] = Ab + gm * lda + c * 8;
  }
#pragma unroll
  for (int j = 0; j < Cfg::B_INSTR; ++j) {
    const int r = (wave * Cfg::B_INSTR + j) * 8 + lr;
    const int c = pc ^ ((r >> 1) & 7);
    long gn = n0 + r;
    if (gn > N - 1) gn = N - 1;
    b_src[j] = Bb + gn * ldb + c * 8;
  }
  auto stage = [&](int buf, long k0) {
    char* sa = smem + buf * Cfg::STAGE_BYTES;
    char* sb = sa + Cfg::A_BYTES;
#pragma unroll
    for (int j = 0; j < Cfg::A_INSTR; ++j) glds16(a_src[j] + k0, sa + (wave * Cfg::A_INSTR + j) * 1024);
#pragma unroll
    for (int j = 0; j < Cfg::B_INSTR; ++j) glds16(b_src[j] + k0, sb + (wave * Cfg::B_INSTR + j) * 1024);
  };

  f32x16 acc[Cfg::MI][Cfg::NI];
#pragma unroll
  for (int i = 0; i < Cfg::MI; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets: lane reads row (l&31) of a 32-row sub-tile, logical chunk 2*s + (l>>5) at k-step s
  const int fr = lane & 31, fh = lane >> 5;
  int a_off[Cfg::MI], a_sw[Cfg::MI], b_off[Cfg::NI], b_sw[Cfg::NI];
#pragma unroll
  for (int i = 0; i < Cfg::MI; ++i) {
    const int r = wm * Cfg::WM + i * 32 + fr;
    a_off[i] = r * 128;
    a_sw[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < Cfg::NI; ++j) {
    const int r = wn * Cfg::WN + j * 32 + fr;
    b_off[j] = r * 128;
    b_sw[j] = (r >> 1) & 7;
  }

  const int nk = (int)(K / BK);
  stage(0, 0);
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (t + 1 < nk) {
      stage(buf ^ 1, (long)(t + 1) * BK);
      // tile t's pieces are the older ones: leave only tile t+1's in flight
      if constexpr (Cfg::A_INSTR + Cfg::B_INSTR == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (Cfg::A_INSTR + Cfg::B_INSTR == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if constexpr (Cfg::A_INSTR + Cfg::B_INSTR == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const char* sa = smem + buf * Cfg::STAGE_BYTES;
    const char* sb = sa + Cfg::A_BYTES;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 af[Cfg::MI], bfr[Cfg::NI];
#pragma unroll
      for (int i = 0; i < Cfg::MI; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(sa + a_off[i] + (((2 * s + fh) ^ a_sw[i]) << 4));
#pragma unroll
      for (int j = 0; j < Cfg::NI; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(sb + b_off[j] + (((2 * s + fh) ^ b_sw[j]) << 4));
#pragma unroll
      for (int i = 0; i < Cfg::MI; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // all reads of `buf` done before iteration t+1 restages it
  }

  const long zoff = z * stride_c;
#pragma unroll
  for (int i = 0; i < Cfg::MI; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm * Cfg::WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        const long n = n0 + wn * Cfg::WN + j * 32 + fr;
        if (m < M && n < N) epi_store(e, zoff, m, n, acc[i][j][r]);
      }
}

template <int BM, int BN>
static int launch_bf16(const vfm_gemm_desc* d, hipStream_t s) {
  using Cfg = GemmCfg<BM, BN>;
  const int tiles_m = cdiv(d->M, BM), tiles_n = cdiv(d->N, BN);
  const size_t shm = 2 * Cfg::STAGE_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)k_gemm_bf16<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr_set = true;
  }
  const long batch = d->batch > 0 ? d->batch : 1;
  hipLaunchKernelGGL((k_gemm_bf16<BM, BN>), dim3(tiles_m * tiles_n, (unsigned)batch), dim3(256), shm, s, (const bf16_t*)d->A,
                     d->sa_m, (const bf16_t*)d->B, d->sb_n, d->M, d->N, d->K, d->stride_a, d->stride_b, d->stride_c, tiles_m,
                     tiles_n, make_epi(d));
  return 0;
}

extern "C" int vfm_gemm(const vfm_gemm_desc* d, void* stream) {
  VFM_CHECK(d && d->A && d->B && d->C, VFM_E_INVAL, "vfm_gemm: null operand");
  VFM_CHECK(d->M >= 0 && d->N >= 0 && d->K >= 0, VFM_E_SHAPE, "vfm_gemm: negative dim");
  VFM_CHECK(d->ldc >= d->N, VFM_E_SHAPE, "vfm_gemm: ldc < N");
  VFM_CHECK(!(d->ep_mode == VFM_EP_MUL_GELU_GRAD || d->ep_mode == VFM_EP_MUL) || d->aux, VFM_E_INVAL, "vfm_gemm: aux missing");
  if (d->M == 0 || d->N == 0) return VFM_OK;
  hipStream_t s = (hipStream_t)stream;
  const long batch = d->batch > 0 ? d->batch : 1;
  if (d->in_dt == VFM_F32) {
    dim3 grid(cdiv(d->N, F32_BN), cdiv(d->M, F32_BM), (unsigned)batch);
    hipLaunchKernelGGL(k_gemm_f32, grid, dim3(256), 0, s, (const float*)d->A, (const float*)d->B, d->M, d->N, d->K, d->sa_m,
                       d->sa_k, d->sb_n, d->sb_k, d->stride_a, d->stride_b, d->stride_c, make_epi(d));
  } else if (d->in_dt == VFM_BF16) {
    VFM_CHECK(d->sa_k == 1 && d->sb_k == 1, VFM_E_UNSUPPORTED, "vfm_gemm(bf16): K must be contiguous (pack/transposed copies)");
    VFM_CHECK(d->K % BK == 0 && d->K > 0, VFM_E_ALIGN, "vfm_gemm(bf16): K=%ld must be a positive multiple of 64 (zero-pad)", d->K);
    VFM_CHECK(d->sa_m % 8 == 0 && d->sb_n % 8 == 0 && ((uintptr_t)d->A % 16 == 0) && ((uintptr_t)d->B % 16 == 0) &&
                  d->stride_a % 8 == 0 && d->stride_b % 8 == 0,
              VFM_E_ALIGN, "vfm_gemm(bf16): operands must be 16-byte aligned with lda/ldb %% 8 == 0");
    if (d->N <= 64) launch_bf16<128, 64>(d, s);
    else launch_bf16<128, 128>(d, s);
  } else {
    VFM_FAIL(VFM_E_INVAL, "vfm_gemm: in_dt");
  }
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
