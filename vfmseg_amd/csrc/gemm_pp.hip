// Ping-pong bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(alpha * A[M,K] * B[N,K]^T), 256 x 256 tiles, 8 waves.
//
// The 8 waves form two groups of four (waves 0-3 = rows 0..127 of the tile, waves 4-7 = rows 128..255; wave w&3 owns a
// 64-column strip), so each SIMD holds one wave of either group.  K is walked in 32-wide K-tiles, two PHASES per K-tile
// (phase = 64 rows x 64 columns x K=32 per wave = 8 MFMAs 32x32x16 = 256 matrix-pipe cycles), ONE s_barrier per phase.
// Between two barriers (interval P) a wave runs
//
//     MFMA(P)    8 MFMAs on the fragments read during interval P-1
//     READ(P)    ds_read the fragments of phase P+1, in place (the registers are free once MFMA(P) has been issued)
//     DMA(P)     issue one 16-KiB LDS-DMA chunk (the slow-to-issue part: 2 x global_load_lds_dwordx4)
//
// group 0 in the order MFMA, READ, DMA and group 1 in the order DMA, MFMA, READ: on every SIMD one wave issues its DMA
// while its partner computes, the fragment reads of either wave have most of an interval to land, and the only waits in
// front of the MFMAs are for data issued an interval earlier.
//
// LDS: ring of 4 K-tile slots x (B chunk | A chunk), 16 KiB each.  A chunk = 256 rows x 32 k (64-B rows; 16-B piece c of
// row r is stored at c ^ ((r>>2)&3): applied to the DMA SOURCE address and again on the fragment reads - conflict-free
// ds_read_b128).  Chunk j = 2*ktile + {0: B, 1: A}; it feeds phases R = 2*ktile (B, A rows qm0) and 2*ktile+1 (A rows qm1),
// i.e. it is read from LDS in intervals R-1, and it is issued in interval j-6.
//   RAW: the counted wait at the end of interval P leaves the 3 (P even) / 4 (P odd) youngest chunks in flight, so every
//        chunk that is read in interval P+1 has landed in all waves before the barrier that opens P+1.
//   WAR: the reads issued in interval P are complete (lgkmcnt(0)) in every wave before it leaves interval P+1; a slot is
//        re-issued at least 3 intervals after its last read (B: read in 2u-1, re-issued in 2u+2; A: 2u, 2u+3).
#include "gemm_dev.h"

namespace {
constexpr int PP_CH = 16384;
constexpr int PP_SLOT = 2 * PP_CH;
constexpr int PP_RING = 4 * PP_SLOT;                 // 128 KiB
constexpr int PP_EPI_BYTES = 8 * 64 * (64 + 4) * 4;  // per wave: two 32-row slabs of its 128 x 64 tile
constexpr int PP_SMEM = PP_EPI_BYTES > PP_RING ? PP_EPI_BYTES : PP_RING;
}  // namespace

template <bool VEC>
__global__ void __launch_bounds__(512)
    k_gemm_pp256(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb, long M, long N, long K, long stride_a,
                 long stride_b, long stride_c, int tiles_m, int tiles_n, EpiParams e, int dbg, SkinnyTail sk) {
  constexpr int BM = 256, BN = 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (sk.nblk > 0 && (int)blockIdx.x >= tiles_m * tiles_n) {  // tail rows of M: extra blocks at the end of the grid
    skinny_tile(sk.A, sk.lda, B, ldb, sk.M, N, K, (long)((int)blockIdx.x - tiles_m * tiles_n) * 32, sk.e, 0, smem);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wc = wave & 3;

  // ---- XCD-aware tile mapping (same as k_gemm_bf16)
  const int ntiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GM = 8;
  const int group = bid / (GM * tiles_n);
  const int first_m = group * GM;
  const int gsz = min(tiles_m - first_m, GM);
  const int tm = first_m + (bid % (GM * tiles_n)) % gsz;
  const int tn = (bid % (GM * tiles_n)) / gsz;
  const long m0 = (long)tm * BM, n0 = (long)tn * BN;
  const long z = blockIdx.y;
  const bf16_t* Ab = A + z * stride_a;
  const bf16_t* Bb = B + z * stride_b;

  // ---- per-lane DMA sources: a piece = 16 rows x 64 B = one wave-instruction; wave w moves pieces 2w, 2w+1 of a chunk.
  // 32-bit byte offsets from the (scalar) matrix base: the DMA then uses the SGPR-base + VGPR-offset addressing form and
  // the K advance is scalar arithmetic (the dispatcher checks that the operands span < 4 GiB).
  unsigned soff[2][2];  // [0: B, 1: A][piece]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 16 + (lane >> 2);
    const int sw = ((lane & 3) ^ ((r >> 2) & 3)) << 3;
    long gm = m0 + r, gn = n0 + r;
    if (gm > M - 1) gm = M - 1;
    if (gn > N - 1) gn = N - 1;
    soff[0][j] = (unsigned)((gn * ldb + sw) * 2);
    soff[1][j] = (unsigned)((gm * lda + sw) * 2);
  }
  auto issue = [&](int c, int u) {  // chunk c of K-tile u -> ring slot u & 3
    char* dst = smem + (u & 3) * PP_SLOT + c * PP_CH + wave * 2048;
    const unsigned long long bv = (unsigned long long)(c ? Ab : Bb) + (unsigned long long)u * 64;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bv), hi = __builtin_amdgcn_readfirstlane((unsigned)(bv >> 32));
    const char* base = (const char*)(((unsigned long long)hi << 32) | lo);  // pinned to SGPRs
    glds16(base + soff[c][0], dst);
    glds16(base + soff[c][1], dst + 1024);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  // fragment read addresses (bytes): ring slot offset + row * 64 + swizzled 16-B piece; [s] = k-step of 16 inside the K-tile
  int ra[2], rb[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int kx = ((2 * s + fh) ^ ((fr >> 2) & 3)) << 4;
    ra[s] = PP_CH + (g * 128 + fr) * 64 + kx;
    rb[s] = (wc * 64 + fr) * 64 + kx;
  }
  auto next_slot = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ra[s] = (ra[s] + PP_SLOT) & (PP_RING - 1);
      rb[s] = (rb[s] + PP_SLOT) & (PP_RING - 1);
    }
  };

  const int nk = (dbg & 2) ? 4 : (int)(K / 32);  // 32-wide K-tiles, >= 4 (checked by the dispatcher)
  bf16x8 af[2][2], bfr[2][2];                    // [i or j][s]

  // X: 0 = steady state; 1..6 = the last six phases (no chunk left to issue; the very last one has nothing left to read)
  auto phase = [&](auto Gc, auto QMc, auto Xc, int u) {
    constexpr int G = decltype(Gc)::value, QM = decltype(QMc)::value, X = decltype(Xc)::value;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G == 1 && X == 0) issue(QM, u + 3);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the fragments of this phase (read during the previous interval)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[QM * 2 + i][j] = VFM_MFMA16(af[i][s], bfr[j][s], acc[QM * 2 + i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (X != 6) {
      if constexpr (QM == 0) {  // next phase: same K-tile, rows qm1
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const bf16x8*>(smem + ra[s] + 4096 + i * 2048);
      } else {  // next phase: next K-tile, rows qm0 and its B fragments
        next_slot();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int s = 0; s < 2; ++s) bfr[j][s] = *reinterpret_cast<const bf16x8*>(smem + rb[s] + j * 2048);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const bf16x8*>(smem + ra[s] + i * 2048);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G == 0 && X == 0) issue(QM, u + 3);
    // chunks that may stay in flight: steady 3 (even phase) / 4 (odd); last six phases 2,2,0,0,0,0
    constexpr int live = X == 0 ? (QM == 0 ? 3 : 4) : (X <= 2 ? 2 : 0);
    wait_vmcnt<2 * live>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: chunks 0..5 = K-tiles 0,1,2; chunks 0,1 land, then the fragments of phase 0 are read
  issue(0, 0);
  issue(1, 0);
  issue(0, 1);
  issue(1, 1);
  issue(0, 2);
  issue(1, 2);
  wait_vmcnt<8>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) bfr[j][s] = *reinterpret_cast<const bf16x8*>(smem + rb[s] + j * 2048);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const bf16x8*>(smem + ra[s] + i * 2048);
  __builtin_amdgcn_sched_barrier(0);

  const unsigned long long t0c = __builtin_readcyclecounter(), t0r = __builtin_amdgcn_s_memrealtime();
  auto mainloop = [&](auto Gc) {  // one straight-line instance per wave group
    int u = 0;
    for (; u < nk - 3; ++u) {
      phase(Gc, IC<0>{}, IC<0>{}, u);
      phase(Gc, IC<1>{}, IC<0>{}, u);
    }
    phase(Gc, IC<0>{}, IC<1>{}, u);
    phase(Gc, IC<1>{}, IC<2>{}, u);
    phase(Gc, IC<0>{}, IC<3>{}, u);
    phase(Gc, IC<1>{}, IC<4>{}, u);
    phase(Gc, IC<0>{}, IC<5>{}, u);
    phase(Gc, IC<1>{}, IC<6>{}, u);
  };
  if (g == 0) mainloop(IC<0>{});
  else mainloop(IC<1>{});
  __builtin_amdgcn_sched_barrier(0);

  if ((dbg & 16) && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {
    unsigned long long* o = (unsigned long long*)e.aux + (blockIdx.x ? 2 : 0);
    o[0] = __builtin_readcyclecounter() - t0c;
    o[1] = __builtin_amdgcn_s_memrealtime() - t0r;
  }
  // ---- epilogue (accumulators -> per-wave fp32 LDS image -> 16-byte rows)
  const long zoff = z * stride_c;
  if (dbg & 1) return;
  __syncthreads();
  float* img = reinterpret_cast<float*>(smem) + wave * 64 * (64 + 4);
  if constexpr (VEC) epi_wave_tile<4, 2, 2>(e, zoff, acc, img, lane, m0 + g * 128, n0 + wc * 64, M, N);
  else epi_scalar<4, 2, 2>(e, zoff, acc, img, lane, m0 + g * 128, n0 + wc * 64, M, N);
}

// ------------------------------------------------------------------------------------------------------------------
// 128 x 128 tiles (the N = 1024 GEMMs of a ViT-L block at M = 4096: exactly one tile per CU).  Same ping-pong structure:
// group g = rows g*64..g*64+63, wave w&3 = a 32-column strip, phase = one 64-wide K-tile = 8 MFMAs per wave; the
// fragments (12 ds_read_b128 per phase) are double-buffered in registers (the accumulator is only 32 VGPRs here), so
//     group 0:  MFMA(P)  READ(P+1)  DMA          group 1:  DMA  READ(P+1)  MFMA(P)
// and every wave ends the interval with its counted vmcnt, lgkmcnt(0) and the barrier.
// LDS: ring of 4 K-tile slots x (A 128 rows x 128 B | B 128 rows x 128 B), the k_gemm_bf16 image (16-B piece c of row r at
// c ^ ((r>>1)&7)), 4 DMA pieces (8 rows x 128 B) per wave and K-tile.  K-tile j is issued in interval j-4 (its slot's last
// reads finished before the barrier that ended interval j-5), is waited for at the end of interval j-2 (two K-tiles = 8
// DMA instructions stay in flight) and read in interval j-1.
namespace {
constexpr int P1_SLOT = 32768;
constexpr int P1_RING = 4 * P1_SLOT;
constexpr int P1_EPI_BYTES = 8 * 64 * (32 + 4) * 4;
constexpr int P1_SMEM = P1_RING;
}  // namespace

template <bool VEC>
__global__ void __launch_bounds__(512)
    k_gemm_pp128(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb, long M, long N, long K, long stride_a,
                 long stride_b, long stride_c, int tiles_m, int tiles_n, EpiParams e, SkinnyTail sk) {
  constexpr int BM = 128, BN = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (sk.nblk > 0 && (int)blockIdx.x >= tiles_m * tiles_n) {  // tail rows of M: extra blocks at the end of the grid
    skinny_tile(sk.A, sk.lda, B, ldb, sk.M, N, K, (long)((int)blockIdx.x - tiles_m * tiles_n) * 32, sk.e, 0, smem);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wc = wave & 3;

  const int ntiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GM = 8;
  const int group = bid / (GM * tiles_n);
  const int first_m = group * GM;
  const int gsz = min(tiles_m - first_m, GM);
  const int tm = first_m + (bid % (GM * tiles_n)) % gsz;
  const int tn = (bid % (GM * tiles_n)) / gsz;
  const long m0 = (long)tm * BM, n0 = (long)tn * BN;
  const long z = blockIdx.y;
  const bf16_t* Ab = A + z * stride_a;
  const bf16_t* Bb = B + z * stride_b;

  // DMA sources: piece = 8 rows x 128 B; wave w moves pieces 2w, 2w+1 of the A image and of the B image
  unsigned soff[2][2];  // [0: A, 1: B][piece]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 8 + (lane >> 3);
    const int sw = ((lane & 7) ^ ((r >> 1) & 7)) << 3;
    long gm = m0 + r, gn = n0 + r;
    if (gm > M - 1) gm = M - 1;
    if (gn > N - 1) gn = N - 1;
    soff[0][j] = (unsigned)((gm * lda + sw) * 2);
    soff[1][j] = (unsigned)((gn * ldb + sw) * 2);
  }
  auto issue = [&](int u) {  // K-tile u -> ring slot u & 3
    char* dst = smem + (u & 3) * P1_SLOT + wave * 2048;
    const char* pa = (const char*)Ab + (long)u * 128;
    const char* pb = (const char*)Bb + (long)u * 128;
    glds16(pa + soff[0][0], dst);
    glds16(pa + soff[0][1], dst + 1024);
    glds16(pb + soff[1][0], dst + 16384);
    glds16(pb + soff[1][1], dst + 16384 + 1024);
  };

  f32x16 acc[2][1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  int ra[4], rb[4];  // byte offsets inside a slot, per k-step s
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int kx = ((2 * s + fh) ^ ((fr >> 1) & 7)) << 4;
    ra[s] = (g * 64 + fr) * 128 + kx;
    rb[s] = 16384 + (wc * 32 + fr) * 128 + kx;
  }
  const int nk = (int)(K / 64);  // >= 4 (checked by the dispatcher)
  bf16x8 af[2][2][4], bfr[2][4];  // [buffer][i][s], [buffer][s]

  auto read = [&](auto Bc, int u) {  // fragments of K-tile u -> register buffer Bc
    constexpr int BUF = decltype(Bc)::value;
    const char* sl = smem + (u & 3) * P1_SLOT;
#pragma unroll
    for (int s = 0; s < 4; ++s) bfr[BUF][s] = *reinterpret_cast<const bf16x8*>(sl + rb[s]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 4; ++s) af[BUF][i][s] = *reinterpret_cast<const bf16x8*>(sl + ra[s] + i * 4096);
  };
  // X: 0 steady; 1..4 = the last four phases (nothing left to issue; the last one has nothing left to read)
  auto phase = [&](auto Gc, auto Bc, auto Xc, int u) {
    constexpr int G = decltype(Gc)::value, BUF = decltype(Bc)::value, X = decltype(Xc)::value;
    auto mfma = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][0] = VFM_MFMA16(af[BUF][i][s], bfr[BUF][s], acc[i][0]);
      __builtin_amdgcn_s_setprio(0);
    };
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G == 0) {
      mfma();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (X != 4) read(IC<BUF ^ 1>{}, u + 1);
      if constexpr (X == 0) issue(u + 4);
    } else {
      if constexpr (X == 0) issue(u + 4);
      if constexpr (X != 4) read(IC<BUF ^ 1>{}, u + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma();
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int live = X == 0 ? 2 : (X == 1 ? 1 : 0);  // K-tiles that may stay in flight
    wait_vmcnt<4 * live>();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: K-tiles 0..3 in flight, 0 and 1 landed, fragments of K-tile 0 in buffer 0
  issue(0);
  issue(1);
  issue(2);
  issue(3);
  wait_vmcnt<8>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  read(IC<0>{}, 0);
  __builtin_amdgcn_sched_barrier(0);

  const int its = (nk - 4) / 2;
  auto mainloop = [&](auto Gc) {
    int u = 0;
    for (int it = 0; it < its; ++it) {
      phase(Gc, IC<0>{}, IC<0>{}, u);
      phase(Gc, IC<1>{}, IC<0>{}, u + 1);
      u += 2;
    }
    if ((nk - 4) & 1) {
      phase(Gc, IC<0>{}, IC<0>{}, u);
      phase(Gc, IC<1>{}, IC<1>{}, u + 1);
      phase(Gc, IC<0>{}, IC<2>{}, u + 2);
      phase(Gc, IC<1>{}, IC<3>{}, u + 3);
      phase(Gc, IC<0>{}, IC<4>{}, u + 4);
    } else {
      phase(Gc, IC<0>{}, IC<1>{}, u);
      phase(Gc, IC<1>{}, IC<2>{}, u + 1);
      phase(Gc, IC<0>{}, IC<3>{}, u + 2);
      phase(Gc, IC<1>{}, IC<4>{}, u + 3);
    }
  };
  if (g == 0) mainloop(IC<0>{});
  else mainloop(IC<1>{});
  __builtin_amdgcn_sched_barrier(0);

  const long zoff = z * stride_c;
  __syncthreads();
  float* img = reinterpret_cast<float*>(smem) + wave * 64 * (32 + 4);
  if constexpr (VEC) epi_wave_tile<2, 1, 2>(e, zoff, acc, img, lane, m0 + g * 64, n0 + wc * 32, M, N);
  else epi_scalar<2, 1, 2>(e, zoff, acc, img, lane, m0 + g * 64, n0 + wc * 32, M, N);
}

template <bool VEC>
static bool launch_pp128_t(const vfm_gemm_desc* d, hipStream_t s, const vfm_gemm_desc* tail) {
  const int tiles_m = cdiv(d->M, 128), tiles_n = cdiv(d->N, 128);
  const long batch = d->batch > 0 ? d->batch : 1;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_gemm_pp128<VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, P1_SMEM);
    attr = true;
  }
  SkinnyTail sk;
  sk.nblk = 0;
  const bool fold = tail && batch == 1;
  if (fold) sk.A = (const bf16_t*)tail->A, sk.lda = tail->sa_m, sk.M = tail->M, sk.nblk = cdiv(tail->N, 32), sk.e = make_epi(tail);
  hipLaunchKernelGGL((k_gemm_pp128<VEC>), dim3(tiles_m * tiles_n + sk.nblk, (unsigned)batch), dim3(512), P1_SMEM, s, (const bf16_t*)d->A,
                     d->sa_m, (const bf16_t*)d->B, d->sb_n, d->M, d->N, d->K, d->stride_a, d->stride_b, d->stride_c, tiles_m, tiles_n,
                     make_epi(d), sk);
  return fold || !tail;
}
bool vfm_gemm_launch_pp128(const vfm_gemm_desc* d, hipStream_t s, bool vec, const vfm_gemm_desc* tail) {
  if (vec) return launch_pp128_t<true>(d, s, tail);
  return launch_pp128_t<false>(d, s, tail);
}

int g_pp_dbg = 0;
template <bool VEC>
static bool launch_pp256_t(const vfm_gemm_desc* d, hipStream_t s, const vfm_gemm_desc* tail) {
  const int tiles_m = cdiv(d->M, 256), tiles_n = cdiv(d->N, 256);
  const long batch = d->batch > 0 ? d->batch : 1;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_gemm_pp256<VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_SMEM);
    attr = true;
  }
  SkinnyTail sk;
  sk.nblk = 0;
  const bool fold = tail && batch == 1;
  if (fold) sk.A = (const bf16_t*)tail->A, sk.lda = tail->sa_m, sk.M = tail->M, sk.nblk = cdiv(tail->N, 32), sk.e = make_epi(tail);
  hipLaunchKernelGGL((k_gemm_pp256<VEC>), dim3(tiles_m * tiles_n + sk.nblk, (unsigned)batch), dim3(512), PP_SMEM, s, (const bf16_t*)d->A,
                     d->sa_m, (const bf16_t*)d->B, d->sb_n, d->M, d->N, d->K, d->stride_a, d->stride_b, d->stride_c, tiles_m, tiles_n,
                     make_epi(d), g_pp_dbg, sk);
  return fold || !tail;
}
bool vfm_gemm_launch_pp256(const vfm_gemm_desc* d, hipStream_t s, bool vec, const vfm_gemm_desc* tail) {
  if (vec) return launch_pp256_t<true>(d, s, tail);
  return launch_pp256_t<false>(d, s, tail);
}
