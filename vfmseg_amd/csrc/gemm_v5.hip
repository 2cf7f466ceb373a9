// 256 x 256 bf16 MFMA GEMM, 16x16x32 MFMA form, for gfx950:  C[M,N] = epilogue(alpha * A[M,K] * B[N,K]^T).
//
// Round 4.  What the vendor library's fastest kernel on these shapes does (Custom_Cijk_..._MT256x256x64_MI16x16x1, read from its code
// object: profiles/r04_vendor_kernel_anatomy.md) and the ring kernels of gemm_w4.hip do not:
//   * v_mfma_f32_16x16x32_bf16 instead of 32x32x16: the same FLOPs per cycle, but under an MFMA-dense loop the chip holds a higher clock
//     on the 16 x 16 shape (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15x wall at equal cycles);
//   * ONE wave per SIMD (4 waves, 128 x 128 wave tiles, the 256-register accumulator in AGPRs): per K-tile each wave issues 128 MFMAs,
//     32 fragment reads and 16 LDS-DMA pieces - 0.25 LDS reads and 0.125 DMA pieces per MFMA, half of what 8 waves of 128 x 64 need;
//   * two whole K-tile stages (2 x 64 KiB) with the DMA of tile t+2 issued INTO the stage of tile t as soon as every wave has that
//     tile's fragments in registers (all 32 of them are read during the first 34 MFMAs of the iteration), so a tile's DMA has a full
//     iteration (~2 us) to land: two barriers per K-tile, none of them waiting for memory in steady state.
// Operand staging as in gemm_w4.hip: K-tiles of 64 = whole 128-byte lines per row, LDS-DMA (global_load_lds_dwordx4), 16-byte piece p of
// row r stored at p ^ ((r >> 1) & 7) (applied to the DMA source address and to the fragment reads: conflict-free for the 16-row x 4-piece
// reads of the 16x16x32 operands).  Epilogues, tail-row blocks and the XCD-aware tile map are the shared ones (gemm_dev.h).
//
// Schedule of iteration t (stage s = t & 1; slots = the 128 MFMAs of the K-tile, half h = the k-range [32h, 32h+32)):
//     slots   0.. 30 (even): fragment reads of half 1 of tile t (8 B, then 8 A)              MFMAs of half 0 run from registers
//     slot   34            : lgkmcnt(0), s_barrier  [R]  - stage s is free
//     slots  36.. 81 (3rd) : the 16 DMA pieces of tile t+2 -> stage s
//     slot   90            : vmcnt(16), s_barrier   [C]  - tile t+1 has landed for every wave (only tile t+2's pieces may be in flight)
//     slots  92..122 (even): fragment reads of half 0 of tile t+1 (stage s ^ 1)
//     end                  : lgkmcnt(0)
#include "gemm_dev.h"



template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (N > 0) {
    sfor<N - 1>(f);
    f(IC<N - 1>{});
  }
}

// MI x NI = 16 x 16 blocks of a wave tile; WM_W x WN_W waves; DBG: 1 = no epilogue (timing diagnostics), STAG: waves on odd SIMDs issue
// their memory operations one MFMA later (the vendor kernel's two loop bodies: requests of the four SIMDs do not arrive together)
template <bool VEC, int MI, int NI, int WM_W, int WN_W, int DBG = 0, int STAG = 1>
__global__ void __launch_bounds__(WM_W* WN_W * 64)
    k_gemm_v5(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb, long M, long N, long K, long stride_a,
              long stride_b, long stride_c, int tiles_m, int tiles_n, EpiParams e, SkinnyTail sk) {
  constexpr int WAVES = WM_W * WN_W;
  constexpr int TM = WM_W * MI * 16, TN = WN_W * NI * 16;
  constexpr int STG = (TM + TN) * 128;                       // bytes per stage: A rows, then B rows
  constexpr int PPA = TM / 8 / WAVES, PPB = TN / 8 / WAVES;   // DMA pieces (8 rows x 128 B) per wave and tile
  constexpr int NP = PPA + PPB, NR = MI + NI, NM = MI * NI;   // per tile: pieces; per half: fragment reads, MFMAs
  static_assert(TM % (8 * WAVES) == 0 && TN % (8 * WAVES) == 0, "whole pieces per wave");
  // slot positions (see the header): reads every RS-th slot, DMA every DS-th
  constexpr int RS = 2, gR = RS * NR + 2, gC = 2 * NM - RS * NR - 6, DS = (gC - 4 - (gR + 2)) / NP;
  static_assert(DS >= 1 && gR + 3 + DS * (NP - 1) < gC && gC + 3 + RS * (NR - 1) < 2 * NM && RS * (NR - 1) + 1 < gR,
                "the K-tile is too short for this schedule");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (sk.nblk > 0 && (int)blockIdx.x >= tiles_m * tiles_n) {   // tail rows of M ([cls] tokens): extra blocks at the end of the grid
    skinny_tile<WAVES>(sk.A, sk.lda, B, ldb, sk.M, N, K, (long)((int)blockIdx.x - tiles_m * tiles_n) * 32, sk.e, 0, smem);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN_W, wn = wave % WN_W;

  // ---- XCD-aware tile mapping (as k_gemm_w4)
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
  const long m0 = (long)tm * TM, n0 = (long)tn * TN;
  const long z = blockIdx.y;
  const bf16_t* Ab = A + z * stride_a;
  const bf16_t* Bb = B + z * stride_b;

  // ---- per-lane DMA sources (32-bit byte offsets; the dispatcher checks the spans)
  unsigned soa[PPA], sob[PPB];
#pragma unroll
  for (int j = 0; j < PPA; ++j) {
    const int r = (wave * PPA + j) * 8 + (lane >> 3);
    long gm = m0 + r;
    if (gm > M - 1) gm = M - 1;
    soa[j] = (unsigned)((gm * lda + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
  }
#pragma unroll
  for (int j = 0; j < PPB; ++j) {
    const int r = (wave * PPB + j) * 8 + (lane >> 3);
    long gn = n0 + r;
    if (gn > N - 1) gn = N - 1;
    sob[j] = (unsigned)((gn * ldb + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
  }
  // piece k of K-tile u into stage st: k < PPB -> B piece k, else A piece k - PPB.  Buffer form of the LDS-DMA (buffer_load_dwordx4 ...
  // offen lds): resource descriptor in SGPRs, the per-lane row offset in ONE VGPR that never changes, the K-tile offset as the scalar
  // offset - no vector instruction per piece beside the load itself (the global_load_lds form needs a 64-bit per-lane address: one
  // v_lshl_add_u64 per piece, and an MFMA gap of the 16x16x32 form has room for ONE vector-side instruction: MI355X_MICROARCH.md,
  // "vector-instruction ISSUE cost")
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)0xffffffffu, 0x00020000);
#endif
  auto dma = [&](auto Kc, int u, int st) {
    constexpr int k = decltype(Kc)::value;
    constexpr bool isb = k < PPB;
    constexpr int j = isb ? k : k - PPB;
    if constexpr (DBG & 2) return;   // diagnostic: no operand stream
    char* dst = smem + st * STG + (isb ? TM * 128 : 0) + (wave * (isb ? PPB : PPA) + j) * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(isb ? rb_src : ra_src, (__attribute__((address_space(3))) void*)dst, 16,
                                             (int)(isb ? sob[j] : soa[j]), u * 128, 0, 0);
#endif
  };

  f32x4v acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets inside a stage: row * 128 + swizzled piece; [h] = k-half
  const int fr = lane & 15, fq = lane >> 4;
  int ra[2], rb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int kx = ((4 * h + fq) ^ ((fr >> 1) & 7)) << 4;
    ra[h] = (wm * MI * 16 + fr) * 128 + kx;
    rb[h] = TM * 128 + (wn * NI * 16 + fr) * 128 + kx;
  }
  bf16x8 fa[2][MI], fb[2][NI];   // [half][block]
  if constexpr (DBG & 4) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[h][i] = *reinterpret_cast<const bf16x8*>(smem + ra[h] + i * 2048);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[h][j] = *reinterpret_cast<const bf16x8*>(smem + rb[h] + j * 2048);
    }
  }
  const int nk = (int)(K / 64);  // >= 2 (checked by the dispatcher)

  // read q of the NR fragment reads of half H from stage st: the NI B fragments first, then the MI A fragments
  auto fread = [&](auto Hc, auto Qc, int st) {
    constexpr int h = decltype(Hc)::value, q = decltype(Qc)::value;
    if constexpr (DBG & 4) return;   // diagnostic: no fragment reads (the MFMAs run on whatever the registers hold)
    const char* sb = smem + st * STG;
    if constexpr (q < NI) fb[h][q] = *reinterpret_cast<const bf16x8*>(sb + rb[h] + q * 2048);
    else fa[h][q - NI] = *reinterpret_cast<const bf16x8*>(sb + ra[h] + (q - NI) * 2048);
  };
  // X: 0 steady, 1 = next-to-last K-tile (nothing left to issue), 2 = last K-tile (nothing left to read either).
  // SG (stagger) = 1: every memory operation one slot later than with SG = 0.  The odd waves run that body, so that at any slot only two
  // of the CU's four SIMDs hand a request to the (one) texture-address unit / LDS: with all four at once a 1-KiB DMA piece waits for
  // three others (~48 cycles: three MFMA slots of a wave that has nothing else to issue).  The barriers stay where they are.
  auto iter = [&](auto Xc, auto Sc, int t) {
    constexpr int X = decltype(Xc)::value, SG = decltype(Sc)::value;
    const int st = t & 1;
    sfor<2 * NM>([&](auto Gc) {
      constexpr int g = decltype(Gc)::value, h = g / NM, m = g % NM, i = m / NI, j = m % NI, gs = g - SG;
      acc[i][j] = VFM_MFMA16S(fa[h][i], fb[h][j], acc[i][j]);
      if constexpr (gs >= 0 && gs % RS == 0 && gs / RS < NR) fread(IC<1>{}, IC<gs / RS>{}, st);                          // half 1 of this tile
      if constexpr (g == gR && X == 0) {
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave holds every fragment of stage st
        __builtin_amdgcn_s_barrier();
      }
      if constexpr (X == 0 && gs >= gR + 2 && (gs - gR - 2) % DS == 0 && (gs - gR - 2) / DS < NP) dma(IC<(gs - gR - 2) / DS>{}, t + 2, st);
      if constexpr (g == gC && X <= 1) {
        if constexpr (X == 0) wait_vmcnt<NP>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
      }
      if constexpr (X <= 1 && gs >= gC + 2 && (gs - gC - 2) % RS == 0 && (gs - gC - 2) / RS < NR) fread(IC<0>{}, IC<(gs - gC - 2) / RS>{}, st ^ 1);
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (X <= 1) __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: tiles 0 and 1 in flight, tile 0 landed, fragments of its half 0
  sfor<NP>([&](auto Kc) { dma(Kc, 0, 0); });
  sfor<NP>([&](auto Kc) { dma(Kc, 1, 1); });
  wait_vmcnt<NP>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  sfor<NR>([&](auto Qc) { fread(IC<0>{}, Qc, 0); });
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_sched_barrier(0);

  auto loop = [&](auto Sc) __attribute__((always_inline)) {
    int t = 0;
    for (; t < nk - 2; ++t) iter(IC<0>{}, Sc, t);
    iter(IC<1>{}, Sc, t);
    iter(IC<2>{}, Sc, t + 1);
  };
  if (STAG && (wave & 1)) loop(IC<1>{});   // (two copies of the loop: the slot of every instruction is a compile-time constant)
  else loop(IC<0>{});

  // ---- epilogue (accumulators -> per-wave fp32 LDS image -> 16-byte rows), one 64-row slab of the wave tile at a time
  if constexpr (DBG & 1) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::"v"(acc[i][j]));
#endif
      }
    return;
  }
  const long zoff = z * stride_c;
  __syncthreads();
  constexpr int NI32 = NI / 2;
  float* img = reinterpret_cast<float*>(smem) + wave * 64 * (NI32 * 32 + 4);
  const long mw = m0 + wm * MI * 16, nw = n0 + wn * NI * 16;
  sfor<MI / 4>([&](auto Hc) {
    constexpr int hb = decltype(Hc)::value;
    f32x4v(&a4)[4][NI] = *reinterpret_cast<f32x4v(*)[4][NI]>(&acc[4 * hb]);   // 16-row blocks 4hb .. 4hb+3: one 64-row slab
    if constexpr (VEC) epi_wave_tile_acc<2, NI32, 2>(e, zoff, Acc16<4, NI>{a4}, img, lane, mw + hb * 64, nw, M, N);
    else epi_scalar_acc<2, NI32, 2>(e, zoff, Acc16<4, NI>{a4}, img, lane, mw + hb * 64, nw, M, N);
  });
}

extern int g_pp_dbg;
template <bool VEC, int MI, int NI, int WM_W, int WN_W, int DBG, int STAG = 1>
static bool launch_v5_t(const vfm_gemm_desc* d, hipStream_t s, const vfm_gemm_desc* tail) {
  constexpr int TM = WM_W * MI * 16, TN = WN_W * NI * 16, WAVES = WM_W * WN_W;
  constexpr int RING = 2 * (TM + TN) * 128, EPI = WAVES * 64 * (NI * 16 + 4) * 4, SK = WAVES * 8192;
  constexpr int SMEM = RING > EPI ? (RING > SK ? RING : SK) : (EPI > SK ? EPI : SK);
  const int tiles_m = cdiv(d->M, TM), tiles_n = cdiv(d->N, TN);
  const long batch = d->batch > 0 ? d->batch : 1;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_gemm_v5<VEC, MI, NI, WM_W, WN_W, DBG, STAG>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr = true;
  }
  SkinnyTail sk;
  sk.nblk = 0;
  const bool fold = tail && batch == 1;
  if (fold) sk.A = (const bf16_t*)tail->A, sk.lda = tail->sa_m, sk.M = tail->M, sk.nblk = cdiv(tail->N, 32), sk.e = make_epi(tail);
  hipLaunchKernelGGL((k_gemm_v5<VEC, MI, NI, WM_W, WN_W, DBG, STAG>), dim3(tiles_m * tiles_n + sk.nblk, (unsigned)batch), dim3(WAVES * 64), SMEM, s,
                     (const bf16_t*)d->A, d->sa_m, (const bf16_t*)d->B, d->sb_n, d->M, d->N, d->K, d->stride_a, d->stride_b, d->stride_c,
                     tiles_m, tiles_n, make_epi(d), sk);
  return fold || !tail;
}
// form 0: 256 x 256 tiles, 4 waves (config 50).  K % 64 == 0, K >= 128 (checked by the dispatcher).  Returns whether the tail rows were
// folded into the launch.
bool vfm_gemm_launch_v5(const vfm_gemm_desc* d, hipStream_t s, bool vec, const vfm_gemm_desc* tail, int form) {
  (void)form;
  if (!vec) return launch_v5_t<false, 8, 8, 2, 2, 0>(d, s, tail);
  if (g_pp_dbg == 1) return launch_v5_t<true, 8, 8, 2, 2, 1>(d, s, tail);   // diagnostic: main loop only
  if (g_pp_dbg == 2) return launch_v5_t<true, 8, 8, 2, 2, 0, 0>(d, s, tail);   // diagnostic: no stagger
  if (g_pp_dbg == 3) return launch_v5_t<true, 8, 8, 2, 2, 1, 0>(d, s, tail);   // diagnostic: main loop only, no stagger
  if (g_pp_dbg == 5) return launch_v5_t<true, 8, 8, 2, 2, 3>(d, s, tail);      // ... main loop without the operand stream
  if (g_pp_dbg == 6) return launch_v5_t<true, 8, 8, 2, 2, 5>(d, s, tail);      // ... main loop without the fragment reads
  if (g_pp_dbg == 7) return launch_v5_t<true, 8, 8, 2, 2, 7>(d, s, tail);      // ... MFMAs (and barriers) only
  return launch_v5_t<true, 8, 8, 2, 2, 0>(d, s, tail);
}
