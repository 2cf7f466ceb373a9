// Persistent two-accumulator bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(alpha * A[M,K] * B[N,K]^T).
//
// Why: in the train step the wide GEMMs (fc1 forward, fc2 input gradient: [4096 x 4096 x 1024]) write 34-67 MB per launch.  The
// 128 x 128 / two-blocks-per-CU kernel (gemm_w4.hip) runs its 2.25 rounds of tiles in lockstep - every resident block is in its
// main loop, then every block is in its epilogue - so nothing hides the stores (measured: 36.7 us without the epilogue, 53.5 us
// with it), and the 256 x 256 one-block-per-CU forms finish all tiles at once and expose the whole store burst.  Here ONE block per
// CU walks `per` sub-tiles of 256 x 128 that share their A rows (a 256 x 256 region for per = 2) and every wave keeps TWO
// accumulator sets: while the K loop of sub-tile i+1 runs on the matrix pipe, the same wave pushes sub-tile i through the epilogue
// (LDS transposition, GELU / multiply, 16-byte stores) one 8-row slab per K-tile, so the stores of sub-tile i trickle out under
// the MFMAs of sub-tile i+1 and only the last sub-tile's epilogue is exposed.  256 x 128 sub-tiles stream 0.0117 operand bytes
// per flop (128 x 128: 0.0156) - the L2 -> LDS stream is what bounds these loops (~61-70 GB/s per CU).
//
// Tile: 256 x 128, 8 waves as 4 (rows) x 2 (columns), wave tile 64 x 64 = 2 x 2 blocks of v_mfma_f32_32x32x16_bf16
// (4 MFMAs and 4 fragment reads per 16-wide k-step).
// Operand ring: EIGHT 16-KiB unit slots (a unit = 128 rows x one 128-byte K-tile row, whole cache lines per LDS-DMA request, 16-byte
// piece p of row r stored at p ^ ((r >> 1) & 7) - on the DMA source address and on the fragment reads).  A K-tile is three units
// (A rows 0-127, A rows 128-255, B rows), unit c lives in slot c % 8: 2.67 K-tiles of prefetch.  Wave w moves pieces 2w, 2w+1 of
// every unit (6 global_load_lds_dwordx4 per K-tile).
// Schedule of iteration g (= K-tile g of the block's flat K-tile sequence over all its sub-tiles; fragments double-buffered):
//     k0: MFMAs (g,s0) | reads (g,s1) | DMA A rows 0-127 of K-tile g+2 (2 pieces)   -> slot of unit 3(g-1)+1 (freed at beta_{g-1})
//     k1: MFMAs (g,s1) | reads (g,s2) | DMA A rows 128-255 of K-tile g+2 (piece 0)  -> slot of unit 3(g-1)+2
//     k2: MFMAs (g,s2) | reads (g,s3) | DMA A rows 128-255 of K-tile g+2 (piece 1)
//     beta_g: lgkmcnt(0) (K-tile g fully read), vmcnt(4) (K-tile g+1 landed; the four A pieces above may fly), s_barrier
//     k3: MFMAs (g,s3) | reads (g+1,s0) | DMA B rows of K-tile g+2 (2 pieces)       -> slot of unit 3g (freed at beta_g)
// (unit c = 3 g + {0: A0, 1: A1, 2: B}; before beta_g exactly the units <= 3 g + 7 have been issued: eight slots.)
// The K-tile sequence runs across sub-tile seams without draining: the DMA cursors switch to the next sub-tile's B rows on their own.
//
// Epilogue of the previous sub-tile inside the first 8 iterations of the next one (slab v = 8 rows x 64 columns of the wave tile):
//     k0: accumulator registers of the slab -> wave-private fp32 image (8 ds_write_b32)
//     k1: image rows back, 8 columns per lane (2 ds_read_b128); alpha / bias; first half of the activation arithmetic
//     k2: second half
//     k3: (behind beta: the slab's aux piece has landed) multiply / pack, one or two 16-byte stores per lane
// aux (fc2's gelu' operand) comes in by LDS-DMA as well - one 1-KiB piece per slab into a two-slot wave-private buffer - because
// an ordinary global load beside LDS-DMA makes hipcc wait vmcnt(0) and drain the ring; bias is DMA'd once per block.
// All waits are counted by hand (loads, stores and LDS-DMA share one in-order vmcnt); every shape this kernel accepts is
// tile-aligned, so no store is ever masked off and the counts are exact.
#include "gemm_dev.h"

namespace {

template <int N, typename F>
__device__ __forceinline__ void ps_for(F&& f) {
  if constexpr (N > 0) {
    ps_for<N - 1>(f);
    f(IC<N - 1>{});
  }
}

// LDS reads the compiler must not see: hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of an ordinary LDS load that may alias the
// destination of an LDS-DMA in flight (the epilogue image, the aux / bias pieces) - which drains the operand ring every K-tile.  The
// loads and their wait are ONE statement with early-clobber outputs (cdna_hip_programming.md 5.7 item 1, form (i)).
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void lds_read2x16(unsigned addr, float4& a, float4& b) {   // 32 contiguous bytes
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(addr) : "memory");
}
__device__ __forceinline__ uint4 lds_read16(unsigned addr) {
  uint4 a;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a) : "v"(addr) : "memory");
  return a;
}

// g = gelu(v), dg = gelu'(v), one element, plain (un-packed) f32 VALU: beside MFMAs a v_pk_fma_f32 costs ~22 cycles MORE than the two
// v_fma_f32 it replaces (MI355X_MICROARCH.md, price of one filler beside MFMAs), so the epilogue pieces that ride inside the K loop
// must not use the packed forms (this file is compiled with -fno-slp-vectorize for the same reason).  Same arithmetic as gelu_pair.
template <bool WANT_DG>
__device__ __forceinline__ void gelu1(float v, float& g, float& dg) {
  const float av = fabsf(v);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, av, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float ex = __builtin_amdgcn_exp2f(v * v * (-0.5f * 1.44269504088896340736f));
  const float erfv = copysignf(1.0f - p * t * ex, v);
  const float cdf = fmaf(0.5f, erfv, 0.5f);
  g = v * cdf;
  if constexpr (WANT_DG) dg = fmaf(v * 0.39894228040143267794f, ex, cdf);
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf16x2(f32x2{lo, hi}); }

enum { PS_BIAS = 0, PS_GELU_DGELU = 1, PS_MUL = 2, PS_GELU = 3 };   // epilogue kinds (all write bf16 C)

constexpr int PS_NSLOT = 8, PS_UNIT = 16384, PS_RING = PS_NSLOT * PS_UNIT, PS_WAVE_AREA = 4096;
constexpr int PS_SMEM = PS_RING + 8 * PS_WAVE_AREA;   // 160 KiB
constexpr int PS_PEEL = 8;                            // slabs per wave tile = K-tiles the previous sub-tile's epilogue rides on

// DBG (timing diagnostics, results garbage; vfm_tune pp_dbg): 1 = no epilogue stores, 2 = no epilogue at all, 4 = no MFMAs, 8 = no operand DMA,
// 16 = no fragment reads
// BURST: all six pieces of an iteration go out in k3, right behind beta (the slots K-tile g just freed take B of K-tile g+2 and A of
// K-tile g+3): the A pieces are in flight three quarters of an iteration longer than when they are spread over k0..k2
template <int EK, int DBG = 0, bool BURST = false>
__global__ void __launch_bounds__(512) k_gemm_ps(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb, long M, long N,
                                                 long K, int tiles_m, int sup_n, int per, EpiParams e, SkinnyTail sk) {
  constexpr int MI = 2, NI = 2, WN_W = 2;
  constexpr int BM = 256, BN = 128, UA = 2, UPT = 3;    // units per K-tile: A0, A1, B
  constexpr bool HAS_AUX = EK == PS_MUL, HAS_BIAS = EK != PS_MUL, HAS_C2 = EK == PS_GELU_DGELU;
  constexpr int NST = HAS_C2 ? 2 : 1;   // 16-byte stores per lane and slab
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsup = tiles_m * sup_n;
  if ((int)blockIdx.x >= nsup) {   // the tail rows of M (the [cls] rows) as extra blocks at the end of the grid: they need a whole CU's LDS
    skinny_tile(sk.A, sk.lda, B, ldb, sk.M, N, K, (long)((int)blockIdx.x - nsup) * 32, sk.e, 0, smem);   // and fill in as CUs free up
    return;
  }
  const int wm = wave / WN_W, wn = wave % WN_W;
  const int fr = lane & 31, fh = lane >> 5;

  // ---- XCD-aware super-tile mapping (a super-tile = `per` sub-tiles that share their A rows)
  int bid = blockIdx.x;
  {
    const int q = nsup >> 3, r = nsup & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GM = 8;
  const int group = bid / (GM * sup_n);
  const int first_m = group * GM;
  const int gsz = min(tiles_m - first_m, GM);
  const int tm = first_m + (bid % (GM * sup_n)) % gsz;
  const int ts = (bid % (GM * sup_n)) / gsz;
  const long m0 = (long)tm * BM;
  const long n00 = (long)ts * per * BN;   // first sub-tile's column origin; sub-tile i: n00 + i * BN
  const int nk = (int)(K / 64);           // (the block's flat K-tile sequence has per * nk entries)

  // ---- per-lane DMA sources (32-bit byte offsets from scalar bases)
  const int prow = wave * 16 + (lane >> 3);   // piece j covers rows prow + 8 j of a unit
  unsigned soffA[UA][2], soffB[2];
#pragma unroll
  for (int u = 0; u < UA; ++u)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = prow + 8 * j;
      soffA[u][j] = (unsigned)(((m0 + u * 128 + r) * lda + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
    }
  auto set_soffB = [&](long n0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = prow + 8 * j;
      soffB[j] = (unsigned)(((n0 + r) * ldb + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
    }
  };
  auto sbase = [&](const void* p, long byte_off) -> const char* {   // 64-bit base pinned to SGPRs
    const unsigned long long bv = (unsigned long long)p + (unsigned long long)byte_off;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bv), hi = __builtin_amdgcn_readfirstlane((unsigned)(bv >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
  };
  // DMA cursors: iteration g issues the A units and the B unit of K-tile g+2 of the flat sequence (clamped to the last K-tile at the
  // end: the extra pieces re-fetch valid data into slots nobody reads any more, which keeps the vmcnt arithmetic uniform)
  int cbI = 0, cbU = 0, caU = 0;   // B cursor: sub-tile, K-tile in it; A cursor: K-tile in its sub-tile (A rows are shared)
  auto advB = [&]() {
    if (++cbU == nk) {
      if (cbI + 1 < per) { cbU = 0; ++cbI; set_soffB(n00 + (long)cbI * BN); }
      else cbU = nk - 1;
    }
  };
  int caI = 0;
  auto advA = [&]() {
    if (++caU == nk) {
      if (caI + 1 < per) { caU = 0; ++caI; }
      else caU = nk - 1;
    }
  };
  auto dmaA = [&](auto Uc, auto Jc, int slot) {
    constexpr int u = decltype(Uc)::value, j = decltype(Jc)::value;
    if constexpr (DBG & 8) return;
    glds16(sbase(A, (long)caU * 128) + soffA[u][j], smem + slot * PS_UNIT + (wave * 2 + j) * 1024);
  };
  auto dmaB = [&](auto Jc, int slot) {
    constexpr int j = decltype(Jc)::value;
    if constexpr (DBG & 8) return;
    glds16(sbase(B, (long)cbU * 128) + soffB[j], smem + slot * PS_UNIT + (wave * 2 + j) * 1024);
  };
  auto wrap = [](int p) { return p >= PS_NSLOT ? p - PS_NSLOT : p; };

  // ---- wave-private epilogue area: [fp32 image 8 x 64 : 2 KiB][aux slot 0 : 1 KiB][aux slot 1 / bias : 1 KiB]
  char* warea = smem + PS_RING + wave * PS_WAVE_AREA;
  float* img = reinterpret_cast<float*>(warea);
  char* auxb = warea + 2048;
  const int erow = lane >> 3, ecol = (lane & 7) * 8;   // epilogue lane map: row of the slab, first of its 8 columns

  // ---- fragment read offsets inside a unit slot
  int ra[4], rb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int kx = ((2 * s + fh) ^ ((fr >> 1) & 7)) << 4;
    ra[s] = ((wm & 1) * 64 + fr) * 128 + kx;     // + i * 4096; unit = wm >> 1
    rb[s] = (wn * 64 + fr) * 128 + kx;           // + j * 4096; unit = UA
  }
  const int ua = wm >> 1;

  f32x16 acc[MI][NI], accp[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f, accp[i][j][r] = 0.f;
  bf16x8 fa[2][MI], fb[2][NI];

  // ---- epilogue pieces (previous sub-tile pm0 / pn0; slab S: rows (S >> 2) * 32 + (S & 3) * 8 .. + 7 of the wave tile)
  long pn0 = 0;       // column origin of the sub-tile held in accp
  float ex[8];            // values in flight between the pieces of one slab (fp32 until the activation has run)
  uint32_t cpk[4], dpk[4];   // ... packed bf16 pairs afterwards (registers are scarce: 128 accumulators + 32 fragments)
  // address of this lane's 8 columns of slab S in a [M, ld] bf16 matrix: a wave-uniform 64-bit base (SGPRs) + a 32-bit lane offset
  const unsigned lane_c = (unsigned)(((long)erow * e.ldc + ecol) * 2);
  const unsigned lane_c2 = HAS_C2 ? (unsigned)(((long)erow * e.ldc2 + ecol) * 2) : 0u;
  const unsigned lane_ax = HAS_AUX ? (unsigned)(((long)erow * e.ld_aux + ecol) * 2) : 0u;
  auto slab_base = [&](auto Sc, const void* p, long ld) -> const char* {
    constexpr int S = decltype(Sc)::value;
    return sbase(p, ((m0 + wm * 64 + (S >> 2) * 32 + (S & 3) * 8) * ld + pn0 + wn * 64) * 2);
  };
  auto aux_dma = [&](auto Sc) {   // slab S's aux piece (bf16 8 x 64) -> aux slot S & 1: lane's own 16 bytes
    constexpr int S = decltype(Sc)::value;
    if constexpr (HAS_AUX) glds16(slab_base(Sc, e.aux, e.ld_aux) + lane_ax, auxb + (S & 1) * 1024);
  };
  auto ep_dump = [&](auto Sc) {
    constexpr int S = decltype(Sc)::value, i = S >> 2, q = S & 3;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) img[(r4 + 4 * fh) * 64 + j * 32 + fr] = accp[i][j][4 * q + r4];
  };
  // rows back, alpha, bias.  No wave barrier between dump and read: the LDS queue of a wave is in order, so the reads see the dump, and
  // the compiler keeps may-alias LDS accesses in program order; __builtin_amdgcn_wave_barrier() here made hipcc put s_waitcnt vmcnt(0)
  // in front of the reads (it fences the LDS-DMA in flight), which drains the operand ring every K-tile
  auto ep_read = [&](auto Sc) {
    float4 v0, v1;
    lds_read2x16(lds_addr(img + erow * 64 + ecol), v0, v1);
    const float al = e.alpha;
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    if constexpr (HAS_BIAS) {
      const int bo = (int)(pn0 - n00) + wn * 64 + ecol;   // bias of the block's columns sits in the wave area (DMA'd in the prologue)
      const float* bl = reinterpret_cast<const float*>(auxb + 1024);
      float4 b0, b1;
      lds_read2x16(lds_addr(bl + bo), b0, b1);
      const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int c = 0; c < 8; ++c) ex[c] = fmaf(v[c], al, b[c]);
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) ex[c] = v[c] * al;
    }
  };
  auto ep_act = [&](auto Hc) {     // half H of the activation arithmetic (columns 4H .. 4H+3)
    constexpr int h = decltype(Hc)::value;
    if constexpr (EK == PS_GELU_DGELU || EK == PS_GELU) {
      float g[4], dg[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) gelu1<HAS_C2>(ex[4 * h + c], g[c], dg[c]);
      cpk[2 * h] = pack2(g[0], g[1]), cpk[2 * h + 1] = pack2(g[2], g[3]);
      if constexpr (HAS_C2) dpk[2 * h] = pack2(dg[0], dg[1]), dpk[2 * h + 1] = pack2(dg[2], dg[3]);
    } else if constexpr (EK == PS_BIAS) {
      cpk[2 * h] = pack2(ex[4 * h], ex[4 * h + 1]), cpk[2 * h + 1] = pack2(ex[4 * h + 2], ex[4 * h + 3]);
    }
  };
  auto ep_store = [&](auto Sc) {
    constexpr int S = decltype(Sc)::value;
    if constexpr (HAS_AUX) {
      const uint4 a = lds_read16(lds_addr(auxb + (S & 1) * 1024 + lane * 16));
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int c = 0; c < 4; ++c)
        cpk[c] = pack2(ex[2 * c] * h16_lo(aw[c]), ex[2 * c + 1] * h16_hi(aw[c]));
    }
    if constexpr (DBG & 1) {
      asm volatile("" ::"v"(cpk[0]), "v"(cpk[1]), "v"(cpk[2]), "v"(cpk[3]));
      if constexpr (HAS_C2) asm volatile("" ::"v"(dpk[0]), "v"(dpk[1]), "v"(dpk[2]), "v"(dpk[3]));
      return;
    }
    st16(const_cast<char*>(slab_base(Sc, e.C, e.ldc)) + lane_c, cpk[0], cpk[1], cpk[2], cpk[3], e.nt);
    if constexpr (HAS_C2) st16(const_cast<char*>(slab_base(Sc, e.C2, e.ldc2)) + lane_c2, dpk[0], dpk[1], dpk[2], dpk[3], e.nt);
  };

  // ---- one K-tile.  EP: -1 = no epilogue piece; S >= 0: slab S of the previous sub-tile rides along.  q0 = slot of unit 3g.
  auto iter = [&](auto EPc, int q0) {
    constexpr int EP = decltype(EPc)::value;
    constexpr bool E = EP >= 0 && !(DBG & 2);
    const int qa = wrap(q0 + ua), qb = wrap(q0 + UA);
    const int q1 = wrap(q0 + UPT);
    const int qa1 = wrap(q1 + ua), qb1 = wrap(q1 + UA);
    const int p1 = q0 == 0 ? PS_NSLOT - 2 : (q0 == 1 ? PS_NSLOT - 1 : q0 - 2), p2 = q0 == 0 ? PS_NSLOT - 1 : q0 - 1;   // slots of K-tile g-1: units 1, 2
    auto kstep = [&](auto CURc, auto RSc, int sa, int sb, auto MEMc) {   // MFMAs on buffer CUR; reads of k-step RS from slots sa / sb
      constexpr int cur = decltype(CURc)::value, nxt = cur ^ 1, rs = decltype(RSc)::value;
      const char* pa = smem + sa * PS_UNIT + ra[rs];
      const char* pb = smem + sb * PS_UNIT + rb[rs];
      ps_for<4>([&](auto Mc) {
        constexpr int m = decltype(Mc)::value, i = m >> 1, j = m & 1;
        if constexpr (!(DBG & 4)) acc[i][j] = VFM_MFMA16(fa[cur][i], fb[cur][j], acc[i][j]);
        if constexpr (!(DBG & 16)) {
          if constexpr (m < 2) fa[nxt][m] = *reinterpret_cast<const bf16x8*>(pa + m * 4096);
          else fb[nxt][m - 2] = *reinterpret_cast<const bf16x8*>(pb + (m - 2) * 4096);
        }
        MEMc(Mc);
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    // k0: A rows 0-127 of K-tile g+2 -> slot p1 (unit 1 of K-tile g-1, freed at beta_{g-1})
    kstep(IC<0>{}, IC<1>{}, qa, qb, [&](auto Mc) {
      constexpr int m = decltype(Mc)::value;
      if constexpr (!BURST && m == 0) dmaA(IC<0>{}, IC<0>{}, p1);
      if constexpr (!BURST && m == 1) dmaA(IC<0>{}, IC<1>{}, p1);
      if constexpr (E && m == 2) ep_dump(IC<(E ? EP : 0)>{});
    });
    kstep(IC<1>{}, IC<2>{}, qa, qb, [&](auto Mc) {
      constexpr int m = decltype(Mc)::value;
      if constexpr (!BURST && m == 0) dmaA(IC<1>{}, IC<0>{}, p2);
      if constexpr (E && m == 1) ep_read(IC<(E ? EP : 0)>{});
      if constexpr (E && m == 3) ep_act(IC<0>{});
    });
    kstep(IC<0>{}, IC<3>{}, qa, qb, [&](auto Mc) {
      constexpr int m = decltype(Mc)::value;
      if constexpr (!BURST && m == 0) { dmaA(IC<1>{}, IC<1>{}, p2); advA(); }
      if constexpr (E && m == 2) ep_act(IC<1>{});
    });
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): every fragment of K-tile g is in registers
    // K-tile g+1 (its B unit went out in k3 of the previous iteration) and this slab's aux piece (issued just before that B unit)
    // must have landed.  Younger than those: the four A pieces above and, when the previous iteration was a peeled one, its NST
    // stores (issued behind the B pieces; every lane stores - the accepted shapes are tile-aligned - so the count is exact)
    wait_vmcnt<(DBG & 8) ? 0 : 4 + ((EP >= 1 && !(DBG & 3)) ? NST : 0)>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // k3: next slab's aux piece first (so that the wait above covers it one iteration later), then the B unit of K-tile g+2 -> slot q0,
    // then this slab's stores
    kstep(IC<1>{}, IC<0>{}, qa1, qb1, [&](auto Mc) {
      constexpr int m = decltype(Mc)::value;
      if constexpr (E && HAS_AUX && m == 0 && EP + 1 < PS_PEEL) aux_dma(IC<(E && EP + 1 < PS_PEEL ? EP + 1 : 0)>{});
      if constexpr (!BURST) {
        if constexpr (m == 1) dmaB(IC<0>{}, q0);
        if constexpr (m == 2) { dmaB(IC<1>{}, q0); advB(); }
      } else {
        if constexpr (m == 0) { dmaB(IC<0>{}, q0); dmaB(IC<1>{}, q0); advB(); }
        if constexpr (m == 1) { dmaA(IC<0>{}, IC<0>{}, wrap(q0 + 1)); dmaA(IC<0>{}, IC<1>{}, wrap(q0 + 1)); }
        if constexpr (m == 2) { dmaA(IC<1>{}, IC<0>{}, wrap(q0 + 2)); dmaA(IC<1>{}, IC<1>{}, wrap(q0 + 2)); advA(); }
      }
      if constexpr (E && m == 3) ep_store(IC<(E ? EP : 0)>{});
    });
  };

  // ---- prologue: bias piece, units 0 .. 5 (K-tiles 0 and 1) in flight, K-tile 0 landed, fragments of (0, s0); cursors at K-tile 2
  set_soffB(n00);
  if constexpr (HAS_BIAS) {   // 256 floats = the bias of this block's per * 128 <= 512 columns ... one piece covers 256 floats: per <= 2
    const long bn = (n00 + lane * 4) % e.bias_mod;
    glds16(sbase(e.bias, 0) + (unsigned)(bn * 4), auxb + 1024);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    dmaA(IC<0>{}, IC<0>{}, 3 * t), dmaA(IC<0>{}, IC<1>{}, 3 * t);
    dmaA(IC<1>{}, IC<0>{}, 3 * t + 1), dmaA(IC<1>{}, IC<1>{}, 3 * t + 1);
    dmaB(IC<0>{}, 3 * t + 2), dmaB(IC<1>{}, 3 * t + 2);
    advA();
    advB();
  }
  if constexpr (BURST) {   // units 6, 7 (A of K-tile 2) as well: the steady state has issued up to unit 3 g + 7 before iteration g
    dmaA(IC<0>{}, IC<0>{}, 6), dmaA(IC<0>{}, IC<1>{}, 6);
    dmaA(IC<1>{}, IC<0>{}, 7), dmaA(IC<1>{}, IC<1>{}, 7);
    advA();
    wait_vmcnt<10>();
  } else {
    wait_vmcnt<6>();   // K-tile 0 (and the bias piece, older still) landed; K-tile 1's six pieces may fly
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < MI; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(smem + ua * PS_UNIT + ra[0] + i * 4096);
#pragma unroll
  for (int j = 0; j < NI; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(smem + UA * PS_UNIT + rb[0] + j * 4096);
  __builtin_amdgcn_sched_barrier(0);

  int q0 = 0;
  for (int sub = 0; sub < per; ++sub) {
    int v = 0;
    if (sub > 0) {
      if constexpr (HAS_AUX) {   // slab 0's aux piece: issued here, needed behind beta of the first peeled iteration.  Younger at that
        aux_dma(IC<0>{});        // wait: X1, X2 of that iteration = 4 pieces: the count of iter() holds
      }
      ps_for<PS_PEEL>([&](auto Vc) {
        iter(Vc, q0);
        q0 = wrap(q0 + UPT);
      });
      v = PS_PEEL;
    }
    for (; v < nk; ++v) {
      iter(IC<-1>{}, q0);
      q0 = wrap(q0 + UPT);
    }
    // the finished sub-tile moves to the second accumulator set
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        accp[i][j] = acc[i][j];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      }
    pn0 = n00 + (long)sub * BN;
  }

  // ---- the last sub-tile's epilogue, nothing left to hide it under
  wait_vmcnt<0>();
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (DBG & 2) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(accp[i][j]));
    return;
  }
  ps_for<PS_PEEL>([&](auto Sc) {
    if constexpr (HAS_AUX) {
      aux_dma(Sc);
      wait_vmcnt<0>();
    }
    ep_dump(Sc);
    ep_read(Sc);
    ep_act(IC<0>{});
    ep_act(IC<1>{});
    ep_store(Sc);
  });
}

template <int EK, int DBG = 0, bool BURST = false>
bool launch_ps_t(const vfm_gemm_desc* d, hipStream_t s, const vfm_gemm_desc* tail, int per) {
  const int tiles_m = (int)(d->M / 256), sup_n = (int)(d->N / (128 * per));
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_gemm_ps<EK, DBG, BURST>, hipFuncAttributeMaxDynamicSharedMemorySize, PS_SMEM);
    attr = true;
  }
  SkinnyTail sk;
  sk.nblk = 0;
  if (tail) sk.A = (const bf16_t*)tail->A, sk.lda = tail->sa_m, sk.M = tail->M, sk.nblk = cdiv(tail->N, 32), sk.e = make_epi(tail);
  hipLaunchKernelGGL((k_gemm_ps<EK, DBG, BURST>), dim3(tiles_m * sup_n + sk.nblk), dim3(512), PS_SMEM, s, (const bf16_t*)d->A, d->sa_m, (const bf16_t*)d->B,
                     d->sb_n, d->M, d->N, d->K, tiles_m, sup_n, per, make_epi(d), sk);
  return true;
}

}  // namespace

// Which epilogue instance serves this descriptor, or -1 (the caller then keeps the tile kernels).
static int ps_kind(const vfm_gemm_desc* d) {
  auto a16 = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
  if (d->c_dt != VFM_BF16 || d->residual || d->colscale || !a16(d->C) || d->ldc % 8) return -1;
  const long bm = d->bias_mod > 0 ? d->bias_mod : d->N;
  if (d->bias && (!a16(d->bias) || bm % 4 || bm < d->N)) return -1;
  switch (d->ep_mode) {
    case VFM_EP_NONE: return (d->bias && !d->C2) ? PS_BIAS : -1;
    case VFM_EP_GELU: return (d->bias && !d->C2) ? PS_GELU : -1;
    case VFM_EP_GELU_DGELU: return (d->bias && d->C2 && d->c2_dt == VFM_BF16 && a16(d->C2) && d->ldc2 % 8 == 0) ? PS_GELU_DGELU : -1;
    case VFM_EP_MUL: return (!d->bias && !d->C2 && d->aux && d->aux_dt == VFM_BF16 && a16(d->aux) && d->ld_aux % 8 == 0) ? PS_MUL : -1;
    default: return -1;
  }
}

// per: sub-tiles of 256 x 128 per block (2: a 256 x 256 region).  Shapes: M % 256 == 0, N % (128 per) == 0, K % 64 == 0, K >= 64 * 8 + 64
// (the previous sub-tile's epilogue rides on the first eight K-tiles), operands contiguous in K, spans < 4 GiB, no batch.
bool vfm_gemm_ps_ok(const vfm_gemm_desc* d, int per) {
  if (ps_kind(d) < 0 || d->batch > 1 || per < 1 || per > 2) return false;
  if (d->sa_k != 1 || d->sb_k != 1 || d->M % 256 || d->N % (128 * per) || d->K % 64 || d->K < 64 * (PS_PEEL + 1)) return false;
  if ((d->M + 256) * d->sa_m >= (1l << 31) || (d->N + 256) * d->sb_n >= (1l << 31)) return false;
  if ((d->M + 8) * d->ldc >= (1l << 31) * 2 || (d->aux && (d->M + 8) * d->ld_aux >= (1l << 31))) return false;
  return ((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0 && d->sa_m % 8 == 0 && d->sb_n % 8 == 0;
}

extern int g_pp_dbg;
int g_ps_burst = 0;   // vfm_tune("ps_burst")
bool vfm_gemm_launch_ps(const vfm_gemm_desc* d, hipStream_t s, const vfm_gemm_desc* tail, int per) {
  if (g_pp_dbg && ps_kind(d) == PS_GELU_DGELU) {   // timing diagnostics on the fc1 forward instance
    switch (g_pp_dbg) {
      case 1: return launch_ps_t<PS_GELU_DGELU, 1>(d, s, tail, per);
      case 2: return launch_ps_t<PS_GELU_DGELU, 2>(d, s, tail, per);
      case 6: return launch_ps_t<PS_GELU_DGELU, 6>(d, s, tail, per);
      case 10: return launch_ps_t<PS_GELU_DGELU, 10>(d, s, tail, per);
      case 26: return launch_ps_t<PS_GELU_DGELU, 26>(d, s, tail, per);
      default: break;
    }
  }
  if (g_ps_burst) {
    switch (ps_kind(d)) {
      case PS_BIAS: return launch_ps_t<PS_BIAS, 0, true>(d, s, tail, per);
      case PS_GELU_DGELU: return launch_ps_t<PS_GELU_DGELU, 0, true>(d, s, tail, per);
      case PS_MUL: return launch_ps_t<PS_MUL, 0, true>(d, s, tail, per);
      case PS_GELU: return launch_ps_t<PS_GELU, 0, true>(d, s, tail, per);
      default: return false;
    }
  }
  switch (ps_kind(d)) {
    case PS_BIAS: return launch_ps_t<PS_BIAS>(d, s, tail, per);
    case PS_GELU_DGELU: return launch_ps_t<PS_GELU_DGELU>(d, s, tail, per);
    case PS_MUL: return launch_ps_t<PS_MUL>(d, s, tail, per);
    case PS_GELU: return launch_ps_t<PS_GELU>(d, s, tail, per);
    default: return false;
  }
}
