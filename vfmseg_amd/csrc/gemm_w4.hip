// Large-wave-tile bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(alpha * A[M,K] * B[N,K]^T), 256 x 256 tiles, 4 waves.
//
// Why another tile kernel: the LDS pipe moves 128 B/clk/CU and one ds_read_b128 of a wave is 1 KiB = 8 clk, while one
// 32x32x16 MFMA keeps a SIMD's matrix pipe busy for 32 clk.  A wave tile of (MI x NI) 32x32 blocks reads MI + NI fragments
// for MI * NI MFMAs per 16-wide k-step, so with all four SIMDs busy the LDS pipe is loaded to
//     4 SIMDs * (MI + NI) * 8 clk / (MI * NI * 32 clk)   (+ the LDS-DMA writes of the operands themselves, 1/8 .. 1/4)
// = 150 % for the 64x32 wave tiles of the 128x128 kernel, 75 % for 128x64 (ping-pong kernel) and 50 % for the 128x128
// wave tile used here: one wave per SIMD, 16 MFMAs per k-step, the 256-register accumulator in AGPRs.
//
// Operand staging: K-tiles of 64, i.e. 128-byte rows, so that every LDS-DMA request is a whole cache line.  (Measured: the
// same bytes requested as 64-B half lines - 32-wide K-tiles - stream at 54-58 GB/s per CU from L2, as whole lines at
// ~100 GB/s; a 256 x 256 tile needs 32 B per matrix-pipe clock, which the half-line form cannot deliver.)
// LDS: ring of five 32-KiB chunk slots (160 KiB); chunk c = 2u -> B rows of K-tile u, c = 2u+1 -> A rows, at slot c % 5.
// A chunk is 256 rows x 128 B; 16-B piece p of row r is stored at p ^ ((r>>1)&7) (applied to the DMA SOURCE address and to
// the fragment reads).  Wave w moves pieces 8w .. 8w+7 (8 rows each) of a chunk: 8 global_load_lds_dwordx4.
//
// Schedule (iteration v = K-tile v = four k-steps of 16 MFMAs; fragments double-buffered per k-step; chunks are issued in
// chunk order, four DMA instructions per k-step):
//     step 0: MFMAs (v,s0) | reads (v,s1)   | DMA chunk 2v+3 [4..7]
//     step 1: MFMAs (v,s1) | reads (v,s2)   | DMA chunk 2v+4 [0..3]
//     step 2: MFMAs (v,s2) | reads (v,s3)   | DMA chunk 2v+4 [4..7]
//     beta_v: lgkmcnt(0) (all of K-tile v has been read), vmcnt(8) (chunks <= 2v+3 = K-tile v+1 landed), s_barrier
//     step 3: MFMAs (v,s3) | reads (v+1,s0) | DMA chunk 2v+5 [0..3]
// RAW: K-tile v+1 is first read in step 3, behind beta_v.  WAR: chunks 2v+5 / 2v+6 replace chunks 2v / 2v+1 (K-tile v) and are
// issued after beta_v.  Every memory instruction sits between two MFMAs: the matrix pipe never waits for an issue.
#include "gemm_dev.h"

// The kernel is generic in the wave tile (MI x NI blocks of 32 x 32) and the wave grid (WM_W x WN_W); the block tile is square
// (T = WM_W * MI * 32 = WN_W * NI * 32 rows and columns) so that A and B chunks have one size (T rows x 128 B):
//     <4, 4, 2, 2>  256 x 256, 4 waves, 128 x 128 wave tiles, AGPR accumulator, one wave per SIMD             (config 32)
//     <4, 2, 2, 4>  256 x 256, 8 waves, 128 x 64 wave tiles, two waves per SIMD                               (config 33)
//     <2, 1, 2, 4>  128 x 128, 8 waves, 64 x 32 wave tiles, 80-KiB ring: two blocks per CU (the layout of the 128 x 128 kernel
//                   of gemm_bf16.hip with the 2.5-K-tile chunk ring instead of two whole-K-tile stages)       (config 34)
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(IC<N - 1>{});
  }
}

// P = prefetch depth: the ring has NCH = 2P + 1 chunk slots (P = 2: the five-chunk ring described above; chunk 2v+2P-1 .. 2v+2P+1
// are issued during iteration v, and the wait before the last k-step leaves 2P-3 whole chunks in flight).  P = 4 (nine slots, 144 KiB
// for 128 x 128 tiles) is for launches with at most one tile per CU, where the LDS of the second block would lie idle.
// CLSIN (8-wave 128 x 128 forms): the tail rows of M (<= 32: the [cls] rows of the token-major layout) are not extra blocks of the
// launch but ride inside ONE regular block per tile column: that block's waves issue one more MFMA per k-step - the tail rows'
// A fragment (straight from global memory, a K-tile ahead) against the B fragment they hold anyway.  Tail blocks cannot share
// a CU with a deep ring (they would reserve its whole LDS and wait for a free CU: measured 46 -> 51 us); this way the
// one-tile-per-CU shapes (N = 1024: 96 launches per train step) get the seven-chunk ring.  The 8 blocks concerned do 3 MFMAs
// per k-step instead of 2, inside a loop that is bound by the operand stream, not by the matrix pipe.
// SPLITK: every tile is computed by TWO blocks, one per half of K (adjacent ids after the XCD map: same XCD, same L2).  The block that
// arrives first at the tile's counter leaves its accumulators in the workspace (fp32, register order: 16 bytes per lane and store) and
// raises the tile's flag; the second one adds them to its own and runs the epilogue.  own + partner is the same fp32 sum whichever half
// finishes last, so the result does not depend on the order.  For launches of at most one tile per CU over a long K (the N = 1024
// GEMMs of the coarse prediction pass: 128 tiles): twice the blocks fill the idle CUs / the second block slot of every CU.
struct SplitK {
  float* ws;        // [tile][threads * MI * NI * 16] partial accumulators of the first finisher
  unsigned* cnt;    // [tile] arrivals (zero between launches)
  unsigned* flag;   // [tile] partial published (zero between launches)
};

template <bool VEC, int MI, int NI, int WM_W, int WN_W, int P = 2, int DBG = 0, bool CLSIN = false, bool SPLITK = false>
__global__ void __launch_bounds__(WM_W* WN_W * 64, (WM_W * WN_W == 4 && MI == 2 && NI == 2) ? 2 : 1)   // (the 4-wave 128 x 128 form: two blocks per CU)
    k_gemm_w4(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb, long M, long N, long K, long stride_a,
              long stride_b, long stride_c, int tiles_m, int tiles_n, EpiParams e, SkinnyTail sk, SplitK spk) {
  // block tile TM x TN (rows of A x rows of B); square in every form but the 192 x 256 one (<6, 2, 1, 4>: M = 9216 x N = 1280 / 1024
  // GEMMs of the 1024^2 predictions, where 256 x 256 tiles fill 180 / 144 of the 256 CUs and 128 x 128 tiles take two rounds)
  constexpr int TM = WM_W * MI * 32, TN = WN_W * NI * 32, T = TM > TN ? TM : TN;
  // (MI odd - the 8-wave 192 x 256 form <3, 2, 2, 4> - : the epilogue takes the last 32-row block on its own, and a wave's pieces of
  // an A chunk split 2 + 1 over the two k-steps that issue them)
  static_assert(TM == TN || !CLSIN, "the tail-row-in-block form is written for square tiles");
  static_assert(!SPLITK || !CLSIN, "the split-K form takes its tail rows as skinny blocks");
  constexpr int KS = SPLITK ? 2 : 1;
  constexpr int WAVES = WM_W * WN_W, NCH = 2 * P + 1;
  constexpr int CH = T * 128;                        // bytes per ring slot: the larger chunk (rows x one 128-B K-tile row)
  constexpr int PPC_A = TM / 8 / WAVES, PPC_B = TN / 8 / WAVES;   // DMA pieces per wave and chunk
  constexpr int PH_A[2] = {(PPC_A + 1) / 2, PPC_A / 2}, PH_B[2] = {(PPC_B + 1) / 2, PPC_B / 2};   // ... in the first / second half (k-step) of a chunk
  constexpr int PPC = PPC_B, PPS = PH_B[0], PPCX = PPC_A > PPC_B ? PPC_A : PPC_B;
  static_assert(PPC_A >= 2 && PPC_B >= 2 && (TM % (8 * WAVES)) == 0 && (TN % (8 * WAVES)) == 0, "every wave moves pieces of both halves of a chunk");
  static_assert(!CLSIN || (PPC_B % 2 == 0), "CLSIN counts equal halves");
  constexpr int NMF = MI * NI, NRD = MI + NI;  // per k-step: MFMAs, fragment reads (+ the DMA pieces of the chunk half issued in it)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (NCH * CH >= 65536 && !CLSIN) {  // the tail rows of M run as extra blocks at the end of the grid
    if (sk.nblk > 0 && (int)blockIdx.x >= tiles_m * tiles_n * KS) {
      skinny_tile<WAVES>(sk.A, sk.lda, B, ldb, sk.M, N, K, (long)((int)blockIdx.x - tiles_m * tiles_n * KS) * 32, sk.e, 0, smem);
      return;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN_W, wn = wave % WN_W;

  // ---- XCD-aware tile mapping (same as k_gemm_bf16)
  const int ntiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int nb = ntiles * KS;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ksplit = SPLITK ? (bid & 1) : 0;   // which half of K
  if constexpr (SPLITK) bid >>= 1;
  const int tile_id = bid;
  constexpr int GM = 8;
  const int group = bid / (GM * tiles_n);
  const int first_m = group * GM;
  const int gsz = min(tiles_m - first_m, GM);
  const int tm = first_m + (bid % (GM * tiles_n)) % gsz;
  const int tn = (bid % (GM * tiles_n)) / gsz;
  const long m0 = (long)tm * TM, n0 = (long)tn * TN;
  const long z = blockIdx.y;
  if constexpr (SPLITK) K >>= 1;   // (the dispatcher checks K % 128 == 0 and K / 2 >= 64 P)
  const bf16_t* Ab = A + z * stride_a + (long)ksplit * K;
  const bf16_t* Bb = B + z * stride_b + (long)ksplit * K;

  // ---- per-lane DMA sources (32-bit byte offsets from the scalar matrix base; the dispatcher checks the span < 4 GiB)
  unsigned soff[2][PPCX];  // [0: B, 1: A][piece]
#pragma unroll
  for (int j = 0; j < PPC_B; ++j) {
    const int r = (wave * PPC_B + j) * 8 + (lane >> 3);
    const int sw = ((lane & 7) ^ ((r >> 1) & 7)) << 3;
    long gn = n0 + r;
    if (gn > N - 1) gn = N - 1;
    soff[0][j] = (unsigned)((gn * ldb + sw) * 2);
  }
#pragma unroll
  for (int j = 0; j < PPC_A; ++j) {
    const int r = (wave * PPC_A + j) * 8 + (lane >> 3);
    const int sw = ((lane & 7) ^ ((r >> 1) & 7)) << 3;
    long gm = m0 + r;
    if (gm > M - 1) gm = M - 1;
    soff[1][j] = (unsigned)((gm * lda + sw) * 2);
  }
  auto dma = [&](auto Cc, auto Jc, int u, int pos) {  // piece J of chunk type C (0 B, 1 A) of K-tile u -> chunk slot pos
    constexpr int c = decltype(Cc)::value, j = decltype(Jc)::value;
    char* dst = smem + pos * CH + (wave * (c ? PPC_A : PPC_B) + j) * 1024;
    const unsigned long long bv = (unsigned long long)(c ? Ab : Bb) + (unsigned long long)u * 128;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bv), hi = __builtin_amdgcn_readfirstlane((unsigned)(bv >> 32));
    const char* base = (const char*)(((unsigned long long)hi << 32) | lo);  // pinned to SGPRs
    glds16(base + soff[c][j], dst);
  };
  auto dma_half = [&](auto Cc, auto Hc, int u, int pos) {  // half H of this wave's pieces of a chunk
    constexpr int c = decltype(Cc)::value, hh = decltype(Hc)::value, pps = c ? PH_A[hh] : PH_B[hh], j0 = hh * (c ? PH_A[0] : PH_B[0]);
    static_for<pps>([&](auto Jc) { dma(Cc, IC<j0 + decltype(Jc)::value>{}, u, pos); });
  };

  f32x16 acc[MI][NI];  // [i][j]: rows wm*MI*32 + i*32, columns wn*NI*32 + j*32
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  // fragment read offsets inside a chunk (bytes): row * 128 + swizzled 16-B piece; [s] = k-step inside the K-tile
  int ra[4], rb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int kx = ((2 * s + fh) ^ ((fr >> 1) & 7)) << 4;
    ra[s] = (wm * MI * 32 + fr) * 128 + kx;
    rb[s] = (wn * NI * 32 + fr) * 128 + kx;
  }
  const int nk = (int)(K / 64);  // >= P (checked by the dispatcher)
  bf16x8 fa[2][MI], fb[2][NI];   // [buffer][block]

  // ---- tail rows inside the block (CLSIN): this block owns them for its 128 columns when tm == tn % tiles_m.
  // Their A fragments go through LDS like everything else (a ring of eight 1-KiB slots behind the chunk ring: eight rows x one
  // 128-B K-tile row, same XOR swizzle), fetched by LDS-DMA four K-tiles ahead.  (Plain global loads into registers made the
  // compiler put an s_waitcnt vmcnt(0) at the loop head - its counter state is unknown across the back edge -, which drains the
  // whole chunk ring every K-tile: 49 -> 75 us.)  Every wave issues the same piece (identical bytes to the same slot), so that
  // all waves keep the same vmcnt arithmetic.
  constexpr int CLS_SLOTS = 8, CLS_D = 4;
  bool cls_blk = false;
  f32x16 acc_cls;
  bf16x8 fc[2];            // tail-row fragments, double-buffered per k-step like fa / fb
  unsigned cls_soff = 0;   // per-lane source offset inside a K-tile row block
  int rc[4] = {0, 0, 0, 0};
  char* cls_base = smem + NCH * CH;
  if constexpr (CLSIN) {
    static_assert(MI == 2 && NI == 1 && WAVES == 8, "CLSIN is written for the 64 x 32 wave tiles of the 8-wave 128 x 128 form");
    cls_blk = sk.nblk > 0 && tm == tn % tiles_m;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_cls[r] = 0.f;
    {
      const int r = lane >> 3;                       // staged row 0..7 (rows beyond the tail repeat its last row)
      long am = r;
      if (am > sk.M - 1) am = sk.M - 1;
      cls_soff = (unsigned)((am * sk.lda + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
    }
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const int rr = fr & 7;                         // accumulator rows 8.. are never stored: any staged row will do
      rc[s2] = rr * 128 + (((2 * s2 + fh) ^ ((rr >> 1) & 7)) << 4);
    }
  }
  auto main_loop = [&](auto CLSc) __attribute__((always_inline)) {
  constexpr bool CLS = decltype(CLSc)::value;
  auto cls_dma = [&](int u) __attribute__((always_inline)) {   // K-tile u (clamped to the last one) -> slot u % CLS_SLOTS
    const int nkt = (int)(K / 64);
    const int uu = u < nkt ? u : nkt - 1;
    const unsigned long long bv = (unsigned long long)sk.A + (unsigned long long)uu * 128;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bv), hi = __builtin_amdgcn_readfirstlane((unsigned)(bv >> 32));
    const char* base = (const char*)(((unsigned long long)hi << 32) | lo);
    glds16(base + cls_soff, cls_base + (u % CLS_SLOTS) * 1024);
  };
  // one k-step: NMF MFMAs on buffer CUR; spread between them the fragment reads of the next k-step (READ: k-step rs of the
  // chunks at slots pb / pa) and PPS DMA pieces (ISSUE: half H of chunk type CH of K-tile u into slot dpos)
  auto kstep = [&](auto CURc, auto READc, auto ISSUEc, auto CHc, auto Hc, int pb_slot, int pa_slot, auto RSc, int u, int dpos, int cls_off = 0) {
    constexpr int cur = decltype(CURc)::value, nxt = cur ^ 1, ch = decltype(CHc)::value, hh = decltype(Hc)::value;
    constexpr int pps = ch ? PH_A[hh] : PH_B[hh], j0 = hh * (ch ? PH_A[0] : PH_B[0]);
    constexpr int rs = decltype(RSc)::value, NOPS = NRD + pps;   // memory ops of this k-step
    constexpr bool READ = decltype(READc)::value, ISSUE = decltype(ISSUEc)::value;
    const char* pa = smem + pa_slot * CH + ra[rs];
    const char* pb = smem + pb_slot * CH + rb[rs];
    auto memop = [&](auto Kc) {  // memory op k of the step: MI A reads, NI B reads, PPS DMA pieces
      constexpr int k = decltype(Kc)::value;
      if constexpr (k < MI) {
        if constexpr (READ) fa[nxt][k] = *reinterpret_cast<const bf16x8*>(pa + k * 4096);
      } else if constexpr (k < NRD) {
        if constexpr (READ) fb[nxt][k - MI] = *reinterpret_cast<const bf16x8*>(pb + (k - MI) * 4096);
      } else {
        if constexpr (ISSUE && !(DBG & 2)) dma(IC<ch>{}, IC<j0 + k - NRD>{}, u, dpos);
      }
    };
    static_for<NMF>([&](auto Mc) {
      constexpr int m = decltype(Mc)::value, i4 = m / NI, j = m % NI;
      acc[i4][j] = VFM_MFMA16(fa[cur][i4], fb[cur][j], acc[i4][j]);
      // memory op k goes behind MFMA floor(k * NMF / NOPS): one per gap where there are more MFMAs than ops
      static_for<NOPS>([&](auto Kc) {
        if constexpr (decltype(Kc)::value * NMF / NOPS == m) memop(Kc);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (CLS) {  // the tail rows' MFMA of this k-step against the wave's own B fragment; next k-step's tail fragment
      acc_cls = VFM_MFMA16(fc[cur], fb[cur][0], acc_cls);
      if constexpr (READ) fc[nxt] = *reinterpret_cast<const bf16x8*>(cls_base + cls_off + rc[rs]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto wrap = [](int p) { return p >= NCH ? p - NCH : p; };
  // X: 0 steady (v <= nk-P-1); X = j+1 for the last P K-tiles v = nk-P+j (j = 0: only the second half of the last chunk is left
  // to issue; afterwards nothing; X = P is the last K-tile: no barrier, no fragments to prefetch)
  auto iter = [&](auto Xc, int v, int q) {  // q = slot of chunk 2v
    constexpr int X = decltype(Xc)::value;
    constexpr bool LAST = X == P;
    const int q1 = wrap(q + 1), q2 = wrap(q + 2), q3 = wrap(q + 3);
    const int d0 = wrap(q + 2 * P - 1), d1 = wrap(q + 2 * P);
    if constexpr (CLS && !LAST) cls_dma(v + CLS_D);   // (one piece, BEFORE this iteration's chunk pieces: see the vmcnt window below)
    const int co = CLS ? (v % CLS_SLOTS) * 1024 : 0, co1 = CLS ? ((v + 1) % CLS_SLOTS) * 1024 : 0;
    kstep(IC<0>{}, IC<true>{}, IC<(X <= 1)>{}, IC<1>{}, IC<1>{}, q, q1, IC<1>{}, v + P - 1, d0, co);
    kstep(IC<1>{}, IC<true>{}, IC<(X == 0)>{}, IC<0>{}, IC<0>{}, q, q1, IC<2>{}, v + P, d1, co);
    kstep(IC<0>{}, IC<true>{}, IC<(X == 0)>{}, IC<0>{}, IC<1>{}, q, q1, IC<3>{}, v + P, d1, co);
    if constexpr (!LAST) {
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
      constexpr int live = X == 0 ? 2 * P - 3 : (2 * P - 4 - 2 * (X - 1) > 0 ? 2 * P - 4 - 2 * (X - 1) : 0);  // whole chunks in flight
      // ... X = 0: chunks 2v+4 .. 2v+2P (P-1 of B, P-2 of A); afterwards whole (B, A) pairs up to the last chunk
      constexpr int live_pieces = X == 0 ? (P - 1) * PPC_B + (P - 2) * PPC_A : (live / 2) * (PPC_A + PPC_B);
      // CLS: the tail-row piece of this iteration sits between the chunk pieces; when the window of pieces that may stay in
      // flight reaches back past it, it is part of the window (else it is older than the window and completes with it)
      constexpr int after = X == 0 ? 3 * PPS : (X == 1 ? PPS : 0);   // chunk pieces issued behind the tail-row piece in this iteration
      constexpr int extra = (CLS && live * PPC > after) ? 1 : 0;
      wait_vmcnt<live_pieces + extra>();
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    kstep(IC<1>{}, IC<!LAST>{}, IC<(X == 0)>{}, IC<1>{}, IC<0>{}, q2, q3, IC<0>{}, v + P, q, co1);
  };

  // ---- prologue: chunks 0 .. 2P-2 and the first half of chunk 2P-1 in flight; K-tile 0 landed; fragments of (0, s0)
  if constexpr (CLS) {   // tail-row pieces of K-tiles 0 .. CLS_D-1 (older than every chunk piece: landed with K-tile 0)
#pragma unroll
    for (int u0 = 0; u0 < CLS_D; ++u0) cls_dma(u0);
  }
  static_for<2 * P - 1>([&](auto Cc) {
    constexpr int c = decltype(Cc)::value;
    dma_half(IC<(c & 1)>{}, IC<0>{}, c >> 1, c), dma_half(IC<(c & 1)>{}, IC<1>{}, c >> 1, c);
  });
  dma_half(IC<1>{}, IC<0>{}, P - 1, 2 * P - 1);
  wait_vmcnt<(P - 1) * PPC_B + (P - 2) * PPC_A + PH_A[0]>();   // K-tile 0 (chunks 0, 1) landed; chunks 2 .. 2P-2 and half of 2P-1 (A) in flight
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < MI; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(smem + CH + ra[0] + i * 4096);
#pragma unroll
  for (int j = 0; j < NI; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(smem + rb[0] + j * 4096);
  if constexpr (CLS) fc[0] = *reinterpret_cast<const bf16x8*>(cls_base + rc[0]);
  __builtin_amdgcn_sched_barrier(0);

  int v = 0, q = 0;
  for (; v < nk - P; ++v) {
    iter(IC<0>{}, v, q);
    q = wrap(q + 2);
  }
  static_for<P>([&](auto Jc) {
    constexpr int j = decltype(Jc)::value;
    iter(IC<j + 1>{}, v + j, q);
    q = wrap(q + 2);
  });
  __builtin_amdgcn_sched_barrier(0);
  };  // main_loop
  if constexpr (CLSIN) {
    if (cls_blk) main_loop(IC<true>{});   // (two copies of the loop: a branch inside it would break the hand-placed schedule)
    else main_loop(IC<false>{});
  } else {
    main_loop(IC<false>{});
  }

  if constexpr (SPLITK) {
    // The partial sums and the flag travel as relaxed agent-scope atomics (8-byte global stores / loads with sc1: coherent across the
    // XCDs' L2s by themselves), ordered by vmcnt(0) + the block barrier.  A release / acquire FENCE at agent scope would write back and
    // invalidate the whole L2 of the XCD per block (buffer_wbl2 / buffer_inv): measured 0.6 us per tile, serialised.
    constexpr int NT = WAVES * 64, NV = MI * NI * 8;   // 8-byte words per lane
    unsigned long long* slab = reinterpret_cast<unsigned long long*>(spk.ws) + (long)tile_id * NV * NT;
    __syncthreads();   // (the ring is no longer read: smem[0] carries the role of this block)
    if (tid == 0) *reinterpret_cast<unsigned*>(smem) = __hip_atomic_fetch_add(&spk.cnt[tile_id], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned arrived = *reinterpret_cast<volatile unsigned*>(smem);
    if (arrived == 0) {   // first: publish the partial sums, leave
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r2 = 0; r2 < 8; ++r2) {
            const unsigned long long w = (unsigned long long)__float_as_uint(acc[i][j][2 * r2]) | ((unsigned long long)__float_as_uint(acc[i][j][2 * r2 + 1]) << 32);
            __hip_atomic_store(&slab[((i * NI + j) * 8 + r2) * NT + tid], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
      __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): this wave's stores have been acknowledged
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&spk.flag[tile_id], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    // second: wait for the partner's partial (it is resident and past its K loop: the wait is short; bounded all the same, so that
    // no wave can spin forever), add, reset the tile's words for the next launch
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(&spk.flag[tile_id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
      __hip_atomic_store(&spk.flag[tile_id], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&spk.cnt[tile_id], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r2 = 0; r2 < 8; ++r2) {
          const unsigned long long w = __hip_atomic_load(&slab[((i * NI + j) * 8 + r2) * NT + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc[i][j][2 * r2] += __uint_as_float((unsigned)w);
          acc[i][j][2 * r2 + 1] += __uint_as_float((unsigned)(w >> 32));
        }
  }
  // ---- epilogue (accumulators -> per-wave fp32 LDS image -> 16-byte rows), one 64-row half of the wave tile at a time
  const long zoff = z * stride_c;
  if constexpr (DBG & 1) return;
  if constexpr (DBG & 4) {   // diagnostic: no epilogue, but the accumulators stay live (the MFMAs are not dead code)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)   // (a "v" constraint does not exist for the host pass: the instantiation would silently vanish)
        asm volatile("" ::"v"(acc[i][j]));
#endif
      }
    return;
  }
  __syncthreads();
  float* img = reinterpret_cast<float*>(smem) + wave * 64 * (NI * 32 + 4);
  const long mw = m0 + wm * MI * 32, nw = n0 + wn * NI * 32;
  static_for<MI / 2>([&](auto Hc) {
    constexpr int h = decltype(Hc)::value;
    f32x16(&a2)[2][NI] = *reinterpret_cast<f32x16(*)[2][NI]>(&acc[2 * h]);   // blocks 2h, 2h+1: one 64-row half
    if constexpr (VEC) epi_wave_tile<2, NI, 2>(e, zoff, a2, img, lane, mw + h * 64, nw, M, N);
    else epi_scalar<2, NI, 2>(e, zoff, a2, img, lane, mw + h * 64, nw, M, N);
  });
  if constexpr (MI % 2 == 1) {   // the last 32-row block of an odd wave tile
    f32x16(&a1)[1][NI] = *reinterpret_cast<f32x16(*)[1][NI]>(&acc[MI - 1]);
    if constexpr (VEC) epi_wave_tile<1, NI, 1>(e, zoff, a1, img, lane, mw + (MI - 1) * 32, nw, M, N);
    else epi_scalar<1, NI, 1>(e, zoff, a1, img, lane, mw + (MI - 1) * 32, nw, M, N);
  }
  if constexpr (CLSIN) {
    // tail rows: accumulator register r of lane (fr, fh) is row (r & 3) + 8 (r >> 2) + 4 fh, column nw + fr; all eight waves
    // computed them (no branch in the loop), the wm == 0 waves store
    if (cls_blk && wm == 0 && nw + fr < N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long row = (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (row < sk.M) epi_store(sk.e, 0, row, nw + fr, acc_cls[r]);
      }
    }
  }
}

extern int g_pp_dbg;
bool vfm_gemm_launch_w4(const vfm_gemm_desc* d, hipStream_t s, bool vec, const vfm_gemm_desc* tail, int form);
// split-K workspace of a stream: partial sums of up to SPLITK_TILES tiles of 128 x 128 + their counter / flag words (zeroed once: every
// launch leaves them zero).  One per stream, because launches on different streams may overlap.
constexpr int SPLITK_TILES = 512;
static SplitK splitk_ws(hipStream_t s) {
  struct Ent { hipStream_t s; SplitK w; };
  static thread_local Ent tab[8];
  static thread_local int n = 0;
  for (int i = 0; i < n; ++i)
    if (tab[i].s == s) return tab[i].w;
  SplitK w = {nullptr, nullptr, nullptr};
  if (n == 8) return w;   // (more streams than this process ever uses: the caller falls back to the unsplit form)
  void* p = nullptr;
  const size_t slab = (size_t)SPLITK_TILES * 128 * 128 * 4, words = (size_t)SPLITK_TILES * 2 * 4;
  if (hipMalloc(&p, slab + words) != hipSuccess) return w;
  (void)hipMemsetAsync((char*)p + slab, 0, words, s);
  w.ws = (float*)p, w.cnt = (unsigned*)((char*)p + slab), w.flag = w.cnt + SPLITK_TILES;
  tab[n].s = s, tab[n].w = w, ++n;
  return w;
}

template <bool VEC, int MI, int NI, int WM_W, int WN_W, int P, int DBG, bool CLSIN = false, bool SPLITK = false>
static bool launch_w4_t(const vfm_gemm_desc* d, hipStream_t s, const vfm_gemm_desc* tail) {
  constexpr int TM = WM_W * MI * 32, TN = WN_W * NI * 32, T = TM > TN ? TM : TN, WAVES = WM_W * WN_W;
  constexpr int RING = (2 * P + 1) * T * 128 + (CLSIN ? 8 * 1024 : 0), EPI = WAVES * 64 * (NI * 32 + 4) * 4, SMEM = RING > EPI ? RING : EPI;
  const int tiles_m = cdiv(d->M, TM), tiles_n = cdiv(d->N, TN);
  const long batch = d->batch > 0 ? d->batch : 1;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_gemm_w4<VEC, MI, NI, WM_W, WN_W, P, DBG, CLSIN, SPLITK>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr = true;
  }
  SplitK spk = {nullptr, nullptr, nullptr};
  if constexpr (SPLITK) spk = splitk_ws(s);
  SkinnyTail sk;
  sk.nblk = 0;
  const bool fold = tail && batch == 1 && RING >= 65536 && (!CLSIN || (tail->M <= 8 && tail->sa_k == 1));
  if (fold) sk.A = (const bf16_t*)tail->A, sk.lda = tail->sa_m, sk.M = tail->M, sk.nblk = cdiv(tail->N, 32), sk.e = make_epi(tail);
  hipLaunchKernelGGL((k_gemm_w4<VEC, MI, NI, WM_W, WN_W, P, DBG, CLSIN, SPLITK>), dim3(tiles_m * tiles_n * (SPLITK ? 2 : 1) + (CLSIN ? 0 : sk.nblk), (unsigned)batch),
                     dim3(WAVES * 64), SMEM, s, (const bf16_t*)d->A, d->sa_m, (const bf16_t*)d->B, d->sb_n, d->M, d->N, d->K, d->stride_a, d->stride_b,
                     d->stride_c, tiles_m, tiles_n, make_epi(d), sk, spk);
  return fold || !tail;
}
// the split-K form of the 128 x 128 ring kernel (see SplitK): K % 128 == 0, K / 2 >= 128, one batch, at most SPLITK_TILES tiles, the vector epilogue
bool vfm_gemm_splitk_ok(const vfm_gemm_desc* d, bool vec) {
  return vec && d->batch <= 1 && d->K % 128 == 0 && d->K >= 512 && (long)cdiv(d->M, 128) * cdiv(d->N, 128) <= SPLITK_TILES;
}
bool vfm_gemm_launch_w4_splitk(const vfm_gemm_desc* d, hipStream_t s, const vfm_gemm_desc* tail) {
  if (!splitk_ws(s).ws) return vfm_gemm_launch_w4(d, s, true, tail, 2);
  return launch_w4_t<true, 2, 1, 2, 4, 2, 0, false, true>(d, s, tail);
}
// form: 4 = 256x256 / 4 waves, 8 = 256x256 / 8 waves, 6 = 192x256 / 4 waves, 7 = 192x256 / 8 waves, 2 = 128x128 / 8 waves (two blocks per CU), 3 / 5 = 128x128
// with the seven- / nine-chunk ring (one block per CU).  Returns whether the tail rows were folded into the launch (otherwise
// the caller runs the skinny kernel).
bool vfm_gemm_launch_w4(const vfm_gemm_desc* d, hipStream_t s, bool vec, const vfm_gemm_desc* tail, int form) {
  if (form == 4) {
    if (!vec) return launch_w4_t<false, 4, 4, 2, 2, 2, 0>(d, s, tail);
    return g_pp_dbg == 1 ? launch_w4_t<true, 4, 4, 2, 2, 2, 1>(d, s, tail) : launch_w4_t<true, 4, 4, 2, 2, 2, 0>(d, s, tail);
  }
  if (form == 9)   // 128 x 128, FOUR waves with 64 x 64 wave tiles (1.0 fragment reads per MFMA instead of the 8-wave form's 1.5), two blocks per CU
    return vec ? launch_w4_t<true, 2, 2, 2, 2, 2, 0>(d, s, tail) : launch_w4_t<false, 2, 2, 2, 2, 2, 0>(d, s, tail);
  if (form == 7) return vec ? launch_w4_t<true, 3, 2, 2, 4, 2, 0>(d, s, tail) : launch_w4_t<false, 3, 2, 2, 4, 2, 0>(d, s, tail);
  if (form == 6) return vec ? launch_w4_t<true, 6, 2, 1, 4, 2, 0>(d, s, tail) : launch_w4_t<false, 6, 2, 1, 4, 2, 0>(d, s, tail);
  if (form == 8) {
    if (!vec) return launch_w4_t<false, 4, 2, 2, 4, 2, 0>(d, s, tail);
    if (g_pp_dbg == 4) return launch_w4_t<true, 4, 2, 2, 4, 2, 4>(d, s, tail);
    return g_pp_dbg == 1 ? launch_w4_t<true, 4, 2, 2, 4, 2, 1>(d, s, tail) : launch_w4_t<true, 4, 2, 2, 4, 2, 0>(d, s, tail);
  }
  // deep rings (one block per CU): tail rows ride inside the regular blocks (CLSIN)
  const bool clsin = tail && (d->batch <= 1) && tail->M <= 8 && tail->sa_k == 1;
  if (form == 5) {
    if (clsin) return vec ? launch_w4_t<true, 2, 1, 2, 4, 4, 0, true>(d, s, tail) : launch_w4_t<false, 2, 1, 2, 4, 4, 0, true>(d, s, tail);
    return vec ? launch_w4_t<true, 2, 1, 2, 4, 4, 0>(d, s, tail) : launch_w4_t<false, 2, 1, 2, 4, 4, 0>(d, s, tail);
  }
  if (form == 3) {
    if (clsin) return vec ? launch_w4_t<true, 2, 1, 2, 4, 3, 0, true>(d, s, tail) : launch_w4_t<false, 2, 1, 2, 4, 3, 0, true>(d, s, tail);
    return vec ? launch_w4_t<true, 2, 1, 2, 4, 3, 0>(d, s, tail) : launch_w4_t<false, 2, 1, 2, 4, 3, 0>(d, s, tail);
  }
  if (!vec) return launch_w4_t<false, 2, 1, 2, 4, 2, 0>(d, s, tail);
  if (g_pp_dbg == 1) return launch_w4_t<true, 2, 1, 2, 4, 2, 1>(d, s, tail);   // diagnostic: no epilogue => the MFMAs are dead code too:
                                                                               // what is left is the operand stream (DMA + barriers)
  if (g_pp_dbg == 4) return launch_w4_t<true, 2, 1, 2, 4, 2, 4>(d, s, tail);   // diagnostic: main loop only (MFMAs live), no epilogue
  return launch_w4_t<true, 2, 1, 2, 4, 2, 0>(d, s, tail);
}
