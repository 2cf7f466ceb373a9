// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(alpha * A[M,K] * B[N,K]^T), fp32 accumulation.
//
//   tile BM x BN x 64, WAVES = WM_W x WN_W waves, each wave a (BM/WM_W) x (BN/WN_W) sub-tile of 32x32x16 MFMAs
//   operands staged by LDS-DMA (global_load_lds_dwordx4) into an NS-deep ring of LDS stages, prefetch distance NS-1,
//     ONE raw s_barrier per K-step, counted s_waitcnt vmcnt(N) so the prefetch stays in flight across the barrier
//   LDS image [rows][64] bf16 = 128-B rows; 16-B chunk c of row r lives at chunk c ^ ((r>>1)&7): the swizzle is applied
//     to the per-lane SOURCE address (the DMA destination is lane-linear) and again on the ds_read_b128 fragment reads
//   epilogue through LDS: accumulators -> row-major fp32 tile -> 16-byte vector loads/stores of bias / residual / aux / C
//   XCD-aware bijective tile mapping (blocks b, b+8 share an XCD's L2)
#include "gemm_dev.h"

template <int BM, int BN, int WM_W, int WN_W, int NS>
struct Cfg {
  static constexpr int WAVES = WM_W * WN_W;
  static constexpr int THREADS = WAVES * 64;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int A_PIECES = BM / 8 / WAVES, B_PIECES = BN / 8 / WAVES;  // 1-KiB LDS-DMA pieces per wave
  static constexpr int PIECES = A_PIECES + B_PIECES;
  static constexpr int WM = BM / WM_W, WN = BN / WN_W;
  static constexpr int MI = WM / 32, NI = WN / 32;
  static constexpr int EPI_LD = WN + 4;                       // fp32 row stride of the epilogue image
  static constexpr int EPI_BYTES = WAVES * 32 * EPI_LD * 4;   // one 32-row slab per wave at a time
  static constexpr int SMEM = (NS * STAGE_BYTES > EPI_BYTES) ? NS * STAGE_BYTES : EPI_BYTES;
  static_assert(BM % (8 * WAVES) == 0 && BN % (8 * WAVES) == 0, "tile rows must split into whole DMA pieces per wave");
  static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA");
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

// BT = true: B is given as [K, N] row-major (N contiguous) - weight-gradient GEMMs reduce over tokens, so the activation
// operand is consumed in place and its MFMA fragments come from transposed LDS reads (ds_read_b64_tr_b16); rows k >= kb_rows
// are clamped (the A operand is zero there).
// AT = true (with BT): A is given as [K, M] row-major too (M contiguous): C = A^T B with BOTH operands token-major, the form
// of a weight gradient dW = X^T dY - no transposed copy of either operand; k-rows past the valid tokens read a zero row.
__device__ __attribute__((aligned(16))) unsigned char g_zero_row[512];
template <int BM, int BN, int WM_W, int WN_W, int NS, bool VEC, bool BT, bool AT = false>
__global__ void __launch_bounds__(WM_W* WN_W * 64)
    k_gemm_bf16(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb, long M, long N, long K,
                long stride_a, long stride_b, long stride_c, int tiles_m, int tiles_n, long kb_rows, EpiParams e, SkinnyTail sk) {
  using C = Cfg<BM, BN, WM_W, WN_W, NS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (!BT && C::THREADS == 512 && C::SMEM >= 65536) {
    if (sk.nblk > 0 && (int)blockIdx.x >= tiles_m * tiles_n) {  // tail rows of M ([cls] tokens): extra blocks at the end of the grid
      skinny_tile(sk.A, sk.lda, B, ldb, sk.M, N, K, (long)((int)blockIdx.x - tiles_m * tiles_n) * 32, sk.e, 0, smem);
      return;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN_W, wn = wave % WN_W;

  // ---- XCD-aware tile mapping
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
  // BT: valid B rows of this batch slice (split-K batches slice the token dimension; the last slice may be short)
  long kb_valid = K;
  if constexpr (BT) {
    if (kb_rows > 0) {  // split-K slices of ONE [kb_rows, N] matrix; kb_rows == 0: independent batches, all K rows valid
      kb_valid = kb_rows - (ldb > 0 ? z * (stride_b / ldb) : 0);
      if (kb_valid > K) kb_valid = K;
      if (kb_valid < 1) kb_valid = 1;
    } else if (kb_rows < 0) {  // independent batches (one per layer) of -kb_rows valid rows each
      kb_valid = -kb_rows;
      if (kb_valid > K) kb_valid = K;
    }
  }

  // ---- per-lane DMA sources (row clamp keeps every load in bounds; clamped rows are never stored)
  const int lr = lane >> 3, pc = lane & 7;
  const bf16_t* a_src[C::A_PIECES];
  const bf16_t* b_src[C::B_PIECES];
  int at_row[C::A_PIECES];
#pragma unroll
  for (int j = 0; j < C::A_PIECES; ++j) {
    if constexpr (!AT) {
      const int r = (wave * C::A_PIECES + j) * 8 + lr;
      long gm = m0 + r;
      if (gm > M - 1) gm = M - 1;
      a_src[j] = Ab + gm * lda + ((pc ^ ((r >> 1) & 7)) << 3);
    } else {
      static_assert(!AT || (BT && BM == 128), "transposed-A path: BM = 128 (256-byte LDS rows), together with transposed B");
      const int piece = wave * C::A_PIECES + j;          // 1 KiB = 4 k-rows x 256 B
      const int r = piece * 4 + (lane >> 4);
      const int c = (lane & 15) ^ ((r & 3) << 2);
      long gm = m0 + c * 8;
      if (gm > M - 8) gm = M - 8;                        // M % 8 == 0: clamped chunks are never stored
      at_row[j] = r;
      a_src[j] = Ab + gm;
    }
  }
  int bt_row[C::B_PIECES];  // BT: k-row of this lane inside the tile, per piece
#pragma unroll
  for (int j = 0; j < C::B_PIECES; ++j) {
    if constexpr (!BT) {
      const int r = (wave * C::B_PIECES + j) * 8 + lr;
      long gn = n0 + r;
      if (gn > N - 1) gn = N - 1;
      b_src[j] = Bb + gn * ldb + ((pc ^ ((r >> 1) & 7)) << 3);
    } else {
      static_assert(!BT || BN == 128, "transposed-B path is built for BN = 128 (256-byte LDS rows)");
      const int piece = wave * C::B_PIECES + j;          // 1 KiB = 4 k-rows x 256 B
      const int r = piece * 4 + (lane >> 4);
      const int c = (lane & 15) ^ ((r & 3) << 2);
      long gn = n0 + c * 8;
      if (gn > N - 8) gn = N - 8;                        // N % 8 == 0: clamped chunks are never stored
      bt_row[j] = r;
      b_src[j] = Bb + gn;
    }
  }
  auto stage = [&](int slot, long k0) {
    char* sa = smem + slot * C::STAGE_BYTES;
    char* sb = sa + C::A_BYTES;
#pragma unroll
    for (int j = 0; j < C::A_PIECES; ++j) {
      if constexpr (!AT) {
        glds16(a_src[j] + k0, sa + (wave * C::A_PIECES + j) * 1024);
      } else {
        const long kr = k0 + at_row[j];
        const bf16_t* src = kr < kb_valid ? a_src[j] + kr * lda : reinterpret_cast<const bf16_t*>(g_zero_row) + (lane & 15) * 8;
        glds16(src, sa + (wave * C::A_PIECES + j) * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < C::B_PIECES; ++j) {
      if constexpr (!BT) {
        glds16(b_src[j] + k0, sb + (wave * C::B_PIECES + j) * 1024);
      } else {
        long kr = k0 + bt_row[j];
        if (kr > kb_valid - 1) kr = kb_valid - 1;
        glds16(b_src[j] + kr * ldb, sb + (wave * C::B_PIECES + j) * 1024);
      }
    }
  };

  f32x16 acc[C::MI][C::NI];
#pragma unroll
  for (int i = 0; i < C::MI; ++i)
#pragma unroll
    for (int j = 0; j < C::NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  int a_off[C::MI], a_sw[C::MI], b_off[C::NI], b_sw[C::NI];
#pragma unroll
  for (int i = 0; i < C::MI; ++i) {
    const int r = wm * C::WM + i * 32 + fr;
    a_off[i] = r * 128;
    a_sw[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < C::NI; ++j) {
    const int r = wn * C::WN + j * 32 + fr;
    b_off[j] = r * 128;
    b_sw[j] = (r >> 1) & 7;
  }

  const int nk = (int)(K / BK);
  // ---- prologue: fill NS-1 stages
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) stage(s, (long)s * BK);

  for (int t = 0; t < nk; ++t) {
    // tile t must have landed: groups issued so far cover tiles t .. min(t+NS-2, nk-1)
    const int inflight = min(NS - 2, nk - 1 - t);  // younger groups allowed to stay in flight
    if (inflight >= NS - 2 && NS >= 2) wait_vmcnt<(NS - 2) * C::PIECES>();
    else if (NS >= 4 && inflight == NS - 3) wait_vmcnt<(NS >= 3 ? (NS - 3) : 0) * C::PIECES>();
    else if (NS >= 5 && inflight == NS - 4) wait_vmcnt<(NS >= 4 ? (NS - 4) : 0) * C::PIECES>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // everyone's pieces of tile t are in LDS; everyone finished reading tile t-1
    if (t + NS - 1 < nk) stage((t + NS - 1) % NS, (long)(t + NS - 1) * BK);  // refills the slot tile t-1 used
    const char* sa = smem + (t % NS) * C::STAGE_BYTES;
    const char* sb = sa + C::A_BYTES;
    // fragment double buffering: the ds_reads of k-step s+1 are issued before the MFMAs of k-step s
    auto read_b = [&](int s2, int j) -> bf16x8 {
      if constexpr (!BT) {
        return *reinterpret_cast<const bf16x8*>(sb + b_off[j] + (((2 * s2 + fh) ^ b_sw[j]) << 4));
      } else {
        const int g = lane >> 4, i2 = lane & 15, q = i2 >> 2, p = i2 & 3, h2 = g >> 1;
        const int chunk = (wn * C::WN + j * 32) / 8 + 2 * (g & 1) + (p >> 1);
        const int r0 = 16 * s2 + 8 * h2 + q, r1 = r0 + 4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(sb + r0 * 256 + ((chunk ^ ((r0 & 3) << 2)) << 4) + ((p & 1) << 3)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(sb + r1 * 256 + ((chunk ^ ((r1 & 3) << 2)) << 4) + ((p & 1) << 3)));
        union { struct { s16x4 a, b; } st; bf16x8 v; } u;
        u.st.a = lo;
        u.st.b = hi;
        return u.v;
      }
    };
    auto read_a = [&](int s2, int i) -> bf16x8 {
      if constexpr (!AT) {
        return *reinterpret_cast<const bf16x8*>(sa + a_off[i] + (((2 * s2 + fh) ^ a_sw[i]) << 4));
      } else {
        const int g = lane >> 4, i2 = lane & 15, q = i2 >> 2, p = i2 & 3, h2 = g >> 1;
        const int chunk = (wm * C::WM + i * 32) / 8 + 2 * (g & 1) + (p >> 1);
        const int r0 = 16 * s2 + 8 * h2 + q, r1 = r0 + 4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(sa + r0 * 256 + ((chunk ^ ((r0 & 3) << 2)) << 4) + ((p & 1) << 3)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(sa + r1 * 256 + ((chunk ^ ((r1 & 3) << 2)) << 4) + ((p & 1) << 3)));
        union { struct { s16x4 a, b; } st; bf16x8 v; } u;
        u.st.a = lo;
        u.st.b = hi;
        return u.v;
      }
    };
    bf16x8 af[2][C::MI], bfr[2][C::NI];
#pragma unroll
    for (int i = 0; i < C::MI; ++i) af[0][i] = read_a(0, i);
#pragma unroll
    for (int j = 0; j < C::NI; ++j) bfr[0][j] = read_b(0, j);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s + 1 < BK / 16) {
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
          af[nxt][i] = read_a(s + 1, i);
#pragma unroll
        for (int j = 0; j < C::NI; ++j) bfr[nxt][j] = read_b(s + 1, j);
      }
#pragma unroll
      for (int i = 0; i < C::MI; ++i)
#pragma unroll
        for (int j = 0; j < C::NI; ++j)
          acc[i][j] = VFM_MFMA16(af[cur][i], bfr[cur][j], acc[i][j]);
    }
  }

  // ---- epilogue
  const long zoff = z * stride_c;
  __syncthreads();  // all LDS reads of the last stage are done; the ring is reused as the epilogue image
  float* img = reinterpret_cast<float*>(smem) + wave * 32 * C::EPI_LD;
  if constexpr (VEC) epi_wave_tile<C::MI, C::NI, 1>(e, zoff, acc, img, lane, m0 + wm * C::WM, n0 + wn * C::WN, M, N);
  else epi_scalar<C::MI, C::NI, 1>(e, zoff, acc, img, lane, m0 + wm * C::WM, n0 + wn * C::WN, M, N);
}

// tail != null: the <= 32 rows past the last full 128-row block run as extra blocks of the same launch when the
// configuration can host them (512 threads, >= 64 KiB LDS); returns false when the caller has to launch them separately
template <int BM, int BN, int WM_W, int WN_W, int NS, bool VEC, bool BT, bool AT = false>
static bool launch_one(const vfm_gemm_desc* d, hipStream_t s, const vfm_gemm_desc* tail = nullptr) {
  using C = Cfg<BM, BN, WM_W, WN_W, NS>;
  const int tiles_m = cdiv(d->M, BM), tiles_n = cdiv(d->N, BN);
  const long batch = d->batch > 0 ? d->batch : 1;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_gemm_bf16<BM, BN, WM_W, WN_W, NS, VEC, BT, AT>, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    attr = true;
  }
  SkinnyTail sk;
  sk.nblk = 0;
  const bool fold = tail && !BT && C::THREADS == 512 && C::SMEM >= 65536 && batch == 1;
  if (fold) {
    sk.A = (const bf16_t*)tail->A, sk.lda = tail->sa_m, sk.M = tail->M, sk.nblk = cdiv(tail->N, 32), sk.e = make_epi(tail);
  }
  dim3 grid(tiles_m * tiles_n + sk.nblk, (unsigned)batch), blk(C::THREADS);
  const long ldb = BT ? d->sb_k : d->sb_n;
  const long kb_rows = d->kb_rows;
  hipLaunchKernelGGL((k_gemm_bf16<BM, BN, WM_W, WN_W, NS, VEC, BT, AT>), grid, blk, C::SMEM, s, (const bf16_t*)d->A, AT ? d->sa_k : d->sa_m,
                     (const bf16_t*)d->B, ldb, d->M, d->N, d->K, d->stride_a, d->stride_b, d->stride_c, tiles_m, tiles_n, kb_rows,
                     make_epi(d), sk);
  return fold || !tail;
}

template <int BM, int BN, int WM_W, int WN_W, int NS>
static bool launch_cfg(const vfm_gemm_desc* d, hipStream_t s, bool vec, const vfm_gemm_desc* tail = nullptr) {
  if (vec) return launch_one<BM, BN, WM_W, WN_W, NS, true, false>(d, s, tail);
  return launch_one<BM, BN, WM_W, WN_W, NS, false, false>(d, s, tail);
}

template <int BM, int WM_W, int WN_W, int NS = 2>
static void launch_bt(const vfm_gemm_desc* d, hipStream_t s, bool vec) {
  if (vec) launch_one<BM, 128, WM_W, WN_W, NS, true, true>(d, s);
  else launch_one<BM, 128, WM_W, WN_W, NS, false, true>(d, s);
}

// ------------------------------------------------------------------------------------------------ skinny tail
__global__ void __launch_bounds__(512) k_gemm_bf16_skinny(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb,
                                                          long M, long N, long K, long stride_a, long stride_b, long stride_c,
                                                          EpiParams e) {
  __shared__ __attribute__((aligned(16))) char stg[65536];
  const long z = blockIdx.y;
  skinny_tile(A + z * stride_a, lda, B + z * stride_b, ldb, M, N, K, (long)blockIdx.x * 32, e, z * stride_c, stg);
}

static void launch_skinny(const vfm_gemm_desc* d, hipStream_t s) {
  const long batch = d->batch > 0 ? d->batch : 1;
  hipLaunchKernelGGL(k_gemm_bf16_skinny, dim3(cdiv(d->N, 32), (unsigned)batch), dim3(512), 0, s, (const bf16_t*)d->A, d->sa_m,
                     (const bf16_t*)d->B, d->sb_n, d->M, d->N, d->K, d->stride_a, d->stride_b, d->stride_c, make_epi(d));
}

static int g_force_cfg = -1;
static int g_split_tail = 3;  // bit 0: peel the tail rows of M off; bit 1: ... except for shapes on the 64-row tiles
static int g_batch_tiles = 1;  // count the batch in the tile-count heuristics
static int g_bt64 = 1;       // transposed-B GEMMs with few tiles use 64-row tiles
static int g_fold_tail = 1;  // run the tail rows as extra blocks of the tile kernel's launch
// bit 0: 256x256 ping-pong kernel instead of config 16; bit 1: 128x128 ping-pong kernel (gemm_pp.hip); bit 3: N >= 2048 stays
// on the 128x128 tiles; bit 4: whole waves of 256x256 tiles go to the 8-wave 64-wide-K-tile kernel (gemm_w4.hip); bit 5: the
// 128x128 tiles run through the five-chunk ring kernel (config 34) instead of the two-stage kernel (config 17).
// Default 40, measured inside the train step (bench.py --tune gemm_use_pp=...): 128x128 tiles / 8 waves / two blocks per CU win
// on every shape of the path - co-resident blocks overlap one tile's epilogue traffic (fc1 writes 64 MB per call) with another
// tile's main loop, which a one-block-per-CU kernel cannot - and the 2.5-K-tile ring beats the two-stage pipeline by 5-18 %
// per shape (+2.6 % images/s).  The big-tile kernels win on long-K / light-epilogue shapes (4096^3: 1.26 PF vs 1.0) and stay
// selectable per call (vfm_tune gemm_cfg 30..33).
static int g_use_pp = 184;  // round 3: 40 -> 184 (bit 4: fc1 forward / fc2 input gradient on 256 x 256 tiles, +1.7 % images/s; bit 7: the qkv projection too, +1.0 %)
long g_nt_bytes = 0;   // vfm_tune("gemm_nt_mb"): outputs of at least this many bytes are stored nontemporally (0 = never)
// the persistent two-accumulator kernel (gemm_ps.hip) for whole rounds of 256 x 256 regions with a bf16 epilogue: OFF by default.  Round 3,
// one-process A/B inside the train step (tools/ab_step.py, profiles/r03_ab_step_gemm.log): persistent kernel 132.7 images/s, 128 x 128 ring
// kernel 133.1, 256 x 256 8-wave ring kernel (gemm_use_pp bit 4) 134.5 - the form with the fewest operand bytes per flop wins once its
// epilogue is the cheap eight-column one; hiding half of the epilogue under the next sub-tile's loop does not pay
static int g_use_ps = 0;
// 192 x 256 tiles (gemm_w4.hip form 7, config 40) for the large-M GEMMs of the 1024^2 predictions (nine windows = 9216 rows) wherever they
// fill the 256 CUs' rounds better than 256 x 256 tiles do.  tools/bench_gemm_step.py SHAPES=eval (profiles/r03_gemm_eval_shapes.log):
// DINOv2 fc2 / proj / fc1 98.1 -> 83.8 / 33.2 -> 29.8 / 100.5 -> 94.8 us, SAM-H qkv / proj / fc2 106.3 -> 89.8 / 42.0 -> 37.0 / 127.5 -> 109.7 us;
// bit 1: 256 x 256 tiles for the shapes they fill to >= 80 % (DINOv2 qkv [9216 x 3072]: 69.3 -> 64.1 us).  vfm_tune gemm_use_192.
static int g_use_192 = 3;
static int g_w44_k = 1024;     // vfm_tune("gemm_w44_k"): from this K on the 128 x 128 ring kernel runs as four waves of 64 x 64 (config 52) instead of eight of
                               // 64 x 32 (config 34): +0.6 % images/s inside the train step, one process, interleaved (profiles/r04_ab_step_w44.log); 0 = never
static int g_deep_sep_k = 0;   // K from which one-tile-per-CU launches WITH tail rows take the seven-chunk ring and leave the tail rows to a skinny
                               // launch of their own (vfm_tune gemm_deep_sep_k; 0 = never)
static int g_deep_tail_k = 0;  // K from which one-tile-per-CU launches WITH tail rows take the seven-chunk ring, the tail rows riding
                               // inside the regular blocks (gemm_w4.hip CLSIN); 0 = never (five-chunk ring + tail blocks).  Measured:
                               // with cold operands 62.8 -> 53.9 us at K = 4096 (19.8 -> 22.2 at K = 1024), inside the train step
                               // (operands warm in L2 / Infinity Cache) 127.1 -> 124.6 images/s with K >= 2048: off
extern "C" int vfm_tune(const char* key, int value) {
  if (key && strcmp(key, "gemm_cfg") == 0) {
    g_force_cfg = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "pp_dbg") == 0) {
    extern int g_pp_dbg;
    g_pp_dbg = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "attn_short_grid") == 0) {
    extern int g_attn_short_grid;
    g_attn_short_grid = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "attn_il") == 0) {
    if (value < 0 || value > 3) VFM_FAIL(VFM_E_INVAL, "vfm_tune(attn_il): 0 .. 3");
    extern int g_attn_il;
    g_attn_il = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "attn_xcd") == 0) {
    extern int g_attn_xcd;
    g_attn_xcd = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "attn_fwd64") == 0) {
    extern int g_attn_fwd64;
    g_attn_fwd64 = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "attn_lds_pad") == 0) {
    extern int g_attn_lds_pad;
    g_attn_lds_pad = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "gemm_w44_k") == 0) {
    if (value < 0) VFM_FAIL(VFM_E_INVAL, "vfm_tune(gemm_w44_k): negative value");
    g_w44_k = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "gemm_deep_sep_k") == 0) {
    if (value < 0) return VFM_E_INVAL;   // (a negative value used to mean "skip the tail rows": a timing diagnostic with wrong results)
    g_deep_sep_k = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "gemm_deep_tail_k") == 0) {
    g_deep_tail_k = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "gemm_nt_mb") == 0) {
    g_nt_bytes = (long)value << 20;
    return VFM_OK;
  }
  if (key && strcmp(key, "ps_burst") == 0) {
#ifdef VFM_EXPERIMENTAL_GEMM
    extern int g_ps_burst;
    g_ps_burst = value;
#endif
    return VFM_OK;
  }
  if (key && strcmp(key, "gemm_use_ps") == 0) {
    g_use_ps = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "gemm_use_pp") == 0) {
    g_use_pp = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "gemm_use_192") == 0) {
    g_use_192 = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "gemm_batch_tiles") == 0) {
    g_batch_tiles = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "gemm_bt64") == 0) {
    g_bt64 = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "gemm_fold_tail") == 0) {
    g_fold_tail = value;
    return VFM_OK;
  }
  if (key && strcmp(key, "gemm_split_tail") == 0) {
    g_split_tail = value;
    return VFM_OK;
  }
  VFM_FAIL(VFM_E_INVAL, "vfm_tune: unknown key");
}

static bool vec_ok(const vfm_gemm_desc* d) {
  auto a16 = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
  const long bm = d->bias_mod > 0 ? d->bias_mod : d->N;
  bool ok = (d->N % 4 == 0) && (d->ldc % 4 == 0) && a16(d->C) && (d->stride_c % 4 == 0);
  if (d->bias) ok = ok && a16(d->bias) && (bm % 4 == 0);
  if (d->colscale) ok = ok && a16(d->colscale);
  if (d->residual) ok = ok && a16(d->residual) && (d->ldr % 4 == 0);
  if (d->aux) ok = ok && a16(d->aux) && (d->ld_aux % 4 == 0);
  if (d->C2) ok = ok && a16(d->C2) && (d->ldc2 % 4 == 0);
  return ok;
}

// config ids (also the sweep space of tools/bench_gemm.py)
//  0: 128x128 2x2 NS2   1: 128x128 2x2 NS3   2: 128x128 2x2 NS4   3: 128x64 2x2 NS3   4: 128x64 2x2 NS4
//  5: 256x128 2x2 NS2   6: 256x128 2x2 NS3   7: 256x128 4x2 NS3   8: 256x256 2x4 NS2  9: 128x256 2x2 NS3
//  10: 64x64 2x2 NS4    11: 256x64 4x2 NS3    12: 128x128 4x2 NS2  13: 128x128 4x2 NS3  14: 256x128 4x2 NS2
//  15: 256x256 4x2 NS2  16: 256x256 4x4 NS2   17: 128x128 2x4 NS2  18: 64x128 2x2 NS2   19: 128x64 2x2 NS2
static int gemm_main(const vfm_gemm_desc* d, hipStream_t s, const vfm_gemm_desc* tail = nullptr, bool* folded = nullptr);
bool vfm_gemm_launch_pp256(const vfm_gemm_desc* d, hipStream_t s, bool vec, const vfm_gemm_desc* tail);  // gemm_pp.hip
bool vfm_gemm_launch_pp128(const vfm_gemm_desc* d, hipStream_t s, bool vec, const vfm_gemm_desc* tail);  // gemm_pp.hip
bool vfm_gemm_launch_w4(const vfm_gemm_desc* d, hipStream_t s, bool vec, const vfm_gemm_desc* tail, int waves);  // gemm_w4.hip
bool vfm_gemm_splitk_ok(const vfm_gemm_desc* d, bool vec);                                                       // gemm_w4.hip
bool vfm_gemm_launch_w4_splitk(const vfm_gemm_desc* d, hipStream_t s, const vfm_gemm_desc* tail);              // gemm_w4.hip
// experiment vehicles, built only with VFMSEG_EXPERIMENTAL=1 (vfmseg_amd/csrc/build.py EXPERIMENTAL): the persistent two-accumulator kernel
// (gemm_ps.hip, round 3) and the 4-wave 16x16x32 256x256 kernel (gemm_v5.hip, round 4).  Neither beat the ring kernels of gemm_w4.hip
// (DESIGN.md section 5.1); without them their configs answer VFM_E_UNSUPPORTED and the knobs that select them are ignored.
#ifdef VFM_EXPERIMENTAL_GEMM
bool vfm_gemm_launch_v5(const vfm_gemm_desc* d, hipStream_t s, bool vec, const vfm_gemm_desc* tail, int form);  // gemm_v5.hip
bool vfm_gemm_ps_ok(const vfm_gemm_desc* d, int per);                                                       // gemm_ps.hip
bool vfm_gemm_launch_ps(const vfm_gemm_desc* d, hipStream_t s, const vfm_gemm_desc* tail, int per);         // gemm_ps.hip
#else
static bool vfm_gemm_launch_v5(const vfm_gemm_desc*, hipStream_t, bool, const vfm_gemm_desc*, int) { return false; }
static bool vfm_gemm_ps_ok(const vfm_gemm_desc*, int) { return false; }
static bool vfm_gemm_launch_ps(const vfm_gemm_desc*, hipStream_t, const vfm_gemm_desc*, int) { return false; }
#endif

int vfm_gemm_bf16_impl(const vfm_gemm_desc* d0, hipStream_t s) {
  // A few rows past a 128-row boundary (M = B*1024 patch tokens + B [cls] tokens) would cost a whole extra row of
  // latency-bound tiles: peel them off into the skinny kernel.
  const long tail = d0->M % 128;
  const bool bt = (d0->sb_n == 1 && d0->sb_k != 1);
  if (bt) return gemm_main(d0, s);
  // shapes that go to the 64-row tiles (few column tiles, long K: the LoRA T GEMM [4100 x 64 x 1024]) take the tail as one
  // more row tile of the same launch - cheaper than a launch of its own
  // (a few tiles over the 512 resident slots still beat a second launch: the coarse 512 x 1024 inference pass, M = 2049, N = 1024,
  // has 33 x 16 tiles and paid 10 us per GEMM for its one [cls] row)
  // (... except where the ping-pong kernel with the tail folded in as skinny blocks is faster still: half a wave of 128 x 128 tiles,
  // N >= 512, K >= 1024 - the rule in gemm_main)
  const long t128m = (long)((d0->M - tail) / 128) * cdiv(d0->N, 128);
  const bool pp_coarse = !(g_use_pp & 64) && tail > 0 && t128m >= 96 && t128m <= 160 && d0->N >= 512 && d0->K >= 1024 && d0->batch <= 1;
  const bool rows64 = !pp_coarse && d0->N > 32 && d0->K >= 512 && (long)cdiv(d0->M, 128) * cdiv(d0->N, 128) <= 160 &&
                      (long)cdiv(d0->M, 64) * cdiv(d0->N, 64) <= 544;
  if (g_split_tail && !(rows64 && (g_split_tail & 2)) && d0->M > 512 && tail > 0 && tail <= 32 && (d0->batch <= 1)) {
    vfm_gemm_desc dm = *d0, dt = *d0;
    const long mm = d0->M - tail;
    dm.M = mm;
    dt.M = tail;
    auto adv = [&](const void* p, int dt_, long ld) -> const void* {
      return p ? (const void*)((const char*)p + mm * ld * (dt_ == VFM_BF16 || dt_ == VFM_SPLIT3 ? 2 : 4)) : nullptr;
    };
    dt.A = adv(d0->A, VFM_BF16, d0->sa_m);
    dt.C = (void*)adv(d0->C, d0->c_dt, d0->ldc);
    dt.residual = adv(d0->residual, d0->r_dt, d0->ldr);
    dt.aux = adv(d0->aux, d0->aux_dt, d0->ld_aux);
    dt.C2 = (void*)adv(d0->C2, d0->c2_dt, d0->ldc2);
    bool folded = false;
    int rc = gemm_main(&dm, s, g_fold_tail ? &dt : nullptr, &folded);
    if (rc) return rc;
    if (!folded) launch_skinny(&dt, s);
    return VFM_OK;
  }
  if (d0->M <= 32 && d0->N >= 32) {
    launch_skinny(d0, s);
    return VFM_OK;
  }
  return gemm_main(d0, s);
}

static int gemm_main(const vfm_gemm_desc* d, hipStream_t s, const vfm_gemm_desc* tail, bool* folded) {
  const bool vec = vec_ok(d);
  bool fd = false;
  if (d->sb_n == 1 && d->sb_k != 1) {  // B given as [K, N]: transposed-B kernels (BN = 128)
    if (d->sa_m == 1 && d->sa_k != 1) {  // ... and A given as [K, M]: C = A^T B, both operands consumed in place
      if (vec) launch_one<128, 128, 2, 4, 2, true, true, true>(d, s);
      else launch_one<128, 128, 2, 4, 2, false, true, true>(d, s);
      return VFM_OK;
    }
    // few 128-row tiles (decoder dgrads: 2048 x 256 outputs over K = 1024..2048): 64-row tiles double the blocks in flight
    const bool few = (long)cdiv(d->M, 128) * cdiv(d->N, 128) * (d->batch > 0 ? d->batch : 1) < 128;
    if (d->M <= 64 || (g_bt64 && few)) launch_bt<64, 1, 4>(d, s, vec);  // (a four-stage ring measured no faster: one wave per SIMD)
    else launch_bt<128, 2, 4>(d, s, vec);
    return VFM_OK;
  }
  int cfg = g_force_cfg;
  if (cfg < 0) {
    // measured on MI355X (tools/bench_gemm.py): with ~one wave of tiles, occupancy (waves per SIMD) decides
    const long nbatch = d->batch > 0 ? d->batch : 1;  // batched launches (SAM's per-window products) fill the chip with their batch
    const long t128 = (long)cdiv(d->M, 128) * cdiv(d->N, 128) * (g_batch_tiles ? nbatch : 1);
    const long t256 = (long)cdiv(d->M, 256) * cdiv(d->N, 256);
    const bool span31 = (d->M + 128) * d->sa_m < (1l << 31) && (d->N + 128) * d->sb_n < (1l << 31);
    const bool span33 = (d->M + 256) * d->sa_m < (1l << 31) && (d->N + 256) * d->sb_n < (1l << 31);
    // fill of the CUs' rounds by whole tiles (one block per CU for the 256-column forms)
    const long t192 = (long)cdiv(d->M, 192) * cdiv(d->N, 256);
    const double eff192 = (double)t192 / (256.0 * cdiv(t192, 256)), eff256 = (double)t256 / (256.0 * cdiv(t256, 256));
    const bool big_m = d->M >= 8192 && d->N >= 1024 && d->K >= 1024 && nbatch == 1 && span33 && !(g_use_pp & 64);
    if (d->N <= 32) cfg = 10;                                        // 64x64 tiles: many rows, few columns
    else if ((g_use_192 & 1) && big_m && eff192 >= eff256 + 0.08)
      cfg = 40;  // 192 x 256 tiles: [9216 x 1024 / 1280]: 192 / 240 tiles instead of 144 / 180; [9216 x 3840 / 4096]: 720 / 768 of 768 slots
    else if ((g_use_192 & 2) && big_m && d->N >= 2048 && t256 > 256 && eff256 >= 0.8 && eff256 < 0.9)
      cfg = 33;  // [9216 x 3072]: 432 tiles of 256 x 256 = 84 % of two rounds (the 128 x 128 form: 1728 tiles, 3.4 rounds of 512 slots)
    else if ((g_use_ps & 1) && d->N >= 2048 && t256 >= 224 && t256 % 256 == 0 && d->K >= 1024 && vfm_gemm_ps_ok(d, 2))
      cfg = 37;  // whole rounds of 256 x 256 regions with a bf16 store-heavy epilogue (fc1 forward, fc2 input gradient [4096 x 4096 x 1024]):
                 // one persistent block per CU, the first half's epilogue under the second half's K loop
    else if (!(g_use_pp & 64) && tail && t128 >= 96 && t128 <= 160 && d->N >= 512 && d->K >= 1024 && span31 && nbatch == 1)
      cfg = 31;  // half a wave of 128x128 tiles over a long K in a backbone GEMM (tail rows present: the coarse eval pass, 2049 tokens x
                 // 1024 columns): the ping-pong kernel beats the 64x64 tiles by 10-20 % (tools/scratch/_coarse_gemm.py: 15.9 / 36.3 us
                 // against 17.5 / 45.8 at K = 1024 / 4096).  The decoder's 2048-token GEMMs of the train step (no tail) keep the
                 // 64x64 tiles: 125.3 vs 124.9 images/s
    else if (!(g_use_pp & 64) && d->N >= 1024 && d->N < 2048 && t256 >= 128 && t256 <= 200 && span33 && nbatch == 1 &&
             (d->K >= 2048 || (t256 >= 170 && d->K >= 1024)))
      cfg = 33;  // 130-200 tiles of 256x256 (eval: nine 1024-token windows at once - DINOv2 fc2 [9225 x 1024 x 4096] 104.5 -> 94.9 us,
                 // SAM-H proj / fc2 [9216 x 1280 x 1280 / 5120] 51.6 -> 46.9 / 142.5 -> 118.0 us): 576-720 tiles of 128x128 are 1.1-1.4
                 // rounds of the 512 resident slots
    else if (!(g_use_pp & 64) && d->N >= 4096 && t256 >= 700 && t256 < 768 && d->K >= 1024 && span33 && nbatch == 1)
      cfg = 33;  // SAM-H fc1 over nine windows [9216 x 5120 x 1280]: 189.4 -> 160.5 us
    else if (!(g_use_pp & 64) && d->N >= 2048 && t256 >= 224 && t256 <= 256 && d->K >= 1024 && span33 && nbatch == 1 && !tail)
      cfg = 33;  // one 256x256 tile on (almost) every CU - SAM-H qkv + LoRA in the train step [4096 x 3840 x 1344]: 866 -> 1016 TFLOP/s
    else if (t128 <= 160 && d->K >= 512 && (long)cdiv(d->M, 64) * cdiv(d->N, 64) <= 544)
      cfg = 10;  // few tiles, long K (LoRA T GEMM 4096x64x1024, decoder projections): 64x64 tiles with the 4-stage ring
    else if (t128 <= 160) cfg = 18;                                  // small problems: 64x128 tiles fill more CUs
    else if ((g_use_pp & 2) && t128 > 160 && t128 <= 272 && d->K >= 256 && (d->M + 128) * d->sa_m < (1l << 31) && (d->N + 128) * d->sb_n < (1l << 31))
      cfg = 31;  // about one 128x128 tile per CU: the ping-pong kernel (one block per CU, 4-slot DMA ring)
    else if ((g_use_pp & 128) && d->N >= 2048 && t256 >= 176 && t256 < 256 && d->K >= 512 && span33 && nbatch == 1)
      cfg = 33;  // three quarters of a round of 256 x 256 tiles (qkv + LoRA forward [4096 x 3072 x 1088]: 192 tiles)
    else if ((g_use_pp & 16) && d->N >= 2048 && (t256 % 256 == 0 || t256 >= 768) && d->K >= 128 && (d->M + 256) * d->sa_m < (1l << 31) &&
             (d->N + 256) * d->sb_n < (1l << 31))
      cfg = 33;  // whole waves of 256x256 tiles: 8 waves, 64-wide K-tiles (whole-line LDS-DMA), 5-chunk ring
    else if (d->N >= 2048 && (t256 % 256 == 0 || t256 >= 768) && !(g_use_pp & 8))
      cfg = ((g_use_pp & 1) && d->K >= 128 && (d->M + 256) * d->sa_m < (1l << 31) && (d->N + 256) * d->sb_n < (1l << 31)) ? 30 : 16;  // 256x256 tiles
    else cfg = 17;                                                   // 128x128, 8 waves, 2 blocks per CU
    if (cfg == 17 && (g_use_pp & 32) && d->K >= 128 && (d->M + 128) * d->sa_m < (1l << 31) && (d->N + 128) * d->sb_n < (1l << 31))
      cfg = 34;  // same tile and wave layout, operands through the five-chunk (2.5 K-tile) LDS-DMA ring of gemm_w4.hip
    if (cfg == 34 && g_w44_k > 0 && d->K >= g_w44_k) cfg = 52;   // vfm_tune("gemm_w44_k"): the 4-wave form of the same tile from this K on (0 = never)
    // at most one tile per CU and no tail blocks to squeeze in: the second block's LDS buys a seven-chunk ring instead
    if (cfg == 34 && !tail && t128 <= 256 && d->K >= 256 && d->M % 128 == 0) cfg = 35;
    // ... and with tail blocks riding along when K is long: one tile per CU streams its operands latency-bound (bytes in flight /
    // latency), so the deeper ring wins more than the tail blocks lose by waiting for a CU (they cannot share the ring's LDS)
    if (cfg == 34 && tail && t128 <= 256 && d->M % 128 == 0 && d->K >= g_deep_tail_k && g_deep_tail_k > 0) cfg = 35;
    // ... or with the tail rows in a launch of their own behind this one (they cannot share a CU with the deep ring's LDS)
    if (cfg == 34 && tail && t128 <= 256 && d->M % 128 == 0 && d->K >= 256 && g_deep_sep_k > 0 && d->K >= g_deep_sep_k) {
      if (folded) *folded = false;   // the caller runs the skinny kernel for the tail rows
      vfm_gemm_launch_w4(d, s, vec, nullptr, 3);
      return VFM_OK;
    }
  }
  switch (cfg) {
    case 0: fd = launch_cfg<128, 128, 2, 2, 2>(d, s, vec, tail); break;
    case 1: fd = launch_cfg<128, 128, 2, 2, 3>(d, s, vec, tail); break;
    case 2: fd = launch_cfg<128, 128, 2, 2, 4>(d, s, vec, tail); break;
    case 3: fd = launch_cfg<128, 64, 2, 2, 3>(d, s, vec, tail); break;
    case 4: fd = launch_cfg<128, 64, 2, 2, 4>(d, s, vec, tail); break;
    case 5: fd = launch_cfg<256, 128, 2, 2, 2>(d, s, vec, tail); break;
    case 6: fd = launch_cfg<256, 128, 2, 2, 3>(d, s, vec, tail); break;
    case 7: fd = launch_cfg<256, 128, 4, 2, 3>(d, s, vec, tail); break;
    case 8: fd = launch_cfg<256, 256, 2, 4, 2>(d, s, vec, tail); break;
    case 9: fd = launch_cfg<128, 256, 2, 2, 3>(d, s, vec, tail); break;
    case 10: fd = launch_cfg<64, 64, 2, 2, 4>(d, s, vec, tail); break;
    case 11: fd = launch_cfg<256, 64, 4, 2, 3>(d, s, vec, tail); break;
    case 12: fd = launch_cfg<128, 128, 4, 2, 2>(d, s, vec, tail); break;
    case 13: fd = launch_cfg<128, 128, 4, 2, 3>(d, s, vec, tail); break;
    case 14: fd = launch_cfg<256, 128, 4, 2, 2>(d, s, vec, tail); break;
    case 15: fd = launch_cfg<256, 256, 4, 2, 2>(d, s, vec, tail); break;
    case 16: fd = launch_cfg<256, 256, 4, 4, 2>(d, s, vec, tail); break;
    case 17: fd = launch_cfg<128, 128, 2, 4, 2>(d, s, vec, tail); break;
    case 18: fd = launch_cfg<64, 128, 2, 2, 2>(d, s, vec, tail); break;
    case 19: fd = launch_cfg<128, 64, 2, 2, 2>(d, s, vec, tail); break;
    case 20: fd = launch_cfg<256, 128, 4, 4, 3>(d, s, vec, tail); break;
    case 21: fd = launch_cfg<128, 256, 4, 4, 3>(d, s, vec, tail); break;
    case 22: fd = launch_cfg<256, 128, 4, 4, 2>(d, s, vec, tail); break;
    case 23: fd = launch_cfg<128, 128, 4, 4, 3>(d, s, vec, tail); break;
    case 24: fd = launch_cfg<128, 128, 4, 4, 2>(d, s, vec, tail); break;
    case 25: fd = launch_cfg<64, 128, 2, 2, 4>(d, s, vec, tail); break;
    case 26: fd = launch_cfg<32, 64, 1, 2, 4>(d, s, vec, tail); break;
    case 27: fd = launch_cfg<32, 128, 1, 4, 4>(d, s, vec, tail); break;
    case 28: fd = launch_cfg<64, 64, 2, 2, 6>(d, s, vec, tail); break;
    case 29: fd = launch_cfg<32, 64, 1, 2, 8>(d, s, vec, tail); break;
    case 31:
      VFM_CHECK(d->K >= 256, VFM_E_UNSUPPORTED, "vfm_gemm(bf16): the 128x128 ping-pong kernel needs K >= 256");
      fd = vfm_gemm_launch_pp128(d, s, vec, tail);
      break;
    case 32:
    case 33:
    case 34:
    case 35:
    case 36:
      VFM_CHECK(d->K >= 128 && d->K % 64 == 0 && (d->M + 256) * d->sa_m < (1l << 31) && (d->N + 256) * d->sb_n < (1l << 31), VFM_E_UNSUPPORTED,
                "vfm_gemm(bf16): the 4-wave 256x256 kernel needs K >= 128 and operands spanning < 4 GiB");
      VFM_CHECK(cfg < 35 || d->K >= 256, VFM_E_UNSUPPORTED, "vfm_gemm(bf16): the deep-ring 128x128 kernels need K >= 256");
      fd = vfm_gemm_launch_w4(d, s, vec, tail, cfg == 32 ? 4 : (cfg == 33 ? 8 : (cfg == 34 ? 2 : (cfg == 35 ? 3 : 5))));
      break;
    case 51:   // 128 x 128 ring kernel, two blocks per tile (one per half of K), in-kernel fix-up.  Never chosen by the rules above: on the
               // shapes it was built for (the N = 1024 GEMMs of the coarse prediction pass, 128 tiles) the exchange of the partial sums costs
               // what the second half of the CUs gains (fc2 [2049 x 1024 x 4096] 38.6 -> 37.6 us, proj 14.0 -> 17.9: profiles/r04_gemm_splitk_eval.log)
      VFM_CHECK(d->K % 64 == 0 && (d->M + 256) * d->sa_m < (1l << 31) && (d->N + 256) * d->sb_n < (1l << 31) && vfm_gemm_splitk_ok(d, vec), VFM_E_UNSUPPORTED,
                "vfm_gemm(bf16): the split-K ring kernel needs K %% 128 == 0, K >= 512, one batch, <= 512 tiles of 128 x 128, a 16-byte aligned epilogue");
      fd = vfm_gemm_launch_w4_splitk(d, s, tail);
      break;
    case 52:   // 128 x 128 tiles, 4 waves of 64 x 64 (gemm_w4.hip form 9)
      VFM_CHECK(d->K >= 128 && d->K % 64 == 0 && (d->M + 256) * d->sa_m < (1l << 31) && (d->N + 256) * d->sb_n < (1l << 31), VFM_E_UNSUPPORTED,
                "vfm_gemm(bf16): the 4-wave 128x128 kernel needs K >= 128 and operands spanning < 4 GiB");
      fd = vfm_gemm_launch_w4(d, s, vec, tail, 9);
      break;
    case 39:   // 192 x 256 tiles, 4 waves (gemm_w4.hip form 6)
    case 40:   // 192 x 256 tiles, 8 waves (form 7)
      VFM_CHECK(d->K >= 128 && d->K % 64 == 0 && (d->M + 256) * d->sa_m < (1l << 31) && (d->N + 256) * d->sb_n < (1l << 31), VFM_E_UNSUPPORTED,
                "vfm_gemm(bf16): the 192x256 kernel needs K >= 128 and operands spanning < 4 GiB");
      fd = vfm_gemm_launch_w4(d, s, vec, tail, cfg == 39 ? 6 : 7);
      break;
    case 50:   // 256 x 256 tiles, 4 waves, 16x16x32 MFMA, two whole K-tile stages (gemm_v5.hip)
#ifndef VFM_EXPERIMENTAL_GEMM
      VFM_FAIL(VFM_E_UNSUPPORTED, "vfm_gemm(bf16): config 50 (gemm_v5.hip) is an experiment: build with VFMSEG_EXPERIMENTAL=1");
#endif
      VFM_CHECK(d->K >= 128 && d->K % 64 == 0 && (d->M + 256) * d->sa_m < (1l << 31) && (d->N + 256) * d->sb_n < (1l << 31), VFM_E_UNSUPPORTED,
                "vfm_gemm(bf16): the 16x16x32 256x256 kernel needs K >= 128 and operands spanning < 4 GiB");
      fd = vfm_gemm_launch_v5(d, s, vec, tail, 0);
      break;
    case 37:
    case 38:
      VFM_CHECK(vfm_gemm_ps_ok(d, cfg == 37 ? 2 : 1), VFM_E_UNSUPPORTED,
                "vfm_gemm(bf16): the persistent kernel needs M %% 256 == 0, N %% %d == 0, K %% 64 == 0, K >= 576 and a bf16 bias / GELU / multiply epilogue",
                cfg == 37 ? 256 : 128);
      fd = vfm_gemm_launch_ps(d, s, tail, cfg == 37 ? 2 : 1);
      break;
    case 30:
      VFM_CHECK(d->K >= 128, VFM_E_UNSUPPORTED, "vfm_gemm(bf16): the ping-pong kernel needs K >= 128");
      fd = vfm_gemm_launch_pp256(d, s, vec, tail);
      break;
    default: VFM_FAIL(VFM_E_INVAL, "vfm_gemm(bf16): unknown config %d", cfg);
  }
  if (folded) *folded = fd && tail;
  return VFM_OK;
}
