// Flash-style SAM attention forward for gfx950 (head dim 80): windowed (14 x 14 windows of the zero-padded grid) and global
// attention with the decomposed relative-position bias, straight from the token-major qkv matrix to the token-major output.
// Reference: rein/models/backbones/sam_vit.py:273-298 (Attention.forward), :301-356 (window partition / unpartition),
// :359-430 (get_rel_pos, add_decomposed_rel_pos).  Replaces, for inference, the materialised form of sam.hip (prep -> batched
// score GEMM -> fp32 score matrix -> row softmax -> P V GEMM -> merge: six launches and ~0.4 GB of traffic per layer).
//
//   score[q, k] = scale * q.k + q.Rh[qh, kh] + q.Rw[qw, kw],      Rh[qh, kh] = tbl_h[qh - kh + S - 1]  (relative-index table)
//
// * The bias needs only T_h = Q tbl_h^T and T_w = Q tbl_w^T  ([queries x (2S-1)], two small MFMA products per wave, K = 80):
//   Bh[q, kh] = T_h[q, qh(q) - kh + S - 1].  Each lane gathers its query's S + S values once (through a wave-private LDS image).
// * It then enters the scores through the MFMA itself: the query operand is extended to [q | Bh[q,:]/scale | Bw[q,:]/scale] and the
//   key tile in LDS to [k | onehot(kh) | onehot(kw)] (the one-hot columns are written by the tile loader from the key index), so
//   the biased score is one dot product of length 80 + 2 SP and the softmax loop carries no per-score gathers.
// * One wave owns 32 queries on the MFMA lane index (S^T = K Q^T, O^T += V^T P^T; P never leaves registers), as in attention_bf16.hip;
//   K / V tiles of 64 keys are staged global -> registers -> LDS (rows are 160 B: no whole-line LDS-DMA shape), double buffered,
//   one barrier per tile.  Window tokens outside the image are real keys whose k / v equal the projection bias (window_partition
//   pads the NORMALISED input with zeros); queries outside the image are computed and dropped (window_unpartition).
// * What the profile asked for (profiles/r02_pmc_sam_flash.txt: the SIMD issue port, not the matrix pipe, is the busy resource): blocks of
//   4 waves whose prologue images alias the K/V ring (two blocks per CU); one key row per staging thread (SfKvStager); the softmax
//   denominator as row 80 of O^T through a ones column of V; output rows through LDS as 16-byte pieces; all prologue loads in one round
//   trip.  SF_EXP (compile-time) keeps the ablations and phase clocks those numbers came from (tools/scratch/_sam_flash_exp.py).
// * Training (vfm_sam_attn_flash_fwd_train) also writes lse (log2 domain) and the bias columns of the query operand for sam_flash_bwd.hip.
#include "sam_flash_dev.h"

template <int S>
__global__ void __launch_bounds__(SamFlashCfg<S>::NT, 2) k_sam_flash_fwd(SamFlashP p) {
  using C = SamFlashCfg<S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, h = lane >> 5;
#if SF_EXP == 9   // phase clocks of wave 0 (s_memtime), summed over the blocks into p.lse[0..15] (tools/scratch/_sam_flash_exp.py)
  unsigned long long sf_tprev = __builtin_amdgcn_s_memtime();
  unsigned sf_tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SF_T(k) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); sf_tacc[k] += (unsigned)(now_ - sf_tprev); sf_tprev = now_; }
#else
#define SF_T(k)
#endif
  // block -> (image, window, head, query block).  Workgroups go round-robin to the eight XCDs by linear id and each XCD has its own
  // L2: all H * QBLK blocks of one window (the same K / V rows; neighbouring heads share cache lines of the 160-byte segments) are
  // given ids congruent mod 8, so that a window's keys are fetched through ONE L2.  The last (windows % 8) windows keep the plain order.
  int bid = blockIdx.x;
  {
    const int inner = p.H * C::QBLK, ngroups = p.nimg * p.nws * p.nws, full = ngroups & ~7;
    if (bid < full * inner) {
      const int xcd = bid & 7, idx = bid >> 3;
      bid = ((idx / inner) * 8 + xcd) * inner + idx % inner;
    }
  }
  const int qb = bid % C::QBLK; bid /= C::QBLK;
  const int head = bid % p.H; bid /= p.H;
  const int wx = bid % p.nws; bid /= p.nws;
  const int wy = bid % p.nws;
  const int img = bid / p.nws;
  const int G = p.G, Cq = p.H * SF_D;
  auto tok_row = [&](int t, bool& inside) -> long {   // window token index -> row of the token-major matrices
    const int ty = t / S, tx = t - ty * S;
    const int gy = wy * S + ty, gx = wx * S + tx;
    inside = gy < G && gx < G;
    return ((long)img * G + gy) * G + gx;
  };
  // one 16-byte piece (8 columns from column c8) of section sec (0 q, 1 k, 2 v) of window token t.  A token outside the image has
  // k / v = the projection bias, read from the block's packed image in LDS (a global load there would put a full-latency
  // s_waitcnt vmcnt(0) into every staging step of an edge window); a query outside the image is dropped, any value will do.
  const char* bimg = smem + 2 * C::TILE;            // [2 sections][80 columns] bf16
  auto load_piece = [&](int sec, int t, int c8) -> uint4 {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t < C::NWIN) {
      bool inside;
      const long row = tok_row(t, inside);
      if (inside) v = *reinterpret_cast<const uint4*>(p.qkv + row * p.ld + sec * Cq + head * SF_D + c8);
      else if (sec > 0) v = *reinterpret_cast<const uint4*>(bimg + (sec - 1) * (2 * SF_D) + c8 * 2);
    }
    return v;
  };
  // ---- K/V tile staging (SfKvStager, sam_flash_dev.h)
  const SfGeo geo{img, wy, wx, G};
  SfKvStager<S> stager;
  auto fetch = [&](int t) __attribute__((always_inline)) { stager.fetch(p.qkv, p.ld, Cq, head, bimg, geo, t, tid); };
  auto commit = [&](int buf, int t) __attribute__((always_inline)) { stager.commit(smem, buf, t, tid); };
  // ---- prologue 1: the bias image, the table images (aliased with the end of K/V stage 1) and this wave's query fragments.  ALL their
  // global loads are issued before the first one is consumed: one round trip (a runtime loop over the table pieces made it one per pass)
  char* timg = smem + 2 * C::TILE - C::TIMG;             // [2][JP rows][176 B]
  const int q0 = qb * (C::NW * 32) + wave * 32;
  const int qi = q0 + fr;                            // this lane's query (window token index)
  const int qc = qi < C::NWIN ? qi : C::NWIN - 1;    // clamped: surplus queries of the last wave are computed and dropped
  bf16x8 qa[C::KSTEPS];
  {
    constexpr int NTP = 2 * C::JP * 10, NTI = (NTP + C::NT - 1) / C::NT;
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    const int bsec = 1 + tid / 10, bc8 = (tid % 10) * 8;
    if (tid < 20 && p.bias) {
      const float* b = p.bias + bsec * Cq + head * SF_D + bc8;
      b0 = *reinterpret_cast<const float4*>(b), b1 = *reinterpret_cast<const float4*>(b + 4);
    }
    uint4 tp[NTI];
#pragma unroll
    for (int i = 0; i < NTI; ++i) {
      const int pc = tid + i * C::NT;
      tp[i] = make_uint4(0, 0, 0, 0);
      if (pc < NTP) {
        const int which = pc / (C::JP * 10), rem = pc - which * (C::JP * 10);
        tp[i] = *reinterpret_cast<const uint4*>((which ? p.tbl_w : p.tbl_h) + rem * 8);   // (row, piece) of a [JP, 80] table = piece rem of its flat array
      }
    }
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
      const uint4 v = load_piece(0, qc, 16 * kk + 8 * h);
      qa[kk] = *reinterpret_cast<const bf16x8*>(&v);
    }
    stager.fetch_first_global(p.qkv, p.ld, Cq, head, geo, tid);   // the first K/V tile (its tokens inside the image) flies with the prologue loads, the table products and the gather
    if (tid < 20)
      *reinterpret_cast<uint4*>(smem + 2 * C::TILE + (bsec - 1) * (2 * SF_D) + bc8 * 2) =
          make_uint4(sf_pack2(b0.x, b0.y), sf_pack2(b0.z, b0.w), sf_pack2(b1.x, b1.y), sf_pack2(b1.z, b1.w));
#pragma unroll
    for (int i = 0; i < NTI; ++i) {
      const int pc = tid + i * C::NT;
      if (pc < NTP) {
        const int which = pc / (C::JP * 10), rem = pc - which * (C::JP * 10), row = rem / 10, c = rem - row * 10;
        *reinterpret_cast<uint4*>(timg + (which * C::JP + row) * C::TS + c * 16) = tp[i];
      }
    }
  }
  __syncthreads();
  SF_T(0)
  stager.fetch_first_bias(bimg, geo, tid);   // (its padded tokens: from the bias image, after the barrier)
  // ---- prologue 2: T^T[j, q] = tbl[j, :] . q  (rows j = relative index), both axes, into the wave-private images
  // (stored as bf16 of T / scale, the value the query operand carries; aliased with the start of the K/V ring)
  vfm_h* th = reinterpret_cast<vfm_h*>(smem) + wave * (2 * C::JP * 32);
  const float inv = 1.0f / p.scale;
#pragma unroll
  for (int which = 0; which < 2; ++which)
#pragma unroll
    for (int jb = 0; jb < C::JP / 32; ++jb) {
      f32x16 acc = sf_zero();
#pragma unroll
      for (int kk = 0; kk < 5; ++kk) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(timg + (which * C::JP + jb * 32 + fr) * C::TS + (2 * kk + h) * 16);
        acc = SF_MFMA(a, qa[kk], acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) th[(which * C::JP + jb * 32 + sf_acc_row(r, h)) * 32 + fr] = (vfm_h)(acc[r] * inv);
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  {  // gather Bh[q, kh] / scale, Bw[q, kw] / scale into the extension k-steps of the query operand
    const int qh = qc / S, qw = qc - qh * S;
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
      for (int ks = 0; ks < C::SP / 16; ++ks) {
        bf16x8 u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int kx = 16 * ks + 8 * h + e;        // key coordinate along this axis
          const int j = (which ? qw : qh) - kx + S - 1;
          u[e] = kx < S ? th[(which * C::JP + j) * 32 + fr] : (vfm_h)0.f;
        }
        qa[5 + which * (C::SP / 16) + ks] = u;
      }
  }
  const long stat_row = ((((long)img * p.nws + wy) * p.nws + wx) * p.H + head) * C::NWINP + qi;
  if (p.qext && qi < C::NWIN) {   // training: the backward reuses the bias columns instead of repeating the table products
#pragma unroll
    for (int e = 0; e < 2 * C::SP / 16; ++e)
      *reinterpret_cast<bf16x8*>(p.qext + stat_row * (2 * C::SP) + 16 * e + 8 * h) = qa[5 + e];
  }
  SF_T(1)
  __syncthreads();  // everyone is done with the prologue images: the K/V ring may be overwritten
  SF_T(2)

  // columns 80..95 of both V stages are never written by the loader: set them once - column 80 to ONE, the rest to zero.  Row 80 of
  // O^T = sum_k P[k, q] then IS the softmax denominator (of the bf16 probabilities the product uses, rescaled with the other rows):
  // the 32 row-sum adds per tile and lane leave the VALU, which is what bounds this kernel.
  for (int i = tid; i < 2 * 64 * 2; i += C::NT) {
    const int buf = i / 128, rem = i - buf * 128, row = rem >> 1, c = rem & 1;
    *reinterpret_cast<uint4*>(smem + buf * C::TILE + 64 * C::KS + row * C::VS + 160 + c * 16) = make_uint4(c == 0 ? VFM_H_ONE : 0u, 0, 0, 0);
  }
  commit(0, 0);
  SF_T(3)
  __syncthreads();
  SF_T(4)

  const float c = p.scale * SF_LOG2E;
  const SfTrLane trv(lane, C::VS, false);
  f32x16 oacc[3] = {sf_zero(), sf_zero(), sf_zero()};
  float m = -INFINITY;
  const bool active = (C::NWIN % (C::NW * 32) == 0) || q0 < C::NWIN;   // wave-uniform
  // One tile of 64 keys for this wave: NKB 32-key blocks enter the score product, the last of them feeds NSL 16-key steps of P V.
  // (A 14 x 14 window has 196 = 3 * 64 + 4 keys: its last tile is one block and one step.)
  auto tile = [&](auto nkb_c, auto nsl_c, int t, const char* kt, const char* vt) __attribute__((always_inline)) {
    constexpr int NKB = decltype(nkb_c)::value, NSL = decltype(nsl_c)::value;
    // S^T = Kext Q^T: the row fragments are read four MFMAs ahead of their use and the two 32-key chains alternate (the compiler's own
    // order keeps two reads in flight and waits before every pair of MFMAs: ~1.5k cycles of exposed LDS latency per tile at 2 waves per SIMD)
    f32x16 sacc[NKB];
    constexpr int NQK = NKB * (SF_EXP == 4 ? 1 : C::KSTEPS), DEP = 4;
    auto kfrag = [&](int i) __attribute__((always_inline)) -> bf16x8 {
      const int kb = i % NKB, kk = i / NKB;
      return *reinterpret_cast<const bf16x8*>(kt + (kb * 32 + fr) * C::KS + (2 * kk + h) * 16);
    };
    bf16x8 fk[DEP];
#pragma unroll
    for (int i = 0; i < DEP && i < NQK; ++i) fk[i] = kfrag(i);
    __builtin_amdgcn_sched_group_barrier(0x100, DEP < NQK ? DEP : NQK, 0);
#pragma unroll
    for (int i = 0; i < NQK; ++i) {
      const int kb = i % NKB, kk = i / NKB;
      sacc[kb] = SF_MFMA(fk[i % DEP], qa[kk], kk == 0 ? sf_zero() : sacc[kb]);
      if (i + DEP < NQK) fk[i % DEP] = kfrag(i + DEP);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // (pins the order: one MFMA, then the read four steps ahead)
      if (i + DEP < NQK) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    SF_T(5)
    // the V^T fragments of the first P V step do not depend on the softmax: read them now
    constexpr int NST = (SF_EXP == 3 ? 0 : (NKB - 1) * 2 + NSL);
    auto vfrag = [&](int n, int j) __attribute__((always_inline)) -> bf16x8 { return trv.frag(vt + 16 * n * C::VS, j); };
    bf16x8 vf[2][3];
    if (NST > 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) vf[0][j] = vfrag(0, j);
    }
    if ((C::NWIN % 64 != 0) && t == C::NTILES - 1) {  // ragged last tile: keys beyond the window do not exist
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * 64 + kb * 32 + sf_acc_row(r, h) >= C::NWIN) sacc[kb][r] = -INFINITY;
    }
    // ---- online softmax over this lane's keys of the tile (the others live in lane ^ 32)
    float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) m4[r & 3] = fmaxf(m4[r & 3], sacc[kb][r]);
    float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const bool need = (mx - m) * c > 8.0f;          // lazy rescale (as attention_bf16.hip): p <= 2^8 keeps bf16 / fp32 sums exact enough
    const float mn = need ? mx : m;
    const float alpha = need ? __builtin_amdgcn_exp2f((m - mn) * c) : 1.0f;
    const float mnc = mn * c;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = SF_EXP == 2 ? fmaf(sacc[kb][r], c, -mnc) : __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], c, -mnc));
        sacc[kb][r] = pv;
      }
    m = mn;
    if (__ballot(alpha != 1.0f) != 0ull) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] *= alpha;
    }
    SF_T(6)
    // ---- O^T[col, query] += V^T[col x key] P^T[key x query], step n = 16 keys; the fragments of step n + 1 are read before the MFMAs of step n
#pragma unroll
    for (int n = 0; n < NST; ++n) {
      if (n + 1 < NST) {
#pragma unroll
        for (int j = 0; j < 3; ++j) vf[(n + 1) & 1][j] = vfrag(n + 1, j);
      }
      bf16x8 pb;
#pragma unroll
      for (int e = 0; e < 8; ++e) pb[e] = (vfm_h)sacc[n >> 1][8 * (n & 1) + e];
#pragma unroll
      for (int j = 0; j < 3; ++j) oacc[j] = SF_MFMA(vf[n & 1][j], pb, oacc[j]);
    }
  };
  constexpr int LASTK = C::NWIN - 64 * (C::NTILES - 1);     // keys of the last tile
  constexpr int NKB_LAST = (LASTK + 31) / 32, NSL_LAST = ((LASTK - 1) % 32 + 16) / 16;
#pragma unroll 1
  for (int t = 0; t < C::NTILES; ++t) {
    const int buf = t & 1;
    SF_T(10)
    if (SF_EXP != 1 && t + 1 < C::NTILES) fetch(t + 1);            // global loads of the next tile fly during this tile's products
    const char* kt = smem + buf * C::TILE;
    const char* vt = kt + 64 * C::KS;
    SF_T(5)
    if (SF_EXP != 5 && active) {                                   // a wave past the window's last query only helps with the tile staging
      if ((NKB_LAST < 2 || NSL_LAST < 2) && t == C::NTILES - 1)
        tile(std::integral_constant<int, NKB_LAST>{}, std::integral_constant<int, NSL_LAST>{}, t, kt, vt);
      else
        tile(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, t, kt, vt);
    }
    SF_T(7)
    if (SF_EXP != 1 && t + 1 < C::NTILES) commit(buf ^ 1, t + 1);        // stage buf^1 was last read during tile t-1: every wave passed that barrier
    SF_T(8)
    __syncthreads();
    SF_T(9)
  }
#if SF_EXP == 9
  SF_T(11)
  if (wave == 0 && lane == 0 && p.lse) {
    for (int k = 0; k < 12; ++k) p.lse[blockIdx.x * 16 + k] = (float)sf_tacc[k];
  }
  return;
#endif
  // ---- epilogue: lane = query, registers = output columns sf_acc_row(r, h) + 32 j (columns >= 80 are padding)
  const float l = __shfl(oacc[2][8], fr, 64);   // row 80 of O^T = block 2, row 16: register 8 of the lower half-wave's lane
  const float mult = 1.f / l;
  if (p.lse && qi < C::NWIN && h == 0) p.lse[stat_row] = m * c + __builtin_amdgcn_logf(l);   // log2 domain: P = exp2(c s - lse)
  bool inside = false;
  const long row = tok_row(qc, inside);
  if (SF_EXP != 7)   // (the loop's last barrier has passed: the K/V ring is free for the wave-private output images)
    sf_store_rows(smem + wave * SF_OIMG, oacc, mult, (qi < C::NWIN && inside) ? (int)row : -1, p.out + head * SF_D, p.ldo, lane);
}

template <int S>
static int launch_sam_flash(const SamFlashP& p, hipStream_t s) {
  using C = SamFlashCfg<S>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_sam_flash_fwd<S>, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    attr = true;
  }
  const long blocks = (long)p.nimg * p.nws * p.nws * p.H * C::QBLK;
  hipLaunchKernelGGL(k_sam_flash_fwd<S>, dim3((unsigned)blocks), dim3(C::NT), C::SMEM, s, p);
  return 0;
}

static int sam_flash_fwd_impl(const void* qkv, long ld, const float* bias, const void* tbl_h, const void* tbl_w, void* out, long ldo, int nimg,
                              int G, int S, int H, int d, float scale, float* lse, void* qext, void* stream) {
  VFM_CHECK(qkv && tbl_h && tbl_w && out, VFM_E_INVAL, "vfm_sam_attn_flash_fwd: null pointer");
  VFM_CHECK(d == SF_D, VFM_E_UNSUPPORTED, "vfm_sam_attn_flash_fwd: head dim %d (only 80 = SAM ViT-H)", d);
  VFM_CHECK((S == 14 && G > 0) || (S == 32 && G == 32), VFM_E_UNSUPPORTED,
            "vfm_sam_attn_flash_fwd: window %d on a %d-token grid (14 x 14 windows or 32 x 32 global)", S, G);
  VFM_CHECK(ld % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0) &&
                ((uintptr_t)tbl_h & 15) == 0 && ((uintptr_t)tbl_w & 15) == 0,
            VFM_E_ALIGN, "vfm_sam_attn_flash_fwd: alignment");
  if (nimg <= 0) return VFM_OK;
  SamFlashP p;
  p.qkv = (const bf16_t*)qkv, p.ld = ld, p.bias = bias, p.tbl_h = (const bf16_t*)tbl_h, p.tbl_w = (const bf16_t*)tbl_w;
  p.out = (bf16_t*)out, p.ldo = ldo, p.nimg = nimg, p.G = G, p.H = H, p.nws = S == 32 ? 1 : (G + S - 1) / S, p.scale = scale;
  p.lse = lse, p.qext = (bf16_t*)qext;
  if (S == 14) launch_sam_flash<14>(p, (hipStream_t)stream);
  else launch_sam_flash<32>(p, (hipStream_t)stream);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

extern "C" int vfm_sam_attn_flash_fwd(const void* qkv, long ld, const float* bias, const void* tbl_h, const void* tbl_w, void* out,
                                      long ldo, int nimg, int G, int S, int H, int d, float scale, void* stream) {
  return sam_flash_fwd_impl(qkv, ld, bias, tbl_h, tbl_w, out, ldo, nimg, G, S, H, d, scale, nullptr, nullptr, stream);
}

extern "C" int vfm_sam_attn_flash_fwd_train(const void* qkv, long ld, const float* bias, const void* tbl_h, const void* tbl_w, void* out,
                                            long ldo, int nimg, int G, int S, int H, int d, float scale, float* lse, void* qext, void* stream) {
  VFM_CHECK(lse && qext && ((uintptr_t)qext & 15) == 0, VFM_E_INVAL, "vfm_sam_attn_flash_fwd_train: lse / qext");
  return sam_flash_fwd_impl(qkv, ld, bias, tbl_h, tbl_w, out, ldo, nimg, G, S, H, d, scale, lse, qext, stream);
}
