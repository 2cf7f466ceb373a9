// Flash-style SAM attention backward for gfx950 (head dim 80, windowed 14 x 14 and global 32 x 32, decomposed relative-position
// bias with FROZEN tables): d(attention output) -> d(qkv), straight from / to the token-major matrices, no score matrix in memory.
// Reference: the autograd of rein/models/backbones/sam_vit.py:273-298 (Attention.forward), :301-356 (window partition /
// unpartition: padded window tokens are keys whose gradient is dropped, padded queries have no output), :392-430
// (add_decomposed_rel_pos).  Replaces, for bf16 training, the materialised form of sam.hip (prep -> dP GEMM -> softmax backward ->
// three batched GEMMs -> prep again -> dQaug GEMM -> merge).
//
//   s[q, k] = scale q.k + Bh[q, kh] + Bw[q, kw],  Bh[q, kh] = q . tbl_h[qh - kh + S - 1];  p = softmax_k(s);  o = p v
//   dv = p^T do;  dp = do v^T;  ds = p * (dp - D), D[q] = do[q] . o[q];
//   dk = scale ds^T q;  dq = scale ds k + sum_kh dBh[q, kh] tbl_h[qh - kh + S - 1] + (same for w),  dBh[q, kh] = sum_{k in row kh} ds[q, k]
//
// The training forward (sam_flash.hip, vfm_sam_attn_flash_fwd_train) leaves lse[q] (log2 domain) and the query operand's bias columns
// qext[q] = [Bh[q, :] / scale | Bw[q, :] / scale] (bf16, the values its score product used).  Two kernels, both shaped like the forward:
// * k_sam_flash_dq: a wave owns 32 queries on the MFMA lane index and walks the key tiles [k | onehot(kh) | onehot(kw)] + v:
//   s^T and dp^T by row fragments, dQext^T += Kext^T ds^T by transposing LDS reads.  Rows 80.. of dQext^T ARE dBh^T / dBw^T (the
//   one-hot columns sum ds over a key row / column); they go through a wave-private LDS image into one more small MFMA product
//   with the table images.  Also writes D[q] for the second kernel.
// * k_sam_flash_dkv: a wave owns 32 keys on the lane index (operand [k | onehot] and v in registers) and walks the query tiles
//   [q | qext] + do (+ lse, D): s and dp by row fragments (rows = queries), dV^T += dO^T p, dK^T += Q^T ds by transposing reads.
#include "sam_flash_dev.h"

struct SamFlashBwdP {
  const bf16_t* qkv; long ld;
  const float* bias;
  const bf16_t* tbl_h; const bf16_t* tbl_w;
  const bf16_t* out; const bf16_t* dout; long ldo;   // token-major [nimg*G*G, H*80]
  const float* lse; const bf16_t* qext;              // from the training forward
  float* dsum;                                        // [nwh, NWINP] scratch: D[q]
  bf16_t* dqkv; long ldg;                             // token-major [nimg*G*G, 3*H*80]
  int nimg, G, H, nws;
  float scale;
};

template <int S>
struct SfBlock {   // block -> (image, window, head, sub-block) with the forward's XCD-aware order, and the window geometry
  using C = SamFlashCfg<S>;
  int sub, head, wx, wy, img, G, H;
  __device__ SfBlock(int bid, int nimg, int nws, int H_, int G_) : G(G_), H(H_) {
    const int inner = H_ * C::QBLK, ngroups = nimg * nws * nws, full = ngroups & ~7;
    if (bid < full * inner) {
      const int xcd = bid & 7, idx = bid >> 3;
      bid = ((idx / inner) * 8 + xcd) * inner + idx % inner;
    }
    sub = bid % C::QBLK; bid /= C::QBLK;
    head = bid % H_; bid /= H_;
    wx = bid % nws; bid /= nws;
    wy = bid % nws;
    img = bid / nws;
  }
  __device__ long wh(int nws) const { return (((long)img * nws + wy) * nws + wx) * H + head; }
  __device__ long tok_row(int t, bool& inside) const {
    const int ty = t / S, tx = t - ty * S;
    const int gy = wy * S + ty, gx = wx * S + tx;
    inside = gy < G && gx < G;
    return ((long)img * G + gy) * G + gx;
  }
};

__device__ __forceinline__ float sf_dot8(uint4 a, uint4 b) {
  const uint32_t* x = reinterpret_cast<const uint32_t*>(&a);
  const uint32_t* y = reinterpret_cast<const uint32_t*>(&b);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s += h16_lo(x[i]) * h16_lo(y[i]);
    s += h16_hi(x[i]) * h16_hi(y[i]);
  }
  return s;
}

// ============================================================ dq (+ D)
template <int S>
__global__ void __launch_bounds__(SamFlashCfg<S>::NT, 2) k_sam_flash_dq(SamFlashBwdP p) {
  using C = SamFlashCfg<S>;
  using CB = SamFlashBwdCfg<S>;
  constexpr int NJ = (SF_D + 2 * C::SP + 31) / 32;   // 32-row blocks of dQext^T: 4 (112 rows) / 5 (144 rows)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, h = lane >> 5;
  const SfBlock<S> B(blockIdx.x, p.nimg, p.nws, p.H, p.G);
  const int head = B.head, Cq = p.H * SF_D;
  const long wh = B.wh(p.nws);

  // bias image of the padded keys (k, v), as the forward: loads first, the LDS write after the other prologue loads are in flight
  char* bimg = smem + 2 * CB::TILE;
  float4 bias0 = make_float4(0.f, 0.f, 0.f, 0.f), bias1 = bias0;
  const int bsec = 1 + tid / 10, bc8 = (tid % 10) * 8;
  if (tid < 20 && p.bias) {
    const float* b = p.bias + bsec * Cq + head * SF_D + bc8;
    bias0 = *reinterpret_cast<const float4*>(b), bias1 = *reinterpret_cast<const float4*>(b + 4);
  }

  // ---- K/V tile staging, as the forward (SfKvStager, sam_flash_dev.h)
  const SfGeo geo{B.img, B.wy, B.wx, B.G};
  SfKvStager<S, CB::KS, CB::TILE, true> stager;
  auto fetch = [&](int t) __attribute__((always_inline)) { stager.fetch(p.qkv, p.ld, Cq, head, bimg, geo, t, tid); };
  auto commit = [&](int buf, int t) __attribute__((always_inline)) { stager.commit(smem, buf, t, tid); };
  // the last four-chunk group of the K rows holds two data and two pad chunks (at swizzled places): the transposing loads of the last
  // row block read them all - zero the group once, the stager then rewrites the data chunks only
  for (int i = tid; i < 2 * 64 * 4; i += C::NT) {
    const int buf = i >> 8, row = (i >> 2) & 63, cc = i & 3;
    *reinterpret_cast<uint4*>(smem + buf * CB::TILE + row * CB::KS + (CB::PADK + cc) * 16) = make_uint4(0, 0, 0, 0);
  }
  stager.fetch_first_global(p.qkv, p.ld, Cq, head, geo, tid);   // (tokens inside the image) in flight together with the query's own loads below
  // ---- this lane's query: operand [q | qext], dO fragments, D = dO . O, lse
  const int q0 = B.sub * (C::NW * 32) + wave * 32;
  const int qi = q0 + fr;
  const int qc = qi < C::NWIN ? qi : C::NWIN - 1;
  const long stat = wh * C::NWINP + qc;
  bool q_inside;
  const long q_row = B.tok_row(qc, q_inside);
  bf16x8 qa[C::KSTEPS], da[5];
  float dsum = 0.f;
#pragma unroll
  for (int kk = 0; kk < 5; ++kk) {
    uint4 vq = make_uint4(0, 0, 0, 0), vd = vq, vo = vq;
    if (q_inside) {
      vq = *reinterpret_cast<const uint4*>(p.qkv + q_row * p.ld + head * SF_D + 16 * kk + 8 * h);
      vd = *reinterpret_cast<const uint4*>(p.dout + q_row * p.ldo + head * SF_D + 16 * kk + 8 * h);
      vo = *reinterpret_cast<const uint4*>(p.out + q_row * p.ldo + head * SF_D + 16 * kk + 8 * h);
    }
    qa[kk] = *reinterpret_cast<const bf16x8*>(&vq);
    da[kk] = *reinterpret_cast<const bf16x8*>(&vd);
    dsum += sf_dot8(vd, vo);
  }
#pragma unroll
  for (int e = 0; e < 2 * C::SP / 16; ++e) qa[5 + e] = *reinterpret_cast<const bf16x8*>(p.qext + stat * (2 * C::SP) + 16 * e + 8 * h);
  dsum += __shfl_xor(dsum, 32, 64);
  if (qi < C::NWIN && h == 0) p.dsum[stat] = dsum;
  const float L = p.lse[stat];
  if (tid < 20)
    *reinterpret_cast<uint4*>(bimg + (bsec - 1) * (2 * SF_D) + bc8 * 2) =
        make_uint4(sf_pack2(bias0.x, bias0.y), sf_pack2(bias0.z, bias0.w), sf_pack2(bias1.x, bias1.y), sf_pack2(bias1.z, bias1.w));
  __syncthreads();   // bias image
  stager.fetch_first_bias(bimg, geo, tid);       // the padded tokens of the first tile
  commit(0, 0);
  __syncthreads();

  const float c = p.scale * SF_LOG2E;
  const SfSwzRow swz(fr, h);
  const SfTrLane trk(lane, CB::KS, true);
  const int rk_e = fr * CB::KS + swz.even, rk_o = fr * CB::KS + swz.odd;   // by-row fragments: lane constants for even / odd k-steps
  const int rv_e = fr * CB::VS + swz.even, rv_o = fr * CB::VS + swz.odd;
  f32x16 dq[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) dq[j] = sf_zero();
  const bool active = (C::NWIN % (C::NW * 32) == 0) || q0 < C::NWIN;
  constexpr int LASTK = C::NWIN - 64 * (C::NTILES - 1);
  constexpr int NKB_LAST = (LASTK + 31) / 32;
#pragma unroll 1
  for (int t = 0; t < C::NTILES; ++t) {
    const int buf = t & 1;
    // the next tile: its global loads fly during this tile's products where the staging registers fit beside the accumulators (S = 14);
    // otherwise it is staged whole before the products (the other block of the CU covers the wait)
    constexpr bool OVERLAP = S == 14;
    if (t + 1 < C::NTILES) {
      fetch(t + 1);
      if (!OVERLAP) commit(buf ^ 1, t + 1);   // stage buf^1 was last read during tile t-1: every wave passed that barrier
    }
    const char* kt = smem + buf * CB::TILE;
    const char* vt = kt + 64 * CB::KS;
    if (active) {
      const int nkb = t == C::NTILES - 1 ? NKB_LAST : 2;
#pragma unroll 1
      for (int kb = 0; kb < nkb; ++kb) {
        f32x16 sacc = sf_zero(), dp = sf_zero();
#pragma unroll
        for (int kk = 0; kk < C::KSTEPS; ++kk) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(kt + kb * 32 * CB::KS + ((kk & 1) ? rk_o : rk_e) + 32 * kk);
          sacc = SF_MFMA(a, qa[kk], sacc);
        }
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(vt + kb * 32 * CB::VS + ((kk & 1) ? rv_o : rv_e) + 32 * kk);
          dp = SF_MFMA(a, da[kk], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = __builtin_amdgcn_exp2f(fmaf(sacc[r], c, -L));
        if ((C::NWIN % 64 != 0) && t == C::NTILES - 1) {   // ragged last tile: keys beyond the window do not exist
          const int key0 = t * 64 + kb * 32;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (key0 + sf_acc_row(r, h) >= C::NWIN) sacc[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] *= dp[r] - dsum;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 pb;
#pragma unroll
          for (int e = 0; e < 8; ++e) pb[e] = (vfm_h)sacc[8 * s + e];
#pragma unroll
          for (int j = 0; j < NJ; ++j) dq[j] = SF_MFMA(trk.frag(kt + (kb * 32 + 16 * s) * CB::KS, j), pb, dq[j]);
        }
      }
    }
    if (OVERLAP && t + 1 < C::NTILES) commit(buf ^ 1, t + 1);
    __syncthreads();
  }

  // ---- epilogue: rows 80.. of dQext^T = dBh^T / dBw^T -> wave-private image [2][SP][32 queries] fp32; table images -> LDS
  float* db = reinterpret_cast<float*>(smem) + wave * (2 * C::SP * 32);
  char* timg = smem + 2 * CB::TILE - C::TIMG;
#pragma unroll
  for (int j = 2; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int e0 = 32 * j + (r & 3) + 8 * (r >> 2) - SF_D;   // + 4 h: the bias column of this register
      if (e0 >= 0 && e0 < 2 * C::SP) db[(e0 + 4 * h) * 32 + fr] = dq[j][r];
    }
  {   // (all the loads before the first LDS write: one round trip)
    constexpr int NTP = 2 * C::JP * 10, NTI = (NTP + C::NT - 1) / C::NT;
    uint4 tp[NTI];
#pragma unroll
    for (int i = 0; i < NTI; ++i) {
      const int pc = tid + i * C::NT;
      tp[i] = make_uint4(0, 0, 0, 0);
      if (pc < NTP) {
        const int which = pc / (C::JP * 10), rem = pc - which * (C::JP * 10);
        tp[i] = *reinterpret_cast<const uint4*>((which ? p.tbl_w : p.tbl_h) + rem * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < NTI; ++i) {
      const int pc = tid + i * C::NT;
      if (pc < NTP) {
        const int which = pc / (C::JP * 10), rem = pc - which * (C::JP * 10), row = rem / 10, cc = rem - row * 10;
        *reinterpret_cast<uint4*>(timg + (which * C::JP + row) * C::TS + cc * 16) = tp[i];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[j][r] *= p.scale;
  {
    const int qh = qc / S, qw = qc - qh * S;
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
      for (int js = 0; js < C::JP / 16; ++js) {
        bf16x8 pb;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int jj = 16 * js + 4 * h + (e & 3) + 8 * (e >> 2);       // the k index the transposing read pairs with element e
          const int kx = (which ? qw : qh) - jj + S - 1;                 // key coordinate whose bias used table row jj
          pb[e] = (kx >= 0 && kx < S) ? (vfm_h)db[(which * C::SP + kx) * 32 + fr] : (vfm_h)0.f;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) dq[j] = SF_MFMA(sf_tr_frag(timg + which * C::JP * C::TS, C::TS, 16 * js, j, lane), pb, dq[j]);
      }
  }
  // output image: behind the dB images (S = 14) or over this wave's own dB image (S = 32, read above); never over the table images
  constexpr int IMG0 = S == 14 ? C::NW * 2 * C::SP * 32 * 4 : 0, IMGSZ = S == 14 ? SF_OIMG : 2 * C::SP * 32 * 4;
  static_assert(IMGSZ >= SF_OIMG && IMG0 + C::NW * IMGSZ <= 2 * CB::TILE - C::TIMG, "output images overlap the table images");
  sf_store_rows(smem + IMG0 + wave * IMGSZ, reinterpret_cast<const f32x16(&)[3]>(dq[0]), 1.0f, (qi < C::NWIN && q_inside) ? (int)q_row : -1,
                p.dqkv + head * SF_D, p.ldg, lane);
}

// ============================================================ dk, dv
template <int S>
__global__ void __launch_bounds__(SamFlashCfg<S>::NT, 2) k_sam_flash_dkv(SamFlashBwdP p) {
  using C = SamFlashCfg<S>;
  using CB = SamFlashBwdCfg<S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, h = lane >> 5;
  const SfBlock<S> B(blockIdx.x, p.nimg, p.nws, p.H, p.G);
  const int head = B.head, Cq = p.H * SF_D;
  const long wh = B.wh(p.nws);

  // ---- query tile staging: 64 queries x ([q | qext] rows, dO rows) + lse / D, global -> registers -> LDS; thread (row = tid >> 2,
  // quarter = tid & 3) owns the pieces {quarter, quarter + 4, quarter + 8 (< 10)} of its query row in q and dO and one or two qext pieces
  constexpr int NE = 2 * C::SP / 32;   // qext pieces per thread
  const int srow = tid >> 2, sq = tid & 3;
  const SfGeo geo{B.img, B.wy, B.wx, B.G};
  uint4 stq[3], std_[3], stge[NE], stgs;
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const int tq = t * 64 + srow;
#pragma unroll
    for (int i = 0; i < 3; ++i) stq[i] = std_[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NE; ++i) stge[i] = make_uint4(0, 0, 0, 0);
    if (tq < C::NWIN) {
      bool inside;
      const long grow = B.tok_row(tq, inside);
      if (inside) {   // a padded query has no output: q does not matter and dO = 0
        const bf16_t* sq_ = p.qkv + grow * p.ld + head * SF_D + sq * 8;
        const bf16_t* sd_ = p.dout + grow * p.ldo + head * SF_D + sq * 8;
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i < 2 || sq < 2) {
            stq[i] = *reinterpret_cast<const uint4*>(sq_ + 32 * i);
            std_[i] = *reinterpret_cast<const uint4*>(sd_ + 32 * i);
          }
      }
      const bf16_t* se_ = p.qext + (wh * C::NWINP + tq) * (2 * C::SP) + sq * 8;
#pragma unroll
      for (int i = 0; i < NE; ++i) stge[i] = *reinterpret_cast<const uint4*>(se_ + 32 * i);
    }
    {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (tid < 32) {   // pieces 0..15: lse (rows beyond the window: +inf, so p = 0), 16..31: D (0 there)
        const int which = tid >> 4, q4 = t * 64 + 4 * (tid & 15);
        const float* src = (which ? p.dsum : p.lse) + wh * C::NWINP + q4;
        const float fill = which ? 0.f : INFINITY;
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = q4 + e < C::NWIN ? src[e] : fill;
        v = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
      }
      stgs = v;
    }
  };
  auto commit = [&](int buf) __attribute__((always_inline)) {
    const int x = (srow >> 2) & 3;   // chunk swizzle (SamFlashBwdCfg)
    char* qt = smem + buf * CB::TILE_DKV + srow * CB::KS;
    char* dt = smem + buf * CB::TILE_DKV + 64 * CB::KS + srow * CB::VS;
    char* st = smem + buf * CB::TILE_DKV + 64 * (CB::KS + CB::VS);
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < 2 || sq < 2) {
        *reinterpret_cast<uint4*>(qt + ((sq + 4 * i) ^ x) * 16) = stq[i];
        *reinterpret_cast<uint4*>(dt + ((sq + 4 * i) ^ x) * 16) = std_[i];
      }
#pragma unroll
    for (int i = 0; i < NE; ++i) *reinterpret_cast<uint4*>(qt + ((10 + sq + 4 * i) ^ x) * 16) = stge[i];
    if (tid < 32) *reinterpret_cast<uint4*>(st + tid * 16) = stgs;
  };
  // the four-chunk groups that hold data AND pad chunks (at swizzled places) are read whole by the transposing loads: zero them once
  // ([q | qext] rows: group PADK; dO rows: chunks 8..11), the stager then rewrites the data chunks only
  for (int i = tid; i < 2 * 64 * 4; i += C::NT) {
    const int buf = i >> 8, row = (i >> 2) & 63, cc = i & 3;
    *reinterpret_cast<uint4*>(smem + buf * CB::TILE_DKV + row * CB::KS + (CB::PADK + cc) * 16) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(smem + buf * CB::TILE_DKV + 64 * CB::KS + row * CB::VS + (8 + cc) * 16) = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();   // (these zeros and the first commit touch the same chunks from different threads)
  fetch(0);   // in flight together with the key's own loads below
  // ---- this lane's key: operands [k | onehot(kh) | onehot(kw)] and v
  const int k0 = B.sub * (C::NW * 32) + wave * 32;
  const int ki = k0 + fr;
  const int kc = ki < C::NWIN ? ki : C::NWIN - 1;
  bool k_inside;
  const long k_row = B.tok_row(kc, k_inside);
  bf16x8 kq[5], vb[5];
#pragma unroll
  for (int kk = 0; kk < 5; ++kk) {
    uint4 vk = make_uint4(0, 0, 0, 0), vv = vk;
    const int c8 = 16 * kk + 8 * h;
    if (k_inside) {
      vk = *reinterpret_cast<const uint4*>(p.qkv + k_row * p.ld + Cq + head * SF_D + c8);
      vv = *reinterpret_cast<const uint4*>(p.qkv + k_row * p.ld + 2 * Cq + head * SF_D + c8);
    } else if (p.bias) {
      const float* b = p.bias + Cq + head * SF_D + c8;
      const float4 a0 = *reinterpret_cast<const float4*>(b), a1 = *reinterpret_cast<const float4*>(b + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(b + Cq), b1 = *reinterpret_cast<const float4*>(b + Cq + 4);
      vk = make_uint4(sf_pack2(a0.x, a0.y), sf_pack2(a0.z, a0.w), sf_pack2(a1.x, a1.y), sf_pack2(a1.z, a1.w));
      vv = make_uint4(sf_pack2(b0.x, b0.y), sf_pack2(b0.z, b0.w), sf_pack2(b1.x, b1.y), sf_pack2(b1.z, b1.w));
    }
    kq[kk] = *reinterpret_cast<const bf16x8*>(&vk);
    vb[kk] = *reinterpret_cast<const bf16x8*>(&vv);
  }
  // the one-hot columns of the key operand are rebuilt from two integers per axis at each use (registers are the scarce resource here):
  // dword d of k-step ks holds bf16 1.0 in its low / high half iff (coordinate - 8 h) >> 1 == 8 ks + d
  int oh_idx[2];
  uint32_t oh_val[2];
  {
    const int kh = kc / S, kw = kc - kh * S;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const int slot = (which ? kw : kh) - 8 * h;
      oh_idx[which] = slot >> 1;
      oh_val[which] = (slot & 1) ? (VFM_H_ONE << 16) : VFM_H_ONE;
    }
  }
  auto onehot = [&](int which, int ks) __attribute__((always_inline)) -> bf16x8 {
    uint32_t w[4];
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) w[dd] = oh_idx[which] == 8 * ks + dd ? oh_val[which] : 0u;
    const uint4 u = make_uint4(w[0], w[1], w[2], w[3]);
    return *reinterpret_cast<const bf16x8*>(&u);
  };
  commit(0);
  __syncthreads();

  const float c = p.scale * SF_LOG2E;
  const SfSwzRow swz(fr, h);
  const SfTrLane trq(lane, CB::KS, true), trd(lane, CB::VS, true);
  const int rq_e = fr * CB::KS + swz.even, rq_o = fr * CB::KS + swz.odd;
  const int rd_e = fr * CB::VS + swz.even, rd_o = fr * CB::VS + swz.odd;
  f32x16 dk[3] = {sf_zero(), sf_zero(), sf_zero()}, dv[3] = {sf_zero(), sf_zero(), sf_zero()};
  const bool active = (C::NWIN % (C::NW * 32) == 0) || k0 < C::NWIN;
  constexpr int LASTQ = C::NWIN - 64 * (C::NTILES - 1);
  constexpr int NQB_LAST = (LASTQ + 31) / 32;
#pragma unroll 1
  for (int t = 0; t < C::NTILES; ++t) {
    const int buf = t & 1;
    if (t + 1 < C::NTILES) {   // staged whole before the products: dk + dv + the key operands leave no room for live staging registers
      fetch(t + 1);
      commit(buf ^ 1);
    }
    const char* qt = smem + buf * CB::TILE_DKV;
    const char* dt = qt + 64 * CB::KS;
    const char* st = dt + 64 * CB::VS;
    if (active) {
      const int nqb = t == C::NTILES - 1 ? NQB_LAST : 2;
#pragma unroll 1
      for (int qb = 0; qb < nqb; ++qb) {
        f32x16 sacc = sf_zero(), dp = sf_zero();   // rows = queries of this 32-block, lanes = keys
#pragma unroll
        for (int kk = 0; kk < C::KSTEPS; ++kk) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(qt + qb * 32 * CB::KS + ((kk & 1) ? rq_o : rq_e) + 32 * kk);
          sacc = SF_MFMA(a, kk < 5 ? kq[kk] : onehot((kk - 5) / (C::SP / 16), (kk - 5) % (C::SP / 16)), sacc);
        }
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(dt + qb * 32 * CB::VS + ((kk & 1) ? rd_o : rd_e) + 32 * kk);
          dp = SF_MFMA(a, vb[kk], dp);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // registers 4 i .. 4 i + 3 = query rows 8 i + 4 h + 0..3 of the block
          const float4 Lr = *reinterpret_cast<const float4*>(st + (qb * 32 + 8 * i + 4 * h) * 4);
          const float4 Dr = *reinterpret_cast<const float4*>(st + 256 + (qb * 32 + 8 * i + 4 * h) * 4);
          const float lr[4] = {Lr.x, Lr.y, Lr.z, Lr.w}, dr[4] = {Dr.x, Dr.y, Dr.z, Dr.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[4 * i + e], c, -lr[e]));
            sacc[4 * i + e] = pv;
            dp[4 * i + e] = pv * (dp[4 * i + e] - dr[e]);
          }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 pp_, ps_;
#pragma unroll
          for (int e = 0; e < 8; ++e) pp_[e] = (vfm_h)sacc[8 * s + e], ps_[e] = (vfm_h)dp[8 * s + e];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            dv[j] = SF_MFMA(trd.frag(dt + (qb * 32 + 16 * s) * CB::VS, j), pp_, dv[j]);
            dk[j] = SF_MFMA(trq.frag(qt + (qb * 32 + 16 * s) * CB::KS, j), ps_, dk[j]);
          }
        }
      }
    }
    __syncthreads();
  }

  const int k_dst = (ki < C::NWIN && k_inside) ? (int)k_row : -1;
  sf_store_rows(smem + wave * SF_OIMG, dk, p.scale, k_dst, p.dqkv + Cq + head * SF_D, p.ldg, lane);
  sf_store_rows(smem + wave * SF_OIMG, dv, 1.0f, k_dst, p.dqkv + 2 * Cq + head * SF_D, p.ldg, lane);
}

template <int S>
static void launch_sam_flash_bwd(const SamFlashBwdP& p, hipStream_t s) {
  using C = SamFlashCfg<S>;
  using CB = SamFlashBwdCfg<S>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_sam_flash_dq<S>, hipFuncAttributeMaxDynamicSharedMemorySize, CB::SMEM_DQ);
    (void)hipFuncSetAttribute((const void*)k_sam_flash_dkv<S>, hipFuncAttributeMaxDynamicSharedMemorySize, CB::SMEM_DKV);
    attr = true;
  }
  const long blocks = (long)p.nimg * p.nws * p.nws * p.H * C::QBLK;
  hipLaunchKernelGGL(k_sam_flash_dq<S>, dim3((unsigned)blocks), dim3(C::NT), CB::SMEM_DQ, s, p);    // writes D, which dkv reads
  hipLaunchKernelGGL(k_sam_flash_dkv<S>, dim3((unsigned)blocks), dim3(C::NT), CB::SMEM_DKV, s, p);
}

extern "C" long vfm_sam_attn_flash_stat_rows(int nimg, int G, int S, int H) {
  const int nws = S == 32 ? 1 : (G + S - 1) / S;
  return (long)nimg * nws * nws * H * (S == 14 ? SamFlashCfg<14>::NWINP : SamFlashCfg<32>::NWINP);
}

extern "C" int vfm_sam_attn_flash_bwd(const void* qkv, long ld, const float* bias, const void* tbl_h, const void* tbl_w, const void* out,
                                      const void* dout, long ldo, const float* lse, const void* qext, float* dsum, void* dqkv, long ldg,
                                      int nimg, int G, int S, int H, int d, float scale, void* stream) {
  VFM_CHECK(qkv && tbl_h && tbl_w && out && dout && lse && qext && dsum && dqkv, VFM_E_INVAL, "vfm_sam_attn_flash_bwd: null pointer");
  VFM_CHECK(d == SF_D, VFM_E_UNSUPPORTED, "vfm_sam_attn_flash_bwd: head dim %d (only 80 = SAM ViT-H)", d);
  VFM_CHECK((S == 14 && G > 0) || (S == 32 && G == 32), VFM_E_UNSUPPORTED,
            "vfm_sam_attn_flash_bwd: window %d on a %d-token grid (14 x 14 windows or 32 x 32 global)", S, G);
  VFM_CHECK(ld % 8 == 0 && ldo % 8 == 0 && ldg % 8 == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)dout & 15) == 0 &&
                ((uintptr_t)dqkv & 15) == 0 && ((uintptr_t)qext & 15) == 0 && ((uintptr_t)lse & 15) == 0 && ((uintptr_t)dsum & 15) == 0 &&
                (!bias || ((uintptr_t)bias & 15) == 0) && ((uintptr_t)tbl_h & 15) == 0 && ((uintptr_t)tbl_w & 15) == 0,
            VFM_E_ALIGN, "vfm_sam_attn_flash_bwd: alignment");
  if (nimg <= 0) return VFM_OK;
  SamFlashBwdP p;
  p.qkv = (const bf16_t*)qkv, p.ld = ld, p.bias = bias, p.tbl_h = (const bf16_t*)tbl_h, p.tbl_w = (const bf16_t*)tbl_w;
  p.out = (const bf16_t*)out, p.dout = (const bf16_t*)dout, p.ldo = ldo, p.lse = lse, p.qext = (const bf16_t*)qext, p.dsum = dsum;
  p.dqkv = (bf16_t*)dqkv, p.ldg = ldg, p.nimg = nimg, p.G = G, p.H = H, p.nws = S == 32 ? 1 : (G + S - 1) / S, p.scale = scale;
  if (S == 14) launch_sam_flash_bwd<14>(p, (hipStream_t)stream);
  else launch_sam_flash_bwd<32>(p, (hipStream_t)stream);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
