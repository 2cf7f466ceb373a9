// Shared declarations of the SAM flash attention kernels (sam_flash.hip forward, sam_flash_bwd.hip backward).
#pragma once
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef vfm_h bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define SF_D 80
#ifndef SF_EXP
#define SF_EXP 0   // timing experiments (tools/scratch/_sam_flash_exp.sh): 1 no restaging, 2 no exp, 3 no P V, 4 no Q K, 5 prologue only
#endif
#define SF_LOG2E 1.4426950408889634f
#define SF_MFMA(a, b, c) VFM_MFMA16(a, b, c)

struct SamFlashP {
  const bf16_t* qkv; long ld;        // token-major [nimg*G*G, 3*H*80]
  const float* bias;                 // [3*H*80] projection bias (qkv of the padded window tokens), may be null
  const bf16_t* tbl_h; const bf16_t* tbl_w;  // [JP, 80] relative-index tables (rows >= 2S-1 zero)
  bf16_t* out; long ldo;             // token-major [nimg*G*G, H*80]
  int nimg, G, H, nws;               // nws windows per side (1 for global)
  float scale;
  float* lse; bf16_t* qext;          // training only (else null): per (window, head, query) log2-sum-exp of the scaled scores
                                     // [nwh, NWINP] and the query operand's bias columns [nwh, NWINP, 2 SP] for the backward
};

template <int S>
struct SamFlashCfg {
  static constexpr int NW = 4;                         // waves per block: 128 queries (two query blocks per 14 x 14 window)
  static constexpr int NT = NW * 64;
  static constexpr int NWIN = S * S;                   // tokens per window
  static constexpr int QBLK = (NWIN + NW * 32 - 1) / (NW * 32);
  static constexpr int SP = S <= 16 ? 16 : 32;         // one-hot columns per axis
  static constexpr int JP = 2 * SP;                    // padded rows of the relative-index tables (>= 2S-1)
  static constexpr int KSTEPS = (SF_D + 2 * SP) / 16;  // 7 / 9 k-steps of the extended score product
  static constexpr int KS = S == 14 ? 240 : 304;       // bytes per row of the K tile (160 + 4 SP, padded so that 16 rows hit 16 slots)
  static constexpr int VS = 192;                       // bytes per row of the V tile (96 columns, 80..95 zero)
  static constexpr int TS = 176;                       // bytes per row of the table image
  static constexpr int TILE = 64 * (KS + VS);
  static constexpr int NTILES = (NWIN + 63) / 64;
  static constexpr int NWINP = QBLK * NW * 32;         // 256 / 1024: row pitch of the per-(window, head) statistics
  static constexpr int TIMG = 2 * JP * TS;             // the two table images, parked at the END of stage 1 during the prologue
  static constexpr int TH_BYTES = NW * 2 * JP * 32 * 2;  // per-wave T_h^T / T_w^T images [JP][32 queries] bf16, from byte 0
  static constexpr int SMEM = 2 * TILE + 4 * SF_D;     // both prologue images alias the K/V ring (two blocks per CU) + the bias image
  static_assert(TH_BYTES + TIMG <= 2 * TILE, "prologue images overlap");
};

__device__ __forceinline__ int sf_acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ f32x16 sf_zero() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ uint32_t sf_pack2(float a, float b) {
  typedef vfm_h b2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const b2 v = __builtin_convertvector(f2{a, b}, b2);
  return *reinterpret_cast<const uint32_t*>(&v);
}


// A operand (rows = stored columns 32 j .., k = stored rows of 16-row step `r16`) by transposing LDS reads, as the forward's P V product
// SWZ: the tile stores chunk c (16 bytes) of row r at chunk c ^ ((r >> 2) & 3) (backward tiles, see SamFlashBwdCfg); r16 is a multiple
// of 16, so the four rows of a read share one XOR value: hh for the low half, hh + 2 for the rows eight further on.
template <bool SWZ = false>
__device__ __forceinline__ bf16x8 sf_tr_frag(const char* base, int stride, int r16, int j, int lane) {
  const int g = lane >> 4, i = lane & 15, q4 = i >> 2, pp = i & 3, hh = g >> 1;
  const int chunk = 4 * j + 2 * (g & 1) + (pp >> 1);
  const int r0 = r16 + 4 * hh + q4;
  const int c_lo = SWZ ? chunk ^ hh : chunk, c_hi = SWZ ? chunk ^ (hh + 2) : chunk;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(base + r0 * stride + c_lo * 16 + ((pp & 1) << 3)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(base + (r0 + 8) * stride + c_hi * 16 + ((pp & 1) << 3)));
  union { struct { s16x4 a, b; } st; bf16x8 v; } u;
  u.st.a = lo;
  u.st.b = hi;
  return u.v;
}

// ---- LDS layout of the BACKWARD tiles.  There every tile is read BOTH by rows (ds_read_b128: 16 rows per LDS pass, wants the rows'
// 16-byte chunks in 16 different bank groups) and transposed (ds_read_b64_tr_b16: 4 rows x 64 bytes per pass, wants the rows' 64-byte
// windows in different quarters of the 64 banks).  No linear pitch serves both (the forward's 240 / 304-byte K rows conflict 4-way when
// read transposed, the 192-byte V rows 4-way when read by rows: 50 % LDS bank-conflict cycles, profiles/r02_pmc_sam_flash.txt), so the
// backward uses pitches of 16 or 48 dwords mod 64 (transposed reads conflict-free) and stores chunk c of row r at c ^ ((r >> 2) & 3)
// (rows r, r + 4, r + 8, r + 12 - which share a bank group at these pitches - then differ in the chunk's low bits).
template <int S>
struct SamFlashBwdCfg {
  using C = SamFlashCfg<S>;
  static constexpr int KS = 320, VS = 192;                 // 80 / 48 dwords
  static constexpr int NCK = (SF_D + 2 * C::SP) / 8;        // data chunks of a [k | onehot] / [q | qext] row: 14 / 18
  static constexpr int PADK = NCK / 4 * 4;                  // their last four-chunk group also holds two pad chunks: kept zero
  static constexpr int TILE = 64 * (KS + VS);
  static constexpr int SMEM_DQ = 2 * TILE + 4 * SF_D;       // + the bias image
  static constexpr int TILE_DKV = TILE + 512;               // + lse[64], D[64]
  static constexpr int SMEM_DKV = 2 * TILE_DKV;
  static_assert(PADK + 4 <= KS / 16 && C::NW * 2 * C::SP * 32 * 4 + C::TIMG <= 2 * TILE, "row pitch / epilogue images");
};
// by-row fragment of a swizzled tile: chunk 2 kk + h of row (.. + fr) sits at (2 kk + h) ^ x, x = (fr >> 2) & 3; with b0 = x & 1, b1 = x >> 1
// that is byte 32 (kk ^ b1) + 16 (h ^ b0): two lane constants (for even and for odd kk) and an immediate
struct SfSwzRow {
  int even, odd;   // byte offsets inside the row for k-step 0 / relative to 32 kk
  __device__ __forceinline__ SfSwzRow(int fr, int h) {
    const int x = (fr >> 2) & 3, b0 = x & 1, b1 = x >> 1;
    even = 16 * (h ^ b0) + 32 * b1;
    odd = 16 * (h ^ b0) - 32 * b1;
  }
  __device__ __forceinline__ int at(int kk) const { return ((kk & 1) ? odd : even) + 32 * kk; }
};

// lane constants of the transposing fragment reads (sf_tr_frag's address arithmetic, done once per kernel: these kernels are bound by
// VALU issue, and the per-call form cost ~8 VALU instructions per fragment): fragment (16-row step at byte `step`, column block j) =
// the two reads at step + 64 j + lo / + hi
struct SfTrLane {
  int lo, hi;
  __device__ __forceinline__ SfTrLane(int lane, int stride, bool swz) {
    const int g = lane >> 4, i = lane & 15, q4 = i >> 2, pp = i & 3, hh = g >> 1;
    const int cl = 2 * (g & 1) + (pp >> 1);
    lo = (4 * hh + q4) * stride + ((swz ? cl ^ hh : cl) * 16) + ((pp & 1) << 3);
    hi = (4 * hh + q4 + 8) * stride + ((swz ? cl ^ (hh + 2) : cl) * 16) + ((pp & 1) << 3);
  }
  __device__ __forceinline__ bf16x8 frag(const char* step, int j) const {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(step + lo + 64 * j));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(step + hi + 64 * j));
    union { struct { s16x4 a, b; } st; bf16x8 v; } u;
    u.st.a = a;
    u.st.b = b;
    return u.v;
  }
};

// ---- window geometry and the K/V tile stager shared by the forward and the dq kernel
struct SfGeo {
  int img, wy, wx, G;
  template <int S>
  __device__ __forceinline__ long tok_row(int t, bool& inside) const {   // window token index -> row of the token-major matrices
    if constexpr (S == 32) {   // global attention: the window IS the 32 x 32 grid (the launcher checks G == 32)
      inside = true;
      return (long)img * (S * S) + t;
    }
    const int ty = t / S, tx = t - ty * S;
    const int gy = wy * S + ty, gx = wx * S + tx;
    inside = gy < G && gx < G;
    return ((long)img * G + gy) * G + gx;
  }
};

// One K/V tile = 64 keys x ([k | onehot(kh) | onehot(kw)] rows of KS bytes, v rows of VS bytes), global -> registers -> LDS.
// Thread (row = tid >> 2, quarter = tid & 3) of the 256-thread block owns the 16-byte pieces {quarter, quarter + 4, quarter + 8 (< 10)}
// of ITS key row in k and in v: one token-row computation and one address per tile and thread, the pieces at immediate offsets
// (a piece-major split costs ~30 VALU instructions of index arithmetic per piece, which made the staging ~60 % of the loop's VALU work).
// A token outside the image has k / v = the projection bias, read from the block's packed image in LDS (bimg: [2][80] bf16).
template <int S, int KSTR = SamFlashCfg<S>::KS, int TILEB = SamFlashCfg<S>::TILE, bool SWZ = false>
struct SfKvStager {
  using C = SamFlashCfg<S>;
  static_assert(C::NT == 256, "row / quarter split of 256 threads");
  uint4 k[3], v[3];
  // A key beyond the window (ragged last tile) is staged as a copy of the last one: its probability is forced to 0 by the kernels, all
  // it has to be is finite - so there is no zero fill and no validity branch in the loop.
  __device__ __forceinline__ long locate(const SfGeo& g, int t, int tid, bool& inside) const {
    const int row = tid >> 2;
    const int key = (C::NWIN % 64 == 0) ? t * 64 + row : min(t * 64 + row, C::NWIN - 1);
    return g.tok_row<S>(key, inside);
  }
  __device__ __forceinline__ void load_global(const bf16_t* src, int Cq, int sq) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int off = i < 2 ? 32 * i : (min(sq + 8, 9) - sq) * 8;   // quarters 2, 3 own two pieces: their third register re-reads piece 9 (never committed)
      k[i] = *reinterpret_cast<const uint4*>(src + off);
      v[i] = *reinterpret_cast<const uint4*>(src + Cq + off);
    }
  }
  __device__ __forceinline__ void load_bias(const char* bimg, int sq) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = i < 2 ? sq + 4 * i : min(sq + 8, 9);
      k[i] = *reinterpret_cast<const uint4*>(bimg + c * 16);
      v[i] = *reinterpret_cast<const uint4*>(bimg + 2 * SF_D + c * 16);
    }
  }
  // whole tile: global loads for the tokens inside the image, the bias rows (LDS image) for the padded ones
  __device__ __forceinline__ void fetch(const bf16_t* qkv, long ld, int Cq, int head, const char* bimg, const SfGeo& g, int t, int tid) {
    bool inside;
    const long tr = locate(g, t, tid, inside);
    const int sq = tid & 3;
    if (inside) load_global(qkv + tr * ld + Cq + head * SF_D + sq * 8, Cq, sq);
    else load_bias(bimg, sq);
  }
  // the first tile in two parts: the global loads before the barrier that publishes the bias image, the bias rows after it
  __device__ __forceinline__ void fetch_first_global(const bf16_t* qkv, long ld, int Cq, int head, const SfGeo& g, int tid) {
    bool inside;
    const long tr = locate(g, 0, tid, inside);
    const int sq = tid & 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) k[i] = v[i] = make_uint4(0, 0, 0, 0);
    if (inside) load_global(qkv + tr * ld + Cq + head * SF_D + sq * 8, Cq, sq);
  }
  __device__ __forceinline__ void fetch_first_bias(const char* bimg, const SfGeo& g, int tid) {
    bool inside;
    (void)locate(g, 0, tid, inside);
    if (!inside) load_bias(bimg, tid & 3);
  }
  __device__ __forceinline__ void commit(char* smem, int buf, int t, int tid) const {
    const int row = tid >> 2, sq = tid & 3, key = t * 64 + row;
    const int x = SWZ ? (row >> 2) & 3 : 0;   // chunk swizzle of the backward tiles (SamFlashBwdCfg)
    char* kt = smem + buf * TILEB + row * KSTR;
    char* vt = smem + buf * TILEB + 64 * KSTR + row * C::VS;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < 2 || sq < 2) {
        *reinterpret_cast<uint4*>(kt + ((sq + 4 * i) ^ x) * 16) = k[i];
        *reinterpret_cast<uint4*>(vt + ((sq + 4 * i) ^ x) * 16) = v[i];
      }
    // the one-hot pieces (2 SP / 8 per row: one or two per thread) come from the key index alone
    const int kh = key / S, kw = key - kh * S;
#pragma unroll
    for (int i = 0; i < 2 * C::SP / 32; ++i) {
      const int c = sq + 4 * i;
      uint4 o = make_uint4(0, 0, 0, 0);
      const int want = (c < C::SP / 8 ? kh : kw + C::SP) - 8 * c;   // position of the 1 inside this piece, if 0..7
      if (want >= 0 && want < 8) {   // (a key beyond the window gets some row's columns: it is masked anyway)
        const uint32_t val = (want & 1) ? (VFM_H_ONE << 16) : VFM_H_ONE;
        const int wi = want >> 1;
        o = make_uint4(wi == 0 ? val : 0u, wi == 1 ? val : 0u, wi == 2 ? val : 0u, wi == 3 ? val : 0u);
      }
      *reinterpret_cast<uint4*>(kt + ((10 + c) ^ x) * 16) = o;
    }
  }
};

// ---- output rows through LDS.  The accumulators hold O^T (lane = row of the output, registers = columns): a direct store is one
// 8-byte piece per lane at a row pitch of kilobytes - 64 partial cache lines per instruction, measured at 25 % of the forward's run
// time.  Instead the wave writes its 32 x 80 block into a private LDS image (rows of SF_OROW bytes, the row's destination index in
// the pad) and stores it back as 16-byte pieces, ten consecutive lanes per 160-byte row.
#define SF_OROW 176
#define SF_OIMG (32 * SF_OROW)
__device__ __forceinline__ void sf_store_rows(char* img, const f32x16 (&acc)[3], float mult, int dst_row /* -1: no output */, bf16_t* dst,
                                              long ld, int lane) {
  const int fr = lane & 31, h = lane >> 5;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int col = 32 * j + 8 * rq + 4 * h;     // four consecutive columns: registers 4 rq .. 4 rq + 3
      if (col < SF_D)
        *reinterpret_cast<uint2*>(img + fr * SF_OROW + col * 2) =
            make_uint2(sf_pack2(acc[j][4 * rq] * mult, acc[j][4 * rq + 1] * mult), sf_pack2(acc[j][4 * rq + 2] * mult, acc[j][4 * rq + 3] * mult));
    }
  if (h == 0) *reinterpret_cast<int*>(img + fr * SF_OROW + 2 * SF_D) = dst_row;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int pc = lane + 64 * i, r = pc / 10, c = pc - r * 10;
    const int rr = *reinterpret_cast<const int*>(img + r * SF_OROW + 2 * SF_D);
    const uint4 v = *reinterpret_cast<const uint4*>(img + r * SF_OROW + c * 16);
    if (rr >= 0) *reinterpret_cast<uint4*>(dst + (long)rr * ld + c * 8) = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();   // the image may be rewritten (dk then dv)
}
