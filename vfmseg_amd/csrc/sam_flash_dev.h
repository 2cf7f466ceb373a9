// Shared declarations of the SAM flash attention kernels (sam_flash.hip forward, sam_flash_bwd.hip backward).
#pragma once
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define SF_D 80
#ifndef SF_EXP
#define SF_EXP 0   // timing experiments (tools/scratch/_sam_flash_exp.sh): 1 no restaging, 2 no exp, 3 no P V, 4 no Q K, 5 prologue only
#endif
#define SF_LOG2E 1.4426950408889634f
#define SF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

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
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const b2 v = __builtin_convertvector(f2{a, b}, b2);
  return *reinterpret_cast<const uint32_t*>(&v);
}

