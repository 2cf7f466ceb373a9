// Device helpers shared by the bf16 flash-attention kernels (attention_bf16.hip, attention_fwd64.hip): LDS tile staging,
// MFMA fragment reads, the [cls]-row VALU paths.
#pragma once
#include "attn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef vfm_h bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define TROWS 64
#define TILE_BYTES (TROWS * 128)
#define LOG2E 1.4426950408889634f

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}
// The same two LDS-DMA loads issued from inline assembly, for kernels that order them by hand (s_waitcnt vmcnt(N) + s_barrier in front of
// the first read of a tile).  The compiler knows nothing of these: with the builtin it waits for vmcnt(0) in front of every
// ds_read_b64_tr_b16 (it cannot tell the transposed read from the tile being filled) - in the middle of the K loop, i.e. for the tile
// that was requested a few hundred clocks earlier - and the three-stage ring hides nothing.  (Its own waits stay safe: loads return in
// order, so a count that ignores these can only wait longer.)  m0 = LDS byte address of the wave's piece; one wait state before the load.
__device__ __forceinline__ void glds16_raw(const void* g, void* l) {
  const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)l;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(la) : "memory", "m0");
}
__device__ __forceinline__ void glds4_raw(const void* g, void* l) {
  const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)l;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(la) : "memory", "m0");
}
// XOR swizzle of the eight 16-byte chunks of a 128-byte tile row.  Rows are 128 B = all 32 banks apart, so whatever rows one LDS
// cycle touches must land in different chunks:
//   ds_read_b128 row fragments: eight consecutive rows per cycle            -> eight different chunks,
//   ds_read_b64_tr_b16: sixteen lanes = four consecutive rows x 32 bytes    -> four different 32-byte windows (chunk >> 1).
// ((row >> 1) & 7, the first version, put rows r and r+1 in the same chunk and all four rows of a transposed read in the same
// window.  Measured with this one: SQ_LDS_BANK_CONFLICT of the forward 1.05 M -> 0.53 M, dK/dV 2.24 M -> 2.10 M, dQ 1.12 M -> 2.10 M,
// and kernel times unchanged within noise - LDS bank conflicts are not what bounds these kernels; profiles/r02_pmc_attention_*.txt.)
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 1) | ((row >> 2) & 1); }
// Workgroups are handed to the eight XCDs round-robin by linear id, and each XCD has its own L2.  With the plain (block, pair)
// order the blocks of one (image, head) pair land on eight different XCDs and every one of them pulls that pair's whole streamed
// operand (K, V or Q, dO) through the fabric: measured 152-160 MB fetched per attention launch at bs 2 against 25-42 MB of
// operands (profiles/r02_pmc_gemm_traffic.json).  xcd_map sends all nbx blocks of a pair to the same XCD (pairs p with equal
// p % 8 share one) whenever the pair count is a multiple of eight: linear id -> (block bx, pair bh).
__device__ __forceinline__ void xcd_map(int lin, int nbx, int npairs, int on, int& bx, int& bh) {
  if (on && (npairs & 7) == 0) {
    const int xcd = lin & 7, idx = lin >> 3;
    bx = idx % nbx;
    bh = (idx / nbx) * 8 + xcd;
  } else {
    bx = lin % nbx;
    bh = lin / nbx;
  }
}
// combine a value with the one in lane ^ 32 on the VALU (v_permlane32_swap; __shfl_xor(v, 32) is an LDS round trip - ds_bpermute)
__device__ __forceinline__ float half_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);   // r[0]: lanes 0..31 seen by both halves, r[1]: lanes 32..63
}
__device__ __forceinline__ float half_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// stage one [64 x 64] bf16 tile: sequence positions s0..s0+63 of image b (clamped), columns col0..col0+63
template <int NW = 4, bool RAW = false>
__device__ __forceinline__ void stage_tile(const bf16_t* base, long ld, int col0, int b, int s0, int n, int n_main, int B, char* tile,
                                           int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 8 / NW; ++j) {
    const int piece = wave * (8 / NW) + j;
    const int r = piece * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz(r);
    int s = s0 + r;
    if (s > n - 1) s = n - 1;
    if constexpr (RAW) glds16_raw(base + tok_row(b, s, n_main, B) * ld + col0 + c * 8, tile + piece * 1024);
    else glds16(base + tok_row(b, s, n_main, B) * ld + col0 + c * 8, tile + piece * 1024);
  }
}
// Interior tiles (all 64 positions are patch tokens of image b: consecutive rows) need no per-lane address arithmetic at all: the lane's
// byte offsets inside a tile (tile_offsets) are the same for every tile, the tile's first row is a wave-uniform base that stays in
// SGPRs.  The general form above costs 12 VALU instructions per piece and tile - four of them quarter-rate 64-bit multiplies - i.e.
// ~380 clocks per wave and tile, a third of what the softmax itself issues.
template <int NW = 4>
__device__ __forceinline__ void tile_offsets(long ld, int wave, int lane, unsigned (&off)[8 / NW]) {
#pragma unroll
  for (int j = 0; j < 8 / NW; ++j) {
    const int r = (wave * (8 / NW) + j) * 8 + (lane >> 3);
    off[j] = (unsigned)((r * ld + (((lane & 7) ^ swz(r)) << 3)) * 2);
  }
}
template <int NW = 4>
__device__ __forceinline__ void stage_tile_fast(const bf16_t* tile_base, const unsigned (&off)[8 / NW], char* tile, int wave) {
  const unsigned long long bv = (unsigned long long)tile_base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bv), hi = __builtin_amdgcn_readfirstlane((unsigned)(bv >> 32));
  const unsigned long long base = ((unsigned long long)hi << 32) | lo;   // pinned to SGPRs
#pragma unroll
  for (int j = 0; j < 8 / NW; ++j) {
    const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(tile + (wave * (8 / NW) + j) * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off[j]), "s"(base), "s"(la) : "memory", "m0");
  }
}
// ... one piece (j of this wave's 8 / NW) of an interior tile: issued between the MFMAs of the first product (see k_attn_bf16_q)
template <int NW = 4>
__device__ __forceinline__ void stage_piece_fast(const bf16_t* tile_base, unsigned off, char* tile, int wave, int j) {
  const unsigned long long bv = (unsigned long long)tile_base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bv), hi = __builtin_amdgcn_readfirstlane((unsigned)(bv >> 32));
  const unsigned long long base = ((unsigned long long)hi << 32) | lo;
  const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(tile + (wave * (8 / NW) + j) * 1024);
  // (no "memory" clobber: the piece fills a stage nobody reads during this iteration, so the compiler may move this iteration's LDS reads
  // across it; volatile keeps its order against the other requests, the waits and the barrier)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(la) : "m0");
}
// first-product A operand: rows rb*32 + (lane&31), 8 consecutive columns of k-step kk (16 columns per step)
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int rb, int kk, int lane) {
  const int r = rb * 32 + (lane & 31);
  return *reinterpret_cast<const bf16x8*>(tile + r * 128 + (((2 * kk + (lane >> 5)) ^ swz(r)) << 4));
}
// second-product A operand = (tile^T)[32 columns of block j][16 rows of step s in row-block rb], delivered in the
// permuted k order of an accumulator-derived B operand: element e <-> tile row rb*32 + 16s + 8(e>>2) + 4h + (e&3)
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int rb, int s, int j, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, h = g >> 1;
  const int chunk = 4 * j + 2 * (g & 1) + (p >> 1);
  const int r0 = rb * 32 + 16 * s + 4 * h + q;
  const int r1 = r0 + 8;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(tile + r0 * 128 + ((chunk ^ swz(r0)) << 4) + ((p & 1) << 3)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(tile + r1 * 128 + ((chunk ^ swz(r1)) << 4) + ((p & 1) << 3)));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}
// The same two fragments from lane offsets computed ONCE per kernel: the swizzle depends only on (row & 3, (row >> 2) & 1), which the
// row-block (32 rows), the k-step (16 rows) and the second half of a transposed read (8 rows) leave alone - so those are immediate
// offsets, and a tile costs one v_add per distinct lane offset (six) instead of one per read (the compiler does not see through the XOR).
struct FragOff {
  int row[4];   // row_frag: [kk]
  int tr[2];    // tr_frag:  [j]
};
__device__ __forceinline__ FragOff frag_offsets(int lane) {
  FragOff o;
  const int r = lane & 31;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) o.row[kk] = r * 128 + (((2 * kk + (lane >> 5)) ^ swz(r)) << 4);
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, h = g >> 1;
  const int r0 = 4 * h + q;
#pragma unroll
  for (int j = 0; j < 2; ++j) o.tr[j] = r0 * 128 + (((4 * j + 2 * (g & 1) + (p >> 1)) ^ swz(r0)) << 4) + ((p & 1) << 3);
  return o;
}
__device__ __forceinline__ bf16x8 row_frag(const char* tile, const FragOff& o, int rb, int kk) {
  return *reinterpret_cast<const bf16x8*>(tile + o.row[kk] + rb * 4096);
}
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, const FragOff& o, int rb, int s, int j) {
  const char* a = tile + o.tr[j] + rb * 4096 + s * 2048;
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
  u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 1024));
  return u.v;
}
// Row-per-lane epilogue store of a 32 x 32 accumulator block pair: lane i holds columns 8g .. 8g+3 of its row, lane i + 32 columns 8g+4 ..
// 8g+7 (8 bytes each).  One v_permlane32_swap per dword between column groups g and g+1 leaves lanes 0-31 with the 16 contiguous bytes of
// group g and lanes 32-63 with those of group g+1: ONE 16-byte store per lane and pair of groups instead of two 8-byte ones.  The tail of
// these kernels is bound by the number of store instructions (64 rows = 64 lines touched by each), not by bytes (guide T21).
// `acc` = 16 registers of the block (columns acc_row(r, h) of this lane's row), `mult` applied first; row_ptr = this lane's row, first column
// of the block; h = lane >> 5.
__device__ __forceinline__ void store_block_rows(const f32x16& acc, float mult, bf16_t* row_ptr, int h) {
#pragma unroll
  for (int g = 0; g < 4; g += 2) {
    uint32_t a[2], b[2];   // groups g, g + 1: two dwords (4 bf16) each
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      a[w] = (uint32_t)f32_to_bf16(acc[4 * g + 2 * w] * mult) | ((uint32_t)f32_to_bf16(acc[4 * g + 2 * w + 1] * mult) << 16);
      b[w] = (uint32_t)f32_to_bf16(acc[4 * g + 4 + 2 * w] * mult) | ((uint32_t)f32_to_bf16(acc[4 * g + 4 + 2 * w + 1] * mult) << 16);
      const auto r = __builtin_amdgcn_permlane32_swap(a[w], b[w], false, false);   // vdst = group g, src = group g + 1
      a[w] = r[0], b[w] = r[1];
    }
    *reinterpret_cast<uint4*>(row_ptr + 8 * g + 8 * h) = make_uint4(a[0], a[1], b[0], b[1]);
  }
}
// registers 8s..8s+7 of a 32x32 f32 accumulator -> bf16x8 B operand of k-step s
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (vfm_h)a[8 * s + e];
  return r;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }  // row of register r

#define MFMA(a, b, c) VFM_MFMA16(a, b, c)

// load the stationary operand's B fragments: 4 k-steps x 8 bf16 of row `row`, columns col0 + 16kk + 8h ..
__device__ __forceinline__ void load_stationary(const bf16_t* base, long ld, long row, int col0, int h, bf16x8 (&f)[4]) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) f[kk] = *reinterpret_cast<const bf16x8*>(base + row * ld + col0 + 16 * kk + 8 * h);
}

// ------------------------------------------------------------------------------------------- single extra row ([cls])
// With the cls-last layout a ViT sequence is nq_main = 1024 patch tokens + 1 [cls] token: as a 9th query block (one valid
// query in 128) and a 9th key block it made 576 equal-cost blocks for 512 resident slots (+30 % forward, +21 % backward).
// The block that owns only the extra row runs these VALU paths instead: dot products with v_dot2c_f32_bf16, one thread
// per streamed row for the scores, (row group, 8 columns) per thread for the weighted sums, fixed-order LDS reductions.
typedef vfm_h bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2u(unsigned a, unsigned b, float acc) {
  return VFM_DOT2(*reinterpret_cast<bf16x2v*>(&a), *reinterpret_cast<bf16x2v*>(&b), acc);
}
__device__ __forceinline__ float dot8(uint4 a, uint4 b) {
  return dot2u(a.w, b.w, dot2u(a.z, b.z, dot2u(a.y, b.y, dot2u(a.x, b.x, 0.f))));
}
__device__ __forceinline__ float sum8(float v) {  // over the 8 lanes that share a row
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}
__device__ __forceinline__ void fma8(float (&acc)[8], float w, uint4 x) {
  acc[0] = fmaf(w, h16_lo(x.x), acc[0]), acc[1] = fmaf(w, h16_hi(x.x), acc[1]);
  acc[2] = fmaf(w, h16_lo(x.y), acc[2]), acc[3] = fmaf(w, h16_hi(x.y), acc[3]);
  acc[4] = fmaf(w, h16_lo(x.z), acc[4]), acc[5] = fmaf(w, h16_hi(x.z), acc[5]);
  acc[6] = fmaf(w, h16_lo(x.w), acc[6]), acc[7] = fmaf(w, h16_hi(x.w), acc[7]);
}
// out[d] = mult * sum_i w[i] * X[row(i)][col0 + d], i < n: thread (i-group tid>>3, 8 columns tid&7), then 32-way LDS reduce
__device__ __forceinline__ void weighted_rowsum(const float* w, const bf16_t* X, long ld, int col0, int b, int n, int n_main, int B,
                                                float mult, float* red, bf16_t* out) {
  const int tid = threadIdx.x, kg = tid >> 3, d8 = tid & 7;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int i = kg; i < n; i += 32)
    fma8(acc, w[i], *reinterpret_cast<const uint4*>(X + tok_row(b, i, n_main, B) * ld + col0 + d8 * 8));
#pragma unroll
  for (int e = 0; e < 8; ++e) red[kg * 64 + d8 * 8 + e] = acc[e];
  __syncthreads();
  if (tid < 64) {
    float o = 0.f;
    for (int g = 0; g < 32; ++g) o += red[g * 64 + tid];
    out[tid] = f32_to_bf16(o * mult);
  }
  __syncthreads();
}
#define ATTN_EXTRA_MAX 2048  // the score arrays live in the kernels' 32-KiB staging area

// Scores: 8 lanes per streamed row (16 B each, one 128-B row per 8 lanes), 32 rows per step, 4 steps in flight.
static __device__ void attn_extra_fwd(const AttnP& p, int b, int hh, int qi, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = tid & 7, rg = tid >> 3;
  const int nk = p.nk_main + p.nk_extra, nq = p.nq_main + p.nq_extra;
  float* sc = reinterpret_cast<float*>(smem);
  float* red = sc + ATTN_EXTRA_MAX + 64;
  float* sh = red + 32 * 64;
  const int col0 = hh * 64;
  const long qrow = tok_row(b, qi, p.nq_main, p.B);
  const uint4 q8 = *reinterpret_cast<const uint4*>((const bf16_t*)p.q + qrow * p.ldq + col0 + sub * 8);
  const bf16_t* Kc = (const bf16_t*)p.k + col0 + sub * 8;
  float mx = -INFINITY;
  for (int k0 = 0; k0 < nk; k0 += 128) {
    uint4 kv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = min(k0 + u * 32 + rg, nk - 1);
      kv[u] = *reinterpret_cast<const uint4*>(Kc + tok_row(b, k, p.nk_main, p.B) * p.ldk);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * 32 + rg;
      const float sv = sum8(dot8(q8, kv[u])) * p.scale;
      if (k < nk) {
        if (sub == 0) sc[k] = sv;
        mx = fmaxf(mx, sv);
      }
    }
  }
  mx = wave_max(mx);
  if (lane == 0) sh[wave] = mx;
  __syncthreads();
  const float m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  float sum = 0.f;
  for (int k = tid; k < nk; k += 256) {
    const float pv = __expf(sc[k] - m);
    sc[k] = pv;
    sum += pv;
  }
  sum = wave_sum(sum);
  if (lane == 0) sh[4 + wave] = sum;
  __syncthreads();
  const float l = (sh[4] + sh[5]) + (sh[6] + sh[7]);
  weighted_rowsum(sc, (const bf16_t*)p.v, p.ldv, col0, b, nk, p.nk_main, p.B, 1.f / l, red, (bf16_t*)p.o + qrow * p.ldo + col0);
  if (tid == 0 && p.lse) p.lse[((long)b * p.H + hh) * nq + qi] = m + __logf(l);
}
// ------------------------------------------------------------------------------------------- [cls] row of the backward
// The [cls] token (sequence position n_main, stored after all patch tokens) used to get VALU blocks of its own in the dQ and
// dK/dV kernels: 64 serial blocks at the end of a 512-block grid (+30 us of a 121-us backward at bs 2).  Its gradients are
// instead gathered where the products already exist: the ragged last tile of every regular block holds, for each of its 32
// stationary positions, P and dS against the [cls] token in one accumulator register; the sum over the positions of
// (that scalar) x (the position's stationary row) is this block's contribution, reduced through a wave-private LDS image and
// added to an fp32 scratch row with one atomic instruction per wave.  k_attn_cls_finish adds the ([cls], [cls]) pair and
// writes the three bf16 rows.  (fp32 atomics: the summation order over the 32 waves of an (image, head) pair is not fixed, so
// these three rows are reproducible to fp32 rounding, not bitwise.)
//   out[col] += sum_i w[i] * F_i[col]:  w = per-lane scalar (already broadcast to both halves), f = the lane's stationary
//   fragments (columns 16 kk + 8 h + e), red = 32 x 65 floats of LDS owned by this wave
__device__ __forceinline__ void cls_partial(float w, const bf16x8 (&f)[4], float* red, int lane, float* dst) {
  const int fr = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[fr * 65 + 16 * kk + 8 * h + e] = w * (float)f[kk][e];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  float sv = 0.f;
#pragma unroll 8
  for (int i = 0; i < 32; ++i) sv += red[i * 65 + lane];
  atomicAdd(dst + lane, sv);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Rank-1 form of ONE extra streamed position (the [cls] key for the query-stationary kernels, the [cls] query for dK/dV): a tile
// of 64 for a single row would cost 8 + 8 MFMAs and 32 exponentials per wave; the same mathematics is a dot product per lane
// (its stationary row against the extra row) and one scalar-times-row update of the accumulators.
__device__ __forceinline__ float dot_frag(const bf16x8 (&a)[4], const bf16x8 (&b)[4]) {   // full 64-column dot (both lane halves)
  float s = 0.f;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf((float)a[kk][e], (float)b[kk][e], s);
  return half_sum(s);
}
// v[j][r] = X[row, col0 + 32 j + acc_row(r, h)]: a row in the accumulators' (output-column) layout
__device__ __forceinline__ void load_outcols(const bf16_t* base, long ld, long row, int col0, int h, float (&v)[2][16]) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const ushort4 u = *reinterpret_cast<const ushort4*>(base + row * ld + col0 + 32 * j + 8 * g + 4 * h);
      v[j][4 * g + 0] = bf16_to_f32(u.x), v[j][4 * g + 1] = bf16_to_f32(u.y), v[j][4 * g + 2] = bf16_to_f32(u.z), v[j][4 * g + 3] = bf16_to_f32(u.w);
    }
}

