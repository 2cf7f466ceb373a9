// bf16 MFMA flash-attention forward for head dim 64 with 64 queries per wave, one wave per SIMD (gfx950).
// EXPERIMENTAL (vfm_tune("attn_fwd64", 1); off by default): at the train-step shape (4 images x 16 heads x 1025 tokens) it runs
// 7-9 % faster than the 32-queries-per-wave kernel of attention_bf16.hip back to back (31.6 vs 34.7 us with the [cls] token, 30.4
// vs 32.1 without) and makes no measurable difference inside the train step (125.1 vs 125.2 images/s), so the proven kernel stays
// the default; see "what was learned" below.
//
// Why it was written: the 32-queries-per-wave kernel lives on two waves per SIMD that are supposed to fill each other's stalls;
// measured (profiles/r02_pmc_attention_before.txt) the matrix pipe is busy 18 % of the time, every wave reads the whole K and V
// tile from LDS for only 32 queries, and each tile's S -> softmax -> PV chain is serial inside a wave.  Here a wave owns TWO
// 32-query blocks and software-pipelines them against each other by hand:
//
//     phase   matrix pipe (8 MFMAs)        in the gaps between them
//     B(t)    O1 += V(t-1)^T P1(t-1)       softmax of S0(t), first half: row maxima, rescale decision, 8 exponentials
//     C(t)    S1(t) = K(t) Q1^T            softmax of S0(t), second half: 24 exponentials, row sums, P0(t) packed to bf16;
//                                          first V(t) fragments for D(t)
//             -- counted vmcnt, s_barrier (tile t+1 landed, tile t-1 free), LDS-DMA of tile t+4 --
//     D(t)    O0 += V(t)^T P0(t)           softmax of S1(t), first half; the eight K(t+1) fragments (held for both q-blocks)
//     A(t+1)  S0(t+1) = K(t+1) Q0^T        softmax of S1(t), second half; first V(t) fragments for B(t+1)
//
// Every softmax is spread over two matrix phases, every LDS fragment is requested several MFMAs before its use (one wave per
// SIMD: nobody else covers an LDS round trip), K fragments are read from LDS once per tile for both q-blocks, and a block of four
// waves streams each K/V tile from global memory for 256 queries instead of 128.  Tiles are 64 keys (8 KiB K + 8 KiB V), five
// stages (three tiles in flight), one barrier per tile.  The [cls] key enters in rank-1 form before the loop and the [cls] query
// runs in VALU blocks at the end of the grid, exactly as in the 32-query kernel.  Every memory and VALU instruction is placed by
// hand between two MFMAs (sched_barrier after each gap).
//
// What was learned (timing variants DBG 1 / 2 / 3 / 5 = no LDS fragment reads / no exponentials / neither / reads issued but
// feeding nothing; PMC in profiles/r02_pmc_attention_fwd64.txt; 4 x 16 x 1024 tokens, back-to-back launches incl. ~4 us of gap):
//      as built 30.4 us | no exponentials 26.0 | reads issued but not consumed 25.7 | no fragment reads 21.1 | neither 19.2
//  * a wave issues 330 VALU instructions per 32 MFMAs and tile; with one wave per SIMD a gap costs 8 cycles for the MFMA plus the
//    issue cycles of everything placed in it (exp 8, the rest 4): ~1700 cycles per tile, 14 us for the loop - the kernel is bound
//    by VALU issue, so the exponentials have to be spread evenly over ALL gaps (first version: all in the S phases, 20 % slower);
//  * on top of that, ~5 us are MFMAs waiting for LDS fragments although every fragment is requested 3-8 MFMAs (100-500 cycles)
//    ahead, and ~4 us are the LDS read instructions themselves; halving the LDS bank conflicts (new tile swizzle) and going from
//    3 to 5 LDS-DMA stages changed nothing;
//  * what DID help both forwards and the backward: sending all blocks of an (image, head) pair to one XCD (xcd_map) - the K/V
//    stream then comes from that XCD's L2 instead of the fabric (152 MB fetched per launch before);
//  * pitfalls met on the way, all visible only in the ISA: a basic-block boundary inside the pipelined region (the `if (t + 2 <
//    nt)` around the DMA issue, the rescale branch) lets the optimiser SINK the exponentials and the row-sum adds out of their
//    gaps down to their first use (sched_barrier does not stop IR-level sinking; fixed by removing the branch and by an empty
//    asm volatile that pins the row sums); fmaxf() on accumulator values emits a canonicalising v_max(x, x) per operand (inline
//    v_max3_f32 instead); __shfl_xor(v, 32) is an LDS round trip (v_permlane32_swap instead); 32-bit DMA offsets hoisted out of
//    the loop become 64-bit per-lane addresses and lose the SGPR-base form (an empty asm keeps the zero-extension in the loop);
//    SLP packs the row-sum adds into v_pk_add_f32 (-fno-slp-vectorize for this file); __builtin_amdgcn_readfirstlane returns
//    int - OR-ing a sign-extended low word into a 64-bit address faults.
#include "attn_bf16_dev.h"

template <int V>
struct ICn {
  static constexpr int value = V;
};
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (N > 0) {
    sfor<N - 1>(f);
    f(ICn<N - 1>{});
  }
}
#define INL __attribute__((always_inline))

#define STAGE_BYTES (2 * TILE_BYTES)

// max(a, b, c) without the canonicalising v_max(x, x) that fmaxf() puts on accumulator values in IEEE mode
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float d;
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
#else
  d = fmaxf(fmaxf(a, b), c);
#endif
  return d;
}

// NST stages of (K tile, V tile): NST - 2 tiles are in flight while one is being multiplied.  (Three stages = one tile in flight
// measured 10 k of a wave's 65 k cycles waiting for the LDS-DMA: an L2 -> LDS round trip under load is longer than one tile.)
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N == 0, "add the literal");
}

template <int DBG, int NST>
__global__ void __launch_bounds__(256, 2) k_attn_fwd64(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NST stages x (K tile, V tile)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = p.nq_main + p.nq_extra;
  int bx, bh;
  {  // the [cls] query's VALU blocks take the last linear ids (see k_attn_bf16_q); the others go to the XCD of their pair
    const int gx = gridDim.x, nbx = gx - (p.nq_extra == 1 ? 1 : 0), nfull = nbx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
    if (lin >= nfull) {
      const int e = lin - nfull;
      attn_extra_fwd(p, e / p.H, e % p.H, p.nq_main, smem);
      return;
    }
    xcd_map(lin, nbx, gridDim.y, p.xcd, bx, bh);
  }
  const int b = bh / p.H, hh = bh % p.H;
  const int col0 = hh * 64;
  const int h = lane >> 5, fr = lane & 31;
  const int q0 = bx * 256 + wave * 64;
  const bf16_t* Kb = (const bf16_t*)p.k;
  const bf16_t* Vb = (const bf16_t*)p.v;
  const float c = p.scale * LOG2E;

  bf16x8 qf[2][4];
#pragma unroll
  for (int Q = 0; Q < 2; ++Q) load_stationary((const bf16_t*)p.q, p.ldq, (long)b * p.nq_main + q0 + 32 * Q + fr, col0, h, qf[Q]);

  // ---- LDS-DMA sources: wave w moves rows 16w .. 16w+15 of a tile (two 1-KiB pieces per operand); 32-bit byte offsets from a
  // scalar base that advances by 64 rows per tile
  unsigned ksrc[2], vsrc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 8 + (lane >> 3);
    const int cc = (lane & 7) ^ swz(r);
    ksrc[j] = (unsigned)((r * p.ldk + cc * 8) * 2);
    vsrc[j] = (unsigned)((r * p.ldv + cc * 8) * 2);
  }
  auto uniform = [](const void* ptr) INL {  // pin a wave-uniform address to SGPRs
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);  // (unsigned: the builtin returns int, and a sign-extended low word would smear the high one)
  };
  const char* kg = uniform(Kb + (long)b * p.nk_main * p.ldk + col0);
  const char* vg = uniform(Vb + (long)b * p.nk_main * p.ldv + col0);
  const long kstep = 128 * p.ldk, vstep = 128 * p.ldv;  // bytes per 64-row tile
  auto stage = [&](int t) INL {                         // tile t -> stage t % NST
    char* kt = smem + (t % NST) * STAGE_BYTES + wave * 2048;
    const char* kbse = uniform(kg + t * kstep);
    const char* vbse = uniform(vg + t * vstep);
    unsigned o[4] = {ksrc[0], ksrc[1], vsrc[0], vsrc[1]};
#if defined(__HIP_DEVICE_COMPILE__)
    // keep the zero-extension of the 32-bit offsets next to the loads (instruction selection works per basic block: hoisted out
    // of the loop they become four 64-bit per-lane addresses and the loads lose their SGPR-base form)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(o[j]));
#endif
    glds16(kbse + o[0], kt), glds16(kbse + o[1], kt + 1024);
    glds16(vbse + o[2], kt + TILE_BYTES), glds16(vbse + o[3], kt + TILE_BYTES + 1024);
  };

  // ---- per-lane fragment addresses inside stage 0 (see row_frag / tr_frag); key blocks and k-steps are immediates, the stage
  // offset is added once per tile (kc / vl / vh below)
  const char* kof[4];
  const char *vlo[2], *vhi[2];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) kof[kk] = smem + fr * 128 + (((2 * kk + h) ^ swz(fr)) << 4);
  {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h2 = g >> 1;
    const int sw0 = swz(4 * h2 + q);  // = swz(row) of every row this lane reads: rows are kb * 32 + 16 s + 4 h2 + q (+ 8), bits 0..2 fixed
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int chunk = 4 * j + 2 * (g & 1) + (pp >> 1);
      vlo[j] = smem + TILE_BYTES + (4 * h2 + q) * 128 + ((chunk ^ sw0) << 4) + ((pp & 1) << 3);
      vhi[j] = vlo[j] + 8 * 128;
    }
  }

  f32x16 oacc[2][2], sacc[2][2];
  float m[2], l[2];
  stage(0);
  if (p.nk_extra == 1) {  // the online softmax starts from the [cls] key: m = its score, p = 1, O = v_cls
    const long crow = (long)p.B * p.nk_main + b;
    bf16x8 kcf[4];
    load_stationary(Kb, p.ldk, crow, col0, h, kcf);
    float oc[2][16];
    load_outcols(Vb, p.ldv, crow, col0, h, oc);
#pragma unroll
    for (int Q = 0; Q < 2; ++Q) {
      m[Q] = dot_frag(qf[Q], kcf);
      l[Q] = h == 0 ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[Q][j][r] = oc[j][r];
    }
  } else {
#pragma unroll
    for (int Q = 0; Q < 2; ++Q) {
      m[Q] = -INFINITY, l[Q] = 0.f;
      oacc[Q][0] = zero16(), oacc[Q][1] = zero16();
    }
  }
#pragma unroll
  for (int t0 = 1; t0 < NST - 1; ++t0) stage(t0);

  // fragment addresses in the stage of the K tile / V tile being read: they start at stage 0 (tile 0) and move by one stage per
  // tile (a wave-uniform delta: + one stage, or back to stage 0)
  const char* kc[4] = {kof[0], kof[1], kof[2], kof[3]};
  const char *vl[2] = {vlo[0], vlo[1]}, *vh[2] = {vhi[0], vhi[1]};
  auto stage_delta = [](int t_new) INL { return (t_new % NST) == 0 ? -(NST - 1) * STAGE_BYTES : STAGE_BYTES; };  // from tile t_new - 1
  auto next_k = [&](int t_new) INL {
    const int dlt = stage_delta(t_new);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kc[kk] += dlt;
  };
  auto next_v = [&](int t_new) INL {
    const int dlt = stage_delta(t_new);
#pragma unroll
    for (int j = 0; j < 2; ++j) vl[j] += dlt, vh[j] += dlt;
  };
  bf16x8 kf[8];     // K fragments of the tile in flight: [kb * 4 + kk], read once per tile for both q-blocks (during D)
  bf16x8 vf[4][2];  // V^T fragments of the PV phase that follows: [(key block, k-step) step][column block j]
  bf16x8 pk[4];     // P^T fragments (B operands) of that phase: [step]
  float m4[4], rs4[4], pe[8], mnc = 0.f, alpha = 1.f;
  if constexpr (DBG & 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) kf[k] = qf[0][k & 3];
#pragma unroll
    for (int k = 0; k < 4; ++k) vf[k][0] = vf[k][1] = qf[0][1];
  }

  bf16x8 kd[8], vd[4][2];  // DBG & 4: the fragment reads are issued but land here and feed nothing (sunk at the end of the phase)
  auto sink = [&](bf16x8& v) INL {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(v));
#endif
  };
  auto ldk = [&](auto Ic) INL {
    constexpr int i = decltype(Ic)::value;
    if constexpr (DBG & 4) {
      kd[i] = *reinterpret_cast<const bf16x8*>(kc[i & 3] + (i >> 2) * 4096);
      return;
    }
    if constexpr (DBG & 1) return;  // timing experiment: no LDS fragment reads (results are garbage)
    kf[i] = *reinterpret_cast<const bf16x8*>(kc[i & 3] + (i >> 2) * 4096);
  };
  auto ldv = [&](auto Kc) INL {  // step k = 2 * kb + s: rows kb * 32 + 16 s + ...; both column blocks
    constexpr int k = decltype(Kc)::value;
    if constexpr ((DBG & 1) && !(DBG & 4)) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vl[j] + k * 2048));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vh[j] + k * 2048));
      union {
        struct {
          s16x4 a, b;
        } s;
        bf16x8 v;
      } u;
      u.s.a = lo, u.s.b = hi;
      if constexpr (DBG & 4) vd[k][j] = u.v;
      else vf[k][j] = u.v;
    }
  };
  auto estep = [&](auto Qc, auto Ec) INL {  // element e of 32: p = exp2(s * c - m * c); four row-sum chains; packed per eight
    constexpr int Q = decltype(Qc)::value, e = decltype(Ec)::value;
    const float pv = (DBG & 2) ? fmaf(sacc[Q][e >> 4][e & 15], c, -mnc) : __builtin_amdgcn_exp2f(fmaf(sacc[Q][e >> 4][e & 15], c, -mnc));
    pe[e & 7] = pv;
    if constexpr (e < 4) rs4[e] = pv;
    else rs4[e & 3] += pv;
    if constexpr ((e & 7) == 7) {  // elements 8k .. 8k+7 are step k's B operand (registers 8s .. 8s+7 of key block kb, k = 2 kb + s)
#pragma unroll
      for (int x = 0; x < 8; ++x) pk[e >> 3][x] = (__bf16)pe[x];
    }
  };
  // The softmax of q-block Q, cut into the gaps of the two matrix phases that follow its S phase:
  //   sm1 (gaps of the other q-block's PV phase): 0-2 row maximum (16 v_max3 in four chains, then across the two lane halves),
  //       3 the lazy-rescale decision (and, rarely, the rescale), 4-7 elements 0..7
  //   sm2 (gaps of the other q-block's S phase): three elements per gap (8..31); row sums closed in gap 7
  auto sv = [&](auto Qc, auto Ec) INL -> float { return sacc[decltype(Qc)::value][decltype(Ec)::value >> 4][decltype(Ec)::value & 15]; };
  auto sm1 = [&](auto Qc, auto Ic) INL {
    constexpr int Q = decltype(Qc)::value, i = decltype(Ic)::value;
    if constexpr (i == 0) {
      m4[0] = max3f(sv(Qc, ICn<0>{}), sv(Qc, ICn<1>{}), sv(Qc, ICn<2>{}));
      m4[1] = max3f(sv(Qc, ICn<3>{}), sv(Qc, ICn<4>{}), sv(Qc, ICn<5>{}));
      m4[2] = max3f(sv(Qc, ICn<6>{}), sv(Qc, ICn<7>{}), sv(Qc, ICn<8>{}));
      m4[3] = max3f(sv(Qc, ICn<9>{}), sv(Qc, ICn<10>{}), sv(Qc, ICn<11>{}));
      m4[0] = max3f(m4[0], sv(Qc, ICn<12>{}), sv(Qc, ICn<13>{}));
      m4[1] = max3f(m4[1], sv(Qc, ICn<14>{}), sv(Qc, ICn<15>{}));
    } else if constexpr (i == 1) {
      m4[2] = max3f(m4[2], sv(Qc, ICn<16>{}), sv(Qc, ICn<17>{}));
      m4[3] = max3f(m4[3], sv(Qc, ICn<18>{}), sv(Qc, ICn<19>{}));
      m4[0] = max3f(m4[0], sv(Qc, ICn<20>{}), sv(Qc, ICn<21>{}));
      m4[1] = max3f(m4[1], sv(Qc, ICn<22>{}), sv(Qc, ICn<23>{}));
      m4[2] = max3f(m4[2], sv(Qc, ICn<24>{}), sv(Qc, ICn<25>{}));
      m4[3] = max3f(m4[3], sv(Qc, ICn<26>{}), sv(Qc, ICn<27>{}));
    } else if constexpr (i == 2) {
      m4[0] = max3f(m4[0], sv(Qc, ICn<28>{}), sv(Qc, ICn<29>{}));
      m4[1] = max3f(m4[1], sv(Qc, ICn<30>{}), sv(Qc, ICn<31>{}));
      m4[0] = half_max(max3f(max3f(m4[0], m4[1], m4[2]), m4[3], m4[3]));
    } else if constexpr (i == 3) {
      // lazy rescale: the reference maximum only moves when the tile maximum exceeds it by more than 2^8 in the exp2 domain
      const float mx = m4[0];
      const bool need = (mx - m[Q]) * c > 8.0f;
      const float mn = need ? mx : m[Q];
      alpha = need ? __builtin_amdgcn_exp2f((m[Q] - mn) * c) : 1.0f;
      mnc = mn * c;
      m[Q] = mn;
      l[Q] *= alpha;
      if (__ballot(alpha != 1.0f) != 0ull) {  // wave-uniform, rare after the first tiles
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[Q][j][r] *= alpha;
      }
    } else {
      estep(Qc, ICn<2 * (i - 4)>{}), estep(Qc, ICn<2 * (i - 4) + 1>{});
    }
  };
  auto sm2 = [&](auto Qc, auto Ic) INL {
    constexpr int Q = decltype(Qc)::value, i = decltype(Ic)::value;
    sfor<3>([&](auto Jc) INL { estep(Qc, ICn<8 + 3 * i + decltype(Jc)::value>{}); });
    if constexpr (i == 7) {
      l[Q] += (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
#if defined(__HIP_DEVICE_COMPILE__)
      // pin the row sums here: l is next read behind the other q-block's rescale branch, and the optimiser sinks the whole add
      // chain (with the probabilities it keeps alive) down to that block
      asm volatile("" : "+v"(l[Q]));
#endif
    }
  };

  // PV phase: O[QP] += V^T P[QP] over the four (key block, k-step) steps (HASPV); P and the V fragments of steps 0, 1 are in
  // registers.  In the gaps: the V fragments of steps 2, 3, the first half of q-block QM's softmax (HASM), the eight fragments of
  // the K tile at kc (RK)
  auto pv_phase = [&](auto QPc, auto QMc, auto HASPVc, auto HASMc, auto RKc) INL {
    constexpr int QP = decltype(QPc)::value;
    constexpr bool HASPV = decltype(HASPVc)::value, HASM = decltype(HASMc)::value, RK = decltype(RKc)::value;
    sfor<8>([&](auto Ic) INL {
      constexpr int i = decltype(Ic)::value, k = i >> 1, j = i & 1;
      if constexpr (HASPV) {
        oacc[QP][j] = MFMA(vf[k][j], pk[k], oacc[QP][j]);
        if constexpr (i < 2) ldv(ICn<i + 2>{});
      }
      if constexpr (RK) ldk(Ic);
      if constexpr (HASM) sm1(QMc, Ic);
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (DBG & 4) {
      if constexpr (RK) sfor<8>([&](auto Ic) INL { sink(kd[decltype(Ic)::value]); });
      if constexpr (HASPV) sink(vd[2][0]), sink(vd[2][1]), sink(vd[3][0]), sink(vd[3][1]);
    }
  };
  // S / exp phase: S[QS] = K Q[QS]^T from the held K fragments (HASS); in the gaps: the second half of q-block QE's softmax and
  // the V fragments (steps 0, 1) of the PV phase that follows
  auto e_phase = [&](auto QSc, auto QEc, auto HASSc) INL {
    constexpr int QS = decltype(QSc)::value;
    constexpr bool HASS = decltype(HASSc)::value;
    sfor<8>([&](auto Ic) INL {
      constexpr int i = decltype(Ic)::value, kb = i >> 2, kk = i & 3;
      if constexpr (HASS) {
        if constexpr (kk == 0) sacc[QS][kb] = MFMA(kf[i], qf[QS][kk], zero16());
        else sacc[QS][kb] = MFMA(kf[i], qf[QS][kk], sacc[QS][kb]);
      }
      if constexpr (i == 3 || i == 4) ldv(ICn<i - 3>{});
      sm2(QEc, Ic);
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (DBG & 4) sink(vd[0][0]), sink(vd[0][1]), sink(vd[1][0]), sink(vd[1][1]);
  };

  const int nt = p.nk_main / TROWS;  // >= NST + 1 (checked by the dispatcher); tile t lives in stage t % NST
  // one tile.  ISSUE: the LDS-DMA of tile t + NST - 1 goes out behind the barrier (into the stage of tile t - 1); otherwise
  // (the last NST - 1 tiles) nothing is issued and the wait before the barrier is for everything
  auto iter = [&](auto FIRSTc, auto ISSUEc, auto LASTc, int t) INL {
    constexpr bool FIRST = decltype(FIRSTc)::value, ISSUE = decltype(ISSUEc)::value, LAST = decltype(LASTc)::value;
    // B(t): P1(t-1) against V(t-1) (vl / vh still point at tile t-1); first half of softmax0(t)
    pv_phase(ICn<1>{}, ICn<0>{}, ICn<!FIRST>{}, ICn<true>{}, ICn<false>{});
    if constexpr (!FIRST) next_v(t);
    e_phase(ICn<1>{}, ICn<0>{}, ICn<true>{});  // C(t)
    if constexpr (ISSUE) wait_vm<4 * (NST - 3)>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if constexpr (ISSUE) stage(t + NST - 1);
    if constexpr (!LAST) next_k(t + 1);
    __builtin_amdgcn_sched_barrier(0);
    pv_phase(ICn<0>{}, ICn<1>{}, ICn<true>{}, ICn<true>{}, ICn<!LAST>{});  // D(t); K(t+1) fragments
    e_phase(ICn<0>{}, ICn<1>{}, ICn<!LAST>{});                             // A(t+1)
  };

  // ---- prologue: tile 0 landed; S0(0)
  wait_vm<4 * (NST - 2)>();
  __builtin_amdgcn_s_barrier();
  sfor<8>([&](auto Ic) INL { ldk(Ic); });
  sfor<8>([&](auto Ic) INL {
    constexpr int i = decltype(Ic)::value, kb = i >> 2, kk = i & 3;
    if constexpr (kk == 0) sacc[0][kb] = MFMA(kf[i], qf[0][kk], zero16());
    else sacc[0][kb] = MFMA(kf[i], qf[0][kk], sacc[0][kb]);
  });
  __builtin_amdgcn_sched_barrier(0);
  iter(ICn<true>{}, ICn<true>{}, ICn<false>{}, 0);
  int t = 1;
  for (; t + NST - 1 < nt; ++t) iter(ICn<false>{}, ICn<true>{}, ICn<false>{}, t);
  for (; t < nt - 1; ++t) iter(ICn<false>{}, ICn<false>{}, ICn<false>{}, t);
  iter(ICn<false>{}, ICn<false>{}, ICn<true>{}, nt - 1);
  pv_phase(ICn<1>{}, ICn<0>{}, ICn<true>{}, ICn<false>{}, ICn<false>{});

  // ---- epilogue: lane = query, registers = output columns acc_row(r, h) + 32 j
#pragma unroll
  for (int Q = 0; Q < 2; ++Q) {
    const float lt = half_sum(l[Q]);
    const float mult = 1.f / lt;
    const int qi = q0 + 32 * Q + fr;
    const long qrow = (long)b * p.nq_main + qi;
    if (h == 0 && p.lse) p.lse[((long)b * p.H + hh) * nq + qi] = m[Q] * p.scale + __logf(lt);
    bf16_t* out = (bf16_t*)p.o;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        ushort4 v;
        v.x = f32_to_bf16(oacc[Q][j][4 * g + 0] * mult);
        v.y = f32_to_bf16(oacc[Q][j][4 * g + 1] * mult);
        v.z = f32_to_bf16(oacc[Q][j][4 * g + 2] * mult);
        v.w = f32_to_bf16(oacc[Q][j][4 * g + 3] * mult);
        *reinterpret_cast<ushort4*>(out + qrow * p.ldo + col0 + 32 * j + 8 * g + 4 * h) = v;
      }
  }
}

// Launches the 64-queries-per-wave forward when the shape fits it (main tokens a multiple of 256 on both sides, at most the [cls]
// token extra on both, enough (image, head) pairs to give most CUs a block); returns false otherwise.
bool vfm_attn_fwd64_launch(const vfm_attn_desc* d, const AttnP& p, hipStream_t s) {
  constexpr int NST = 5;
  if (d->nq_main != d->nk_main || d->nq_main % 256 != 0 || d->nq_extra != d->nk_extra) return false;
  if (d->nk_main / TROWS < NST + 1) return false;
  if (d->nk_main + d->nk_extra > ATTN_EXTRA_MAX) return false;
  const long blocks = (long)(d->nq_main / 256) * d->B * d->H;
  if (blocks < 192) return false;
  extern int g_attn_fwd64;
  const dim3 grid(d->nq_main / 256 + (d->nq_extra ? 1 : 0), d->B * d->H);
  constexpr int SMEM = NST * STAGE_BYTES;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_attn_fwd64<0, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    (void)hipFuncSetAttribute((const void*)k_attn_fwd64<1, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    (void)hipFuncSetAttribute((const void*)k_attn_fwd64<2, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    (void)hipFuncSetAttribute((const void*)k_attn_fwd64<3, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    (void)hipFuncSetAttribute((const void*)k_attn_fwd64<5, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr = true;
  }
  if (g_attn_fwd64 == 3) hipLaunchKernelGGL((k_attn_fwd64<1, NST>), grid, dim3(256), SMEM, s, p);  // timing experiments
  else if (g_attn_fwd64 == 5) hipLaunchKernelGGL((k_attn_fwd64<2, NST>), grid, dim3(256), SMEM, s, p);
  else if (g_attn_fwd64 == 7) hipLaunchKernelGGL((k_attn_fwd64<3, NST>), grid, dim3(256), SMEM, s, p);
  else if (g_attn_fwd64 == 9) hipLaunchKernelGGL((k_attn_fwd64<5, NST>), grid, dim3(256), SMEM, s, p);  // reads issued, feed nothing
  else hipLaunchKernelGGL((k_attn_fwd64<0, NST>), grid, dim3(256), SMEM, s, p);
  return true;
}
