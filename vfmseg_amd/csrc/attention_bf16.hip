// bf16 MFMA flash attention for head dim 64 (forward, dQ, dK/dV), gfx950.
//
// One wave owns 32 "stationary" sequence positions (queries for forward/dQ, keys for dK/dV): those sit on the MFMA
// *lane* (column) index, so softmax statistics, LSE and delta are per-lane scalars and the first product's f32
// accumulator is, after a bf16 convert, directly the B operand of the second product ("accumulator tile as the next
// MFMA's operand", cdna_hip_programming.md §3) - no LDS round trip for P / dS.
//   forward : S^T = K Q^T            ;  O^T  += V^T P^T
//   dQ      : S^T, dP^T = V dO^T     ;  dQ^T += K^T dS^T
//   dK/dV   : S = Q K^T, dP = dO V^T ;  dV^T += dO^T P ,  dK^T += Q^T dS
// The streamed operand (K,V / Q,dO) is staged in 64-row LDS tiles by LDS-DMA (global_load_lds_dwordx4), double
// buffered, 16-byte chunks XOR-swizzled by row so that both the row reads (ds_read_b128, first product) and the
// transposed reads (ds_read_b64_tr_b16, second product) spread over the banks.
#include "attn_bf16_dev.h"

// ------------------------------------------------------------------------------------------------------ forward
// NW = waves per block (32 stationary positions each): 4, or 2 for short problems whose 4-wave grid would leave CUs idle
// (the decoder's 16 (image, head) pairs x 1024 queries are only 128 blocks of 128 queries).
template <bool DQ, int NW = 4>
__global__ void __launch_bounds__(NW * 64, 2) k_attn_bf16_q(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 3 stages x (K tile, V tile)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = p.nq_main + p.nq_extra, nk = p.nk_main + p.nk_extra;
  int bx, bh;
  {
    // The [cls] query of the forward gets a block of its own that runs the VALU path.  Those blocks take the LAST linear ids:
    // workgroups are placed on the CUs in id order at launch, and a light block in the middle of the order pushes a third full
    // block onto some CUs (measured: +8 us on a 39-us kernel even when the light block returns at once).
    const bool light = !DQ && NW == 4 && p.nq_extra == 1 && (p.nq_main & 127) == 0 && nk <= ATTN_EXTRA_MAX;
    const int gx = gridDim.x, nbx = gx - (light ? 1 : 0), nfull = nbx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
    if (lin >= nfull) {
      const int e = lin - nfull;
      if constexpr (!DQ) attn_extra_fwd(p, e / p.H, e % p.H, p.nq_main, smem);
      return;
    }
    xcd_map(lin, nbx, gridDim.y, p.xcd, bx, bh);
  }
  // backward: the [cls] query has no block (its dQ comes from the dK/dV kernel's partials, see cls_partial); this kernel gathers
  // the [cls] KEY's dK / dV from its regular blocks
  const bool cls_key = DQ && NW == 4 && p.nk_extra == 1 && p.cls_scratch != nullptr;
  const int b = bh / p.H, hh = bh % p.H;
  const int col0 = hh * 64;
  const int h = lane >> 5;
  const FragOff fo = frag_offsets(lane);
  const int q0 = bx * (NW * 32) + wave * 32;
  const int qi = q0 + (lane & 31);
  const bool qvalid = qi < nq;
  const long qrow = tok_row(b, qvalid ? qi : nq - 1, p.nq_main, p.B);
  const bf16_t* Kb = (const bf16_t*)p.k;
  const bf16_t* Vb = (const bf16_t*)p.v;
  bf16x8 qf[4], dof[4];
  load_stationary((const bf16_t*)p.q, p.ldq, qrow, col0, h, qf);
  float lse_l = 0.f, delta_l = 0.f;
  if (DQ) {
    load_stationary((const bf16_t*)p.dout, p.ld_do, qrow, col0, h, dof);
    lse_l = p.lse[((long)b * p.H + hh) * nq + (qvalid ? qi : nq - 1)] * LOG2E;
    // delta = rowsum(dO o O), computed here (each lane holds half of the row's 64 columns) and published - NEGATED - for the dK/dV
    // kernel that runs next (no separate delta launch): there -delta is the initial value of the dP accumulators, so that
    // dS = P (dP - delta) needs no subtraction (32 vector instructions per wave and tile that the compiler does not pair)
    bf16x8 of[4];
    load_stationary((const bf16_t*)p.o, p.ldo, qrow, col0, h, of);
    float dsum = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int e2 = 0; e2 < 8; ++e2) dsum = fmaf((float)dof[kk][e2], (float)of[kk][e2], dsum);
    delta_l = half_sum(dsum);
    if (qvalid && h == 0) p.delta[((long)b * p.H + hh) * nq + qi] = -delta_l;
    if (cls_key && p.nq_extra == 1 && bx == 0 && wave == 0) {   // delta of the [cls] query, for the dK/dV kernel's last query tile
      const long crow = tok_row(b, p.nq_main, p.nq_main, p.B);
      bf16x8 cg[4], co[4];
      load_stationary((const bf16_t*)p.dout, p.ld_do, crow, col0, h, cg);
      load_stationary((const bf16_t*)p.o, p.ldo, crow, col0, h, co);
      float cs = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) cs = fmaf((float)cg[kk][e2], (float)co[kk][e2], cs);
      cs = half_sum(cs);
      if (lane == 0) p.delta[((long)b * p.H + hh) * nq + p.nq_main] = -cs;
    }
  }
  float p_cls = 0.f, ds_cls = 0.f;   // P^T / dS^T of this lane's query against the [cls] key (row 0 of the ragged last tile)
  const float c = p.scale * LOG2E;
  f32x16 oacc[2] = {zero16(), zero16()};
  float m = -INFINITY, l = 0.f;
  // the [cls] key in rank-1 form (4-wave grids): forward always, dQ when the [cls] partial sums are gathered here
  const bool r1 = NW == 4 && p.nk_extra == 1 && (!DQ || cls_key);
  const int nk_loop = r1 ? p.nk_main : nk;
  const int nt = (nk_loop + TROWS - 1) / TROWS;
  unsigned koff[8 / NW], voff[8 / NW];   // this lane's byte offsets inside an interior tile
  tile_offsets<NW>(p.ldk, wave, lane, koff);
  tile_offsets<NW>(p.ldv, wave, lane, voff);
  const int n_int = min(nk_loop, p.nk_main) / TROWS;   // tiles 0 .. n_int-1 hold 64 consecutive patch-token rows of image b
  const bool span_ok = 64l * p.ldk * 2 < (1l << 31) && 64l * p.ldv * 2 < (1l << 31);
  auto stage = [&](int buf, int t) {
    char* kt = smem + buf * 2 * TILE_BYTES;
    if (t < n_int && span_ok) {
      const long row0 = (long)b * p.nk_main + (long)t * TROWS;
      stage_tile_fast<NW>(Kb + row0 * p.ldk + col0, koff, kt, wave);
      stage_tile_fast<NW>(Vb + row0 * p.ldv + col0, voff, kt + TILE_BYTES, wave);
    } else {
      stage_tile<NW, true>(Kb, p.ldk, col0, b, t * TROWS, nk_loop, p.nk_main, p.B, kt, wave, lane);
      stage_tile<NW, true>(Vb, p.ldv, col0, b, t * TROWS, nk_loop, p.nk_main, p.B, kt + TILE_BYTES, wave, lane);
    }
  };
  // three K/V stages, ONE barrier per tile: the barrier that publishes tile t also certifies that every wave is done with tile
  // t-1, whose stage is the one tile t+2 will be written to (by the stage() call of the NEXT iteration)
  stage(0, 0);
  // (the extra row's loads are issued behind the first tile's DMA: their latency is the DMA's)
  if (r1) {
    const long crow = tok_row(b, p.nk_main, p.nk_main, p.B);
    bf16x8 kcf[4];
    load_stationary(Kb, p.ldk, crow, col0, h, kcf);
    const float s_e = dot_frag(qf, kcf);
    float oc[2][16];
    if (!DQ) {   // the online softmax starts from the [cls] key: m = its score, p = 1, O = v_cls
      load_outcols(Vb, p.ldv, crow, col0, h, oc);
      m = s_e;
      l = h == 0 ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = oc[j][r];
    } else {     // dS against the [cls] key: dQ^T += dS k_cls; P / dS kept for the [cls] key's own gradients
      bf16x8 vcf[4];
      load_stationary(Vb, p.ldv, crow, col0, h, vcf);
      const float dp_e = dot_frag(dof, vcf);
      p_cls = __builtin_amdgcn_exp2f(fmaf(s_e, c, -lse_l));
      ds_cls = p_cls * (dp_e - delta_l);
      load_outcols(Kb, p.ldk, crow, col0, h, oc);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = ds_cls * oc[j][r];
    }
  }
  // The stationary operands have landed before the loop starts, and the compiler knows it: with loads still pending on its scoreboard
  // at the loop head it would wait for them at their first use INSIDE the loop - every iteration, as vmcnt(0), i.e. for the tile that
  // was just requested (the LDS-DMA loads are invisible to it, its own waits are not).
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    asm volatile("" ::"v"(qf[kk]));
    if constexpr (DQ) asm volatile("" ::"v"(dof[kk]));
  }
  asm volatile("" ::"v"(lse_l), "v"(delta_l), "v"(m), "v"(l), "v"(p_cls), "v"(ds_cls));
#pragma unroll
  for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(oacc[j]));
  // Where the next tile is requested.  At the top of the iteration the four waves of a block - and those of the block that shares the CU -
  // issue their LDS-DMA pieces at the same moment: 16-32 wave-instructions queue in front of the CU's address unit and every wave stands
  // ~370 clocks in front of its barrier (s_memtime stamps: 15 % of a 2400-clock tile).  An interior next tile is therefore requested
  // piece by piece BETWEEN the MFMAs of Q K^T (its stage was released by the barrier this iteration has just passed; it has a whole
  // iteration to land and is waited for, as vmcnt(0), at the top of the next one).  Ragged / [cls]-carrying tiles keep the request at the top.
  constexpr int PT = 8 / NW;   // pieces per wave and tile
  int buf = 0;
  for (int t = 0; t < nt; ++t) {
    const int nbuf = buf == 2 ? 0 : buf + 1;
    const bool il = (p.il & (DQ ? 2 : 1)) && (t + 1 < nt) && (t + 1 < n_int) && span_ok;   // next tile: requested between the MFMAs below
    if (t + 1 < nt && !il) {
      stage(nbuf, t + 1);
      if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const char* kt = smem + buf * 2 * TILE_BYTES;
    const char* vt = kt + TILE_BYTES;
    f32x16 sacc[2] = {zero16(), zero16()};
    {
      bf16x8 kfr[8];   // all row fragments of the K tile first: no LDS read has to cross a DMA request (the asm is a memory barrier to the compiler)
#pragma unroll
      for (int i = 0; i < 8; ++i) kfr[i] = row_frag(kt, fo, i >> 2, i & 3);
      char* nkt = smem + nbuf * 2 * TILE_BYTES;
      const long row1 = (long)b * p.nk_main + (long)(t + 1) * TROWS;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sacc[i >> 2] = MFMA(kfr[i], qf[i & 3], sacc[i >> 2]);
        if (il) {
          if constexpr (NW == 4) {
            if (i & 1) {
              const int j = i >> 1;   // 0, 1: K pieces; 2, 3: V pieces
              if (j < PT) stage_piece_fast<NW>(Kb + row1 * p.ldk + col0, koff[j], nkt, wave, j);
              else stage_piece_fast<NW>(Vb + row1 * p.ldv + col0, voff[j - PT], nkt + TILE_BYTES, wave, j - PT);
            }
          } else {
            if (i < PT) stage_piece_fast<NW>(Kb + row1 * p.ldk + col0, koff[i], nkt, wave, i);
            else stage_piece_fast<NW>(Vb + row1 * p.ldv + col0, voff[i - PT], nkt + TILE_BYTES, wave, i - PT);
          }
        }
      }
    }
    const bool tail = (t == nt - 1) && (nk_loop % TROWS != 0);
    if (!DQ) {
      // ---- online softmax over this lane's 32 keys of the tile (the other 32 live in lane ^ 32)
      float mx = -INFINITY;
      if (tail) {  asm volatile("" ::: "memory");  // keeps this a real branch; wave-uniform: only the ragged last tile pays for the key mask
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (t * TROWS + kb * 32 + acc_row(r, h) >= nk_loop) sacc[kb][r] = -INFINITY;
      }
      {  // four independent max chains (a single 32-deep dependent chain leaves the VALU idle at two waves per SIMD)
        float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) m4[r & 3] = fmaxf(m4[r & 3], sacc[kb][r]);
        mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      }
      mx = half_max(mx);
      // lazy rescale: the reference maximum m only moves when the tile maximum exceeds it by more than 2^8 in the exp2
      // domain (p <= 256 stays exact enough in bf16 / fp32 sums); after the first tiles the 32 accumulator rescales and
      // the alpha exp are skipped for the whole wave.  m = -inf on the first tile, so it always takes the update there.
      const bool need = (mx - m) * c > 8.0f;
      const float mn = need ? mx : m;
      const float alpha = need ? __builtin_amdgcn_exp2f((m - mn) * c) : 1.0f;
      const float mnc = mn * c;
      // exponent arguments and row sums on register PAIRS (v_pk_fma_f32 / v_pk_add_f32: half the issue slots of the scalar forms; the
      // 32 v_exp_f32 are quarter rate and cannot be paired)
      f32x2 rs2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};  // two independent row-sum chains of pairs
      const f32x2 c2 = f32x2{c, c}, mnc2 = f32x2{-mnc, -mnc};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 x = __builtin_elementwise_fma(f32x2{sacc[kb][r], sacc[kb][r + 1]}, c2, mnc2);
          const f32x2 pv = f32x2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
          sacc[kb][r] = pv.x, sacc[kb][r + 1] = pv.y;
          rs2[(r >> 1) & 1] += pv;
        }
      l = l * alpha + ((rs2[0].x + rs2[0].y) + (rs2[1].x + rs2[1].y));
      m = mn;
      if (__ballot(alpha != 1.0f) != 0ull) {  // the running maximum settles after the first tiles: skip the 32 rescales
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[j][r] *= alpha;
      }
    } else {
      // ---- dS^T = P^T * (dP^T - delta), P^T = exp(scale*S^T - lse)
      f32x16 dpacc[2] = {zero16(), zero16()};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) dpacc[kb] = MFMA(row_frag(vt, fo, kb, kk), dof[kk], dpacc[kb]);
      {   // on register pairs (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32), as the forward's softmax
        const f32x2 c2 = f32x2{c, c}, nl2 = f32x2{-lse_l, -lse_l}, nd2 = f32x2{-delta_l, -delta_l};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 x = __builtin_elementwise_fma(f32x2{sacc[kb][r], sacc[kb][r + 1]}, c2, nl2);
            const f32x2 pv = f32x2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
            const f32x2 ds = pv * (f32x2{dpacc[kb][r], dpacc[kb][r + 1]} + nd2);
            sacc[kb][r] = ds.x, sacc[kb][r + 1] = ds.y;
          }
      }
      if (tail) {  asm volatile("" ::: "memory");  // keeps this a real branch; wave-uniform: clamped duplicate keys of the ragged last tile contribute nothing
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (t * TROWS + kb * 32 + acc_row(r, h) >= nk_loop) sacc[kb][r] = 0.f;
      }
    }
    // ---- second product: acc^T[col, query] += tile^T[col x key] * X[key x query]
    const char* t2 = DQ ? kt : vt;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pb = acc_frag(sacc[kb], s);
#pragma unroll
        for (int j = 0; j < 2; ++j) oacc[j] = MFMA(tr_frag(t2, fo, kb, s, j), pb, oacc[j]);
      }
    buf = nbuf;
  }
  if constexpr (DQ) {
    if (cls_key) {   // this wave's 32 queries -> contribution to dK[cls] (unscaled) and dV[cls]
      __syncthreads();   // every wave is done with the K/V stages: the LDS becomes reduction scratch
      float* red = reinterpret_cast<float*>(smem) + wave * (32 * 65);
      float* scr = p.cls_scratch + ((long)b * p.H + hh) * 192;
      float wp = p_cls, wd = ds_cls;   // (rank-1 form: every lane holds its query's values)
      if (!qvalid) wp = 0.f, wd = 0.f;
      cls_partial(wd, qf, red, lane, scr + 64);    // dK[cls] += sum_q dS[q, cls] q[q, :]
      cls_partial(wp, dof, red, lane, scr + 128);  // dV[cls] += sum_q P[q, cls] dO[q, :]
    }
  }
  // ---- epilogue: lane = query, registers = output columns acc_row(r, h) + 32 j
  float mult;
  if (!DQ) {
    l = half_sum(l);
    mult = 1.f / l;
    if (qvalid && h == 0 && p.lse) p.lse[((long)b * p.H + hh) * nq + qi] = m * p.scale + __logf(l);
  } else {
    mult = p.scale;
  }
  if (qvalid) {
    bf16_t* out = DQ ? (bf16_t*)p.dq : (bf16_t*)p.o;
    const long ldo = DQ ? p.ld_dq : p.ldo;
#pragma unroll
    for (int j = 0; j < 2; ++j) store_block_rows(oacc[j], mult, out + qrow * ldo + col0 + 32 * j, h);
  }
}

// ------------------------------------------------------------------------------------------------------ dK / dV
template <int NW = 4>
__global__ void __launch_bounds__(NW * 64, 2) k_attn_bf16_dkv(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 3 stages x (Q tile, dO tile, lse[64], delta[64] (+dummy))
  constexpr int STAGE = 2 * TILE_BYTES + 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = p.nq_main + p.nq_extra, nk = p.nk_main + p.nk_extra;
  int bx, bh;
  xcd_map(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, p.xcd, bx, bh);
  // the [cls] key has no block here: its dK / dV are gathered by the dQ kernel (cls_partial); this kernel gathers dQ[cls]
  const bool cls_query = NW == 4 && p.nq_extra == 1 && p.cls_scratch != nullptr;
  float ds_cls = 0.f;
  const int b = bh / p.H, hh = bh % p.H;
  const int col0 = hh * 64;
  const int h = lane >> 5;
  const FragOff fo = frag_offsets(lane);
  const int k0 = bx * (NW * 32) + wave * 32;
  const int ki = k0 + (lane & 31);
  const bool kvalid = ki < nk;
  const long krow = tok_row(b, kvalid ? ki : nk - 1, p.nk_main, p.B);
  bf16x8 kf[4], vf[4];
  load_stationary((const bf16_t*)p.k, p.ldk, krow, col0, h, kf);
  load_stationary((const bf16_t*)p.v, p.ldv, krow, col0, h, vf);
  const bf16_t* Qb = (const bf16_t*)p.q;
  const bf16_t* Ob = (const bf16_t*)p.dout;
  const float* lse_g = p.lse + ((long)b * p.H + hh) * nq;
  const float* del_g = p.delta + ((long)b * p.H + hh) * nq;
  const float c = p.scale * LOG2E;
  f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
  const int nq_loop = cls_query ? p.nq_main : nq;
  const int nt = (nq_loop + TROWS - 1) / TROWS;
  unsigned qoff[8 / NW], ooff[8 / NW];   // (see k_attn_bf16_q)
  tile_offsets<NW>(p.ldq, wave, lane, qoff);
  tile_offsets<NW>(p.ld_do, wave, lane, ooff);
  const int n_int = min(nq_loop, p.nq_main) / TROWS;
  const bool span_ok = 64l * p.ldq * 2 < (1l << 31) && 64l * p.ld_do * 2 < (1l << 31);
  auto stage = [&](int buf, int t) {
    char* qt = smem + buf * STAGE;
    if (t < n_int && span_ok) {
      const long row0 = (long)b * p.nq_main + (long)t * TROWS;
      stage_tile_fast<NW>(Qb + row0 * p.ldq + col0, qoff, qt, wave);
      stage_tile_fast<NW>(Ob + row0 * p.ld_do + col0, ooff, qt + TILE_BYTES, wave);
    } else {
      stage_tile<NW, true>(Qb, p.ldq, col0, b, t * TROWS, nq_loop, p.nq_main, p.B, qt, wave, lane);
      stage_tile<NW, true>(Ob, p.ld_do, col0, b, t * TROWS, nq_loop, p.nq_main, p.B, qt + TILE_BYTES, wave, lane);
    }
    int qq = t * TROWS + lane;
    if (qq > nq_loop - 1) qq = nq_loop - 1;
    // one 256-byte piece per wave: wave 0 -> lse, wave 1 -> delta, waves 2,3 -> scratch (keeps vmcnt uniform)
    const float* src = (wave & 1) ? del_g : lse_g;
    glds4_raw(src + qq, qt + 2 * TILE_BYTES + wave * 256);
  };
  stage(0, 0);
  if (cls_query) {   // the [cls] query in rank-1 form: dV^T += P dO_cls, dK^T += dS q_cls; dS kept for dQ[cls]
    const long crow = tok_row(b, p.nq_main, p.nq_main, p.B);
    bf16x8 qcf[4], gcf[4];
    load_stationary(Qb, p.ldq, crow, col0, h, qcf);
    load_stationary(Ob, p.ld_do, crow, col0, h, gcf);
    const float s_e = dot_frag(kf, qcf), dp_e = dot_frag(vf, gcf);
    const float p_e = __builtin_amdgcn_exp2f(fmaf(s_e, c, -lse_g[p.nq_main] * LOG2E));
    ds_cls = p_e * (dp_e + del_g[p.nq_main]);   // (p.delta holds -delta)
    float oc[2][16];
    load_outcols(Ob, p.ld_do, crow, col0, h, oc);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dv[j][r] = p_e * oc[j][r];
    load_outcols(Qb, p.ldq, crow, col0, h, oc);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dk[j][r] = ds_cls * oc[j][r];
  }
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(kf[kk]), "v"(vf[kk]));   // (landed before the loop: see k_attn_bf16_q)
  asm volatile("" ::"v"(ds_cls));
#pragma unroll
  for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(dk[j]), "v"(dv[j]));
  int buf = 0;
  for (int t = 0; t < nt; ++t) {  // three stages, one barrier per tile (see k_attn_bf16_q; requesting the next tile between the MFMAs
                                   // measured no gain here - 91.6 / 92.9 us per layer with / without -, so the request stays at the top)
    const int nbuf = buf == 2 ? 0 : buf + 1;
    if (t + 1 < nt) {
      stage(nbuf, t + 1);
      if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const char* qt = smem + buf * STAGE;
    const char* ot = qt + TILE_BYTES;
    const float* lse_s = reinterpret_cast<const float*>(qt + 2 * TILE_BYTES);
    const float* del_s = lse_s + 64;
    const bool tail = (t == nt - 1) && (nq_loop % TROWS != 0);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 sacc = zero16(), dpacc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {   // dP starts from -delta of its query row (register r <-> row acc_row(r, h))
        const float4 nd = *reinterpret_cast<const float4*>(del_s + qb * 32 + 8 * g + 4 * h);
        dpacc[4 * g] = nd.x, dpacc[4 * g + 1] = nd.y, dpacc[4 * g + 2] = nd.z, dpacc[4 * g + 3] = nd.w;
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        sacc = MFMA(row_frag(qt, fo, qb, kk), kf[kk], sacc);
        dpacc = MFMA(row_frag(ot, fo, qb, kk), vf[kk], dpacc);
      }
      f32x16 pacc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 ls = *reinterpret_cast<const float4*>(lse_s + qb * 32 + 8 * g + 4 * h);
        const f32x2 c2 = f32x2{c, c}, nl2e = f32x2{-LOG2E, -LOG2E};
        const f32x2 lsv[2] = {f32x2{ls.x, ls.y} * nl2e, f32x2{ls.z, ls.w} * nl2e};
#pragma unroll
        for (int e = 0; e < 2; ++e) {   // register pairs: v_pk_fma_f32 / v_pk_mul_f32
          const int r = 4 * g + 2 * e;
          const f32x2 x = __builtin_elementwise_fma(f32x2{sacc[r], sacc[r + 1]}, c2, lsv[e]);
          const f32x2 pv = f32x2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
          const f32x2 ds = pv * f32x2{dpacc[r], dpacc[r + 1]};
          pacc[r] = pv.x, pacc[r + 1] = pv.y;
          sacc[r] = ds.x, sacc[r + 1] = ds.y;
        }
      }
      if (tail) {  asm volatile("" ::: "memory");  // keeps this a real branch; wave-uniform: clamped duplicate queries of the ragged last tile contribute nothing
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * TROWS + qb * 32 + acc_row(r, h) >= nq_loop) pacc[r] = 0.f, sacc[r] = 0.f;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pb = acc_frag(pacc, s), dsb = acc_frag(sacc, s);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          dv[j] = MFMA(tr_frag(ot, fo, qb, s, j), pb, dv[j]);
          dk[j] = MFMA(tr_frag(qt, fo, qb, s, j), dsb, dk[j]);
        }
      }
    }
    buf = nbuf;
  }
  if (cls_query) {   // this wave's 32 keys -> contribution to dQ[cls] (unscaled)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem) + wave * (32 * 65);
    float wd = ds_cls;   // (rank-1 form: every lane holds its key's value)
    if (!kvalid) wd = 0.f;
    cls_partial(wd, kf, red, lane, p.cls_scratch + ((long)b * p.H + hh) * 192);
  }
  if (kvalid) {
    bf16_t* odk = (bf16_t*)p.dk;
    bf16_t* odv = (bf16_t*)p.dv;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      store_block_rows(dk[j], p.scale, odk + krow * p.ld_dk + col0 + 32 * j, h);
      store_block_rows(dv[j], 1.0f, odv + krow * p.ld_dv + col0 + 32 * j, h);
    }
  }
}

static bool aligned_ok(const vfm_attn_desc* d, bool bwd) {
  auto ok = [](const void* ptr, long ld) { return ((uintptr_t)ptr % 16 == 0) && (ld % 8 == 0); };
  bool r = ok(d->q, d->ldq) && ok(d->k, d->ldk) && ok(d->v, d->ldv) && ok(d->o, d->ldo);
  if (bwd) r = r && ok(d->dout, d->ld_do) && ok(d->dq, d->ld_dq) && ok(d->dk, d->ld_dk) && ok(d->dv, d->ld_dv);
  return r;
}

// 2-wave blocks (64 stationary positions) when the 4-wave grid would give the CUs at most ~1.5 blocks: half-size blocks then
// balance better (2049 tokens x 16 heads, the coarse eval pass: 272 blocks of 128 = two for 16 CUs and one for the rest, against
// 528 of 64 = two or three of half the size everywhere; three fit a CU).  vfm_tune("attn_short_grid").
int g_attn_short_grid = 256;   // (384 measured neutral on the eval leg: 16.6 ms/img either way)
static bool short_grid(const vfm_attn_desc* d, int n) { return (long)cdiv(n, 128) * d->B * d->H < g_attn_short_grid; }

int g_attn_il = 3;        // vfm_tune("attn_il")
int g_attn_xcd = 1;       // vfm_tune("attn_xcd"): 0 = plain (block, pair) order (A/B of the XCD-local order)
int g_attn_fwd64 = 0;     // vfm_tune("attn_fwd64"): 1 = use the experimental 64-queries-per-wave forward (attention_fwd64.hip) where it fits
#ifdef VFM_EXPERIMENTAL_FWD64
bool vfm_attn_fwd64_launch(const vfm_attn_desc* d, const AttnP& p, hipStream_t s);
#else   // not built (vfmseg_amd/csrc/build.py EXPERIMENTAL): the knob is accepted and ignored
static bool vfm_attn_fwd64_launch(const vfm_attn_desc*, const AttnP&, hipStream_t) { return false; }
#endif
int g_attn_lds_pad = 0;   // vfm_tune("attn_lds_pad"): extra dynamic LDS per forward block (occupancy experiments)
int vfm_attn_bf16_fwd_impl(const vfm_attn_desc* d, hipStream_t s) {
  VFM_CHECK(aligned_ok(d, false), VFM_E_ALIGN, "vfm_attn_fwd(bf16): operands must be 16-byte aligned, ld %% 8 == 0");
  const AttnP p = to_p(d);
  const int nq = d->nq_main + d->nq_extra;
  if (g_attn_fwd64 && vfm_attn_fwd64_launch(d, p, s)) {
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  const size_t shm = 6 * TILE_BYTES + g_attn_lds_pad;
  static int attr_pad = -1;
  if (attr_pad != g_attn_lds_pad) {
    (void)hipFuncSetAttribute((const void*)k_attn_bf16_q<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr_pad = g_attn_lds_pad;
  }
  if (short_grid(d, nq)) hipLaunchKernelGGL((k_attn_bf16_q<false, 2>), dim3(cdiv(nq, 64), d->B * d->H), dim3(128), shm, s, p);
  else hipLaunchKernelGGL((k_attn_bf16_q<false, 4>), dim3(cdiv(nq, 128), d->B * d->H), dim3(256), shm, s, p);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ([cls], [cls]) pair + fp32 scratch -> the bf16 [cls] rows of dq / dk / dv; the scratch is left zeroed for the next call.
// One 64-lane block per (image, head): lane = column.
__global__ void __launch_bounds__(64) k_attn_cls_finish(AttnP p) {
  const int bh = blockIdx.x, b = bh / p.H, hh = bh % p.H, lane = threadIdx.x;
  const int nq = p.nq_main + p.nq_extra;
  const long row = tok_row(b, p.nq_main, p.nq_main, p.B);
  const int col = hh * 64 + lane;
  const float qc = bf16_to_f32(((const bf16_t*)p.q)[row * p.ldq + col]), kc = bf16_to_f32(((const bf16_t*)p.k)[row * p.ldk + col]);
  const float vc = bf16_to_f32(((const bf16_t*)p.v)[row * p.ldv + col]), gc = bf16_to_f32(((const bf16_t*)p.dout)[row * p.ld_do + col]);
  const float s = wave_sum(qc * kc) * p.scale, dp = wave_sum(gc * vc);
  const float lse = p.lse[(long)bh * nq + p.nq_main], ndelta = p.delta[(long)bh * nq + p.nq_main];   // (p.delta holds -delta)
  const float pv = __expf(s - lse), ds = pv * (dp + ndelta);
  float* scr = p.cls_scratch + (long)bh * 192;
  ((bf16_t*)p.dq)[row * p.ld_dq + col] = f32_to_bf16((scr[lane] + ds * kc) * p.scale);
  ((bf16_t*)p.dk)[row * p.ld_dk + col] = f32_to_bf16((scr[64 + lane] + ds * qc) * p.scale);
  ((bf16_t*)p.dv)[row * p.ld_dv + col] = f32_to_bf16(scr[128 + lane] + pv * gc);
  scr[lane] = 0.f, scr[64 + lane] = 0.f, scr[128 + lane] = 0.f;
}

int vfm_attn_bf16_bwd_impl(const vfm_attn_desc* d, hipStream_t s) {
  VFM_CHECK(aligned_ok(d, true), VFM_E_ALIGN, "vfm_attn_bwd(bf16): operands must be 16-byte aligned, ld %% 8 == 0");
  AttnP p = to_p(d);
  const int nq = d->nq_main + d->nq_extra, nk = d->nk_main + d->nk_extra;
  // one extra [cls] token on both sides of a 4-wave grid (the ViT backbones): its rows are gathered from the regular blocks into
  // the fp32 scratch that follows delta[B, H, nq] in the workspace (192 floats per (image, head), zero on entry, zeroed again
  // by the finishing kernel) - no VALU blocks of their own
  const bool cls = d->nq_extra == 1 && d->nk_extra == 1 && d->nq_main == d->nk_main && (d->nq_main % 128) == 0 && !short_grid(d, nq);
  p.cls_scratch = cls ? d->delta + (long)d->B * d->H * nq : nullptr;
  // (any other shape with extra tokens runs them through the ragged last blocks / tiles of the regular path)
  if (short_grid(d, nq)) hipLaunchKernelGGL((k_attn_bf16_q<true, 2>), dim3(cdiv(nq, 64), d->B * d->H), dim3(128), 6 * TILE_BYTES, s, p);
  else hipLaunchKernelGGL((k_attn_bf16_q<true, 4>), dim3(cdiv(cls ? d->nq_main : nq, 128), d->B * d->H), dim3(256), 6 * TILE_BYTES, s, p);
  if (short_grid(d, nk)) hipLaunchKernelGGL((k_attn_bf16_dkv<2>), dim3(cdiv(nk, 64), d->B * d->H), dim3(128), 3 * (2 * TILE_BYTES + 1024), s, p);
  else hipLaunchKernelGGL((k_attn_bf16_dkv<4>), dim3(cdiv(cls ? d->nk_main : nk, 128), d->B * d->H), dim3(256), 3 * (2 * TILE_BYTES + 1024), s, p);
  if (cls) hipLaunchKernelGGL(k_attn_cls_finish, dim3(d->B * d->H), dim3(64), 0, s, p);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
