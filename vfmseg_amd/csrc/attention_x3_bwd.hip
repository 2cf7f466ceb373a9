// Split-bf16 ("bf16 x 3") flash attention BACKWARD for head dim 64, gfx950: fp32 in, fp32 out, every product on the bf16 MFMA.
//
// Round 4.  In the bf16x3 mode (precision.py: the in-tolerance mode with a usable speed) the train step's attention backward still ran on
// the exact-fp32 VALU kernels (attention_f32.hip): 158 of the step's 202 ms (profiles/r04_bf16x3_step_profile_before.txt).  Same recipe as
// the forward (attention_x3.hip): an fp32 operand x is carried as hi = bf16(x), lo = bf16(x - hi), a product a b as
// a_hi b_hi + a_hi b_lo + a_lo b_hi with fp32 accumulation (the dropped lo x lo term is 2^-18 relative).  Structure of attention_bf16.hip's
// two backward kernels - 32 stationary positions per wave on the MFMA lane index, the first products' accumulators (P, dS in fp32) split in
// registers into the B operands of the second products - without their [cls] special cases: the extra token is simply the ragged last
// tile / block.
//   dQ    (queries stationary; K, V streamed):   S^T = K Q^T,  dP^T = V dO^T,  dS^T = P^T o (dP^T - delta),  dQ^T += K^T dS^T
//   dK/dV (keys stationary; Q, dO streamed):     S = Q K^T,    dP = dO V^T,    dV^T += dO^T P,  dK^T += Q^T dS
// Stationary rows are read as fp32 and split in registers; streamed operands arrive split (vfm_split3 pattern 1: [hi | lo | hi], the hi
// half of head h at columns h*64.., the lo half at lo_off + h*64..).  delta = rowsum(dO o O) is computed in fp32 by the dQ kernel and
// published for the dK/dV kernel that runs next.  Two stages x four 8-KiB tiles (+ lse / delta rows) = 64-66 KiB: two blocks per CU.
#include "attn_bf16_dev.h"

namespace {

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hi[e] = (vfm_h)x[e];
    lo[e] = (vfm_h)(x[e] - (float)hi[e]);
  }
}
// registers 8s..8s+7 of a 32x32 f32 accumulator -> split B operand of k-step s
__device__ __forceinline__ void acc_split(const f32x16& a, int s, bf16x8& hi, bf16x8& lo) {
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = a[8 * s + e];
  split8(x, hi, lo);
}
// the stationary operand's B fragments from an fp32 row: 4 k-steps x 8 values of row `row`, columns col0 + 16kk + 8h ..
__device__ __forceinline__ void load_split(const float* base, long ld, long row, int col0, int h, bf16x8 (&hi)[4], bf16x8 (&lo)[4]) {
  const float* rp = base + row * ld + col0 + 8 * h;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const float4 a = *reinterpret_cast<const float4*>(rp + 16 * kk), c4 = *reinterpret_cast<const float4*>(rp + 16 * kk + 4);
    const float x[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
    split8(x, hi[kk], lo[kk]);
  }
}
// acc += A_lo B_hi + A_hi B_lo + A_hi B_hi (small terms first)
__device__ __forceinline__ f32x16 mfma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x16 acc) {
  acc = MFMA(al, bh, acc);
  acc = MFMA(ah, bl, acc);
  return MFMA(ah, bh, acc);
}

struct X3P {
  const bf16_t* q3; const bf16_t* k3; const bf16_t* v3; const bf16_t* do3;   // split operands: q3 / k3 / v3 [rows, ld3], do3 [rows, ld_g]
  long ld3; int lo_off;
  long ld_g; int lo_off_g;
};

// ------------------------------------------------------------------------------------------------------ dQ
__global__ void __launch_bounds__(256, 2) k_attn_x3_dq(AttnP p, X3P x) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (Kh, Kl, Vh, Vl)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = p.nq_main + p.nq_extra, nk = p.nk_main + p.nk_extra;
  int bx, bh;
  xcd_map(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, p.xcd, bx, bh);
  const int b = bh / p.H, hh = bh % p.H;
  const int col0 = hh * 64;
  const int h = lane >> 5;
  const int qi = bx * 128 + wave * 32 + (lane & 31);
  const bool qvalid = qi < nq;
  const long qrow = tok_row(b, qvalid ? qi : nq - 1, p.nq_main, p.B);
  bf16x8 qh[4], ql[4], gh[4], gl[4];
  load_split((const float*)p.q, p.ldq, qrow, col0, h, qh, ql);
  load_split((const float*)p.dout, p.ld_do, qrow, col0, h, gh, gl);
  float delta_l;
  {   // delta = rowsum(dO o O) in fp32: each lane holds half of the row's 64 columns
    const float* gp = (const float*)p.dout + qrow * p.ld_do + col0 + 8 * h;
    const float* op = (const float*)p.o + qrow * p.ldo + col0 + 8 * h;
    float dsum = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum = fmaf(gp[16 * kk + e], op[16 * kk + e], dsum);
    delta_l = half_sum(dsum);
    if (qvalid && h == 0) p.delta[((long)b * p.H + hh) * nq + qi] = delta_l;
  }
  const float lse_l = p.lse[((long)b * p.H + hh) * nq + (qvalid ? qi : nq - 1)] * LOG2E;
  const float c = p.scale * LOG2E;
  f32x16 oacc[2] = {zero16(), zero16()};
  const int nt = (nk + TROWS - 1) / TROWS;
  auto stage = [&](int buf, int t) {
    char* st = smem + buf * 4 * TILE_BYTES;
    stage_tile<4>(x.k3, x.ld3, col0, b, t * TROWS, nk, p.nk_main, p.B, st, wave, lane);
    stage_tile<4>(x.k3, x.ld3, col0 + x.lo_off, b, t * TROWS, nk, p.nk_main, p.B, st + TILE_BYTES, wave, lane);
    stage_tile<4>(x.v3, x.ld3, col0, b, t * TROWS, nk, p.nk_main, p.B, st + 2 * TILE_BYTES, wave, lane);
    stage_tile<4>(x.v3, x.ld3, col0 + x.lo_off, b, t * TROWS, nk, p.nk_main, p.B, st + 3 * TILE_BYTES, wave, lane);
  };
  stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) {
      stage(buf ^ 1, t + 1);   // (the barrier that closed the previous iteration freed that stage)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const char* kth = smem + buf * 4 * TILE_BYTES;
    const char* ktl = kth + TILE_BYTES;
    const char* vth = kth + 2 * TILE_BYTES;
    const char* vtl = kth + 3 * TILE_BYTES;
    f32x16 sacc[2] = {zero16(), zero16()}, dpacc[2] = {zero16(), zero16()};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        sacc[kb] = mfma3(row_frag(kth, kb, kk, lane), row_frag(ktl, kb, kk, lane), qh[kk], ql[kk], sacc[kb]);
        dpacc[kb] = mfma3(row_frag(vth, kb, kk, lane), row_frag(vtl, kb, kk, lane), gh[kk], gl[kk], dpacc[kb]);
      }
    // dS^T = P^T o (dP^T - delta), P^T = exp(scale S^T - lse)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], c, -lse_l));
        sacc[kb][r] = pv * (dpacc[kb][r] - delta_l);
      }
    if ((t == nt - 1) && (nk % TROWS != 0)) {   // clamped duplicate keys of the ragged last tile contribute nothing
      asm volatile("" ::: "memory");
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * TROWS + kb * 32 + acc_row(r, h) >= nk) sacc[kb][r] = 0.f;
    }
    // dQ^T[col, query] += K^T[col x key] dS^T[key x query]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 dsh, dsl;
        acc_split(sacc[kb], s, dsh, dsl);
#pragma unroll
        for (int j = 0; j < 2; ++j) oacc[j] = mfma3(tr_frag(kth, kb, s, j, lane), tr_frag(ktl, kb, s, j, lane), dsh, dsl, oacc[j]);
      }
    __builtin_amdgcn_s_barrier();   // every wave is done with this stage before the next iteration's DMA overwrites it
  }
  if (qvalid) {
    float* out = (float*)p.dq + qrow * p.ld_dq + col0 + 4 * h;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(out + 32 * j + 8 * g) = make_float4(oacc[j][4 * g] * p.scale, oacc[j][4 * g + 1] * p.scale,
                                                                     oacc[j][4 * g + 2] * p.scale, oacc[j][4 * g + 3] * p.scale);
  }
}

// ------------------------------------------------------------------------------------------------------ dK / dV
__global__ void __launch_bounds__(256, 2) k_attn_x3_dkv(AttnP p, X3P x) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (Qh, Ql, dOh, dOl, lse[64] | delta[64] | scratch)
  constexpr int STAGE = 4 * TILE_BYTES + 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = p.nq_main + p.nq_extra, nk = p.nk_main + p.nk_extra;
  int bx, bh;
  xcd_map(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, p.xcd, bx, bh);
  const int b = bh / p.H, hh = bh % p.H;
  const int col0 = hh * 64;
  const int h = lane >> 5;
  const int ki = bx * 128 + wave * 32 + (lane & 31);
  const bool kvalid = ki < nk;
  const long krow = tok_row(b, kvalid ? ki : nk - 1, p.nk_main, p.B);
  bf16x8 kh[4], kl[4], vh[4], vl[4];
  load_split((const float*)p.k, p.ldk, krow, col0, h, kh, kl);
  load_split((const float*)p.v, p.ldv, krow, col0, h, vh, vl);
  const float* lse_g = p.lse + ((long)b * p.H + hh) * nq;
  const float* del_g = p.delta + ((long)b * p.H + hh) * nq;
  const float c = p.scale * LOG2E;
  f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
  const int nt = (nq + TROWS - 1) / TROWS;
  auto stage = [&](int buf, int t) {
    char* st = smem + buf * STAGE;
    stage_tile<4>(x.q3, x.ld3, col0, b, t * TROWS, nq, p.nq_main, p.B, st, wave, lane);
    stage_tile<4>(x.q3, x.ld3, col0 + x.lo_off, b, t * TROWS, nq, p.nq_main, p.B, st + TILE_BYTES, wave, lane);
    stage_tile<4>(x.do3, x.ld_g, col0, b, t * TROWS, nq, p.nq_main, p.B, st + 2 * TILE_BYTES, wave, lane);
    stage_tile<4>(x.do3, x.ld_g, col0 + x.lo_off_g, b, t * TROWS, nq, p.nq_main, p.B, st + 3 * TILE_BYTES, wave, lane);
    int qq = t * TROWS + lane;
    if (qq > nq - 1) qq = nq - 1;
    // one 256-byte piece per wave: wave 0 -> lse, wave 1 -> delta, waves 2, 3 -> scratch (keeps vmcnt uniform)
    const float* src = (wave & 1) ? del_g : lse_g;
    glds4(src + qq, st + 4 * TILE_BYTES + wave * 256);
  };
  stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) {
      stage(buf ^ 1, t + 1);
      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const char* qth = smem + buf * STAGE;
    const char* qtl = qth + TILE_BYTES;
    const char* gth = qth + 2 * TILE_BYTES;
    const char* gtl = qth + 3 * TILE_BYTES;
    const float* lse_s = reinterpret_cast<const float*>(qth + 4 * TILE_BYTES);
    const float* del_s = lse_s + 64;
    const bool tail = (t == nt - 1) && (nq % TROWS != 0);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 sacc = zero16(), dpacc = zero16();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        sacc = mfma3(row_frag(qth, qb, kk, lane), row_frag(qtl, qb, kk, lane), kh[kk], kl[kk], sacc);
        dpacc = mfma3(row_frag(gth, qb, kk, lane), row_frag(gtl, qb, kk, lane), vh[kk], vl[kk], dpacc);
      }
      f32x16 pacc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 ls = *reinterpret_cast<const float4*>(lse_s + qb * 32 + 8 * g + 4 * h);
        const float4 dl = *reinterpret_cast<const float4*>(del_s + qb * 32 + 8 * g + 4 * h);
        const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dlv[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[r], c, -lsv[e] * LOG2E));
          pacc[r] = pv;
          sacc[r] = pv * (dpacc[r] - dlv[e]);
        }
      }
      if (tail) {   // clamped duplicate queries of the ragged last tile contribute nothing
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * TROWS + qb * 32 + acc_row(r, h) >= nq) pacc[r] = 0.f, sacc[r] = 0.f;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 ph, pl, dsh, dsl;
        acc_split(pacc, s, ph, pl);
        acc_split(sacc, s, dsh, dsl);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          dv[j] = mfma3(tr_frag(gth, qb, s, j, lane), tr_frag(gtl, qb, s, j, lane), ph, pl, dv[j]);
          dk[j] = mfma3(tr_frag(qth, qb, s, j, lane), tr_frag(qtl, qb, s, j, lane), dsh, dsl, dk[j]);
        }
      }
    }
    __builtin_amdgcn_s_barrier();
  }
  if (kvalid) {
    float* odk = (float*)p.dk + krow * p.ld_dk + col0 + 4 * h;
    float* odv = (float*)p.dv + krow * p.ld_dv + col0 + 4 * h;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(odk + 32 * j + 8 * g) = make_float4(dk[j][4 * g] * p.scale, dk[j][4 * g + 1] * p.scale,
                                                                     dk[j][4 * g + 2] * p.scale, dk[j][4 * g + 3] * p.scale);
        *reinterpret_cast<float4*>(odv + 32 * j + 8 * g) = make_float4(dv[j][4 * g], dv[j][4 * g + 1], dv[j][4 * g + 2], dv[j][4 * g + 3]);
      }
  }
}

}  // namespace

extern "C" int vfm_attn_bwd_x3(const vfm_attn_desc* d, const void* q3, const void* k3, const void* v3, long ld3, long lo_off, const void* do3,
                               long ld_do3, long lo_off_do3, void* stream) {
  auto ok = [](const void* ptr, long ld, long al) { return ptr && ((uintptr_t)ptr % 16 == 0) && (ld % al == 0); };
  VFM_CHECK(d && d->d == 64 && d->dt == VFM_F32, VFM_E_UNSUPPORTED, "vfm_attn_bwd_x3: head dim 64, fp32 q / k / v / o / dout and gradients");
  VFM_CHECK(ok(d->q, d->ldq, 4) && ok(d->k, d->ldk, 4) && ok(d->v, d->ldv, 4) && ok(d->o, d->ldo, 4) && ok(d->dout, d->ld_do, 4) &&
                ok(d->dq, d->ld_dq, 4) && ok(d->dk, d->ld_dk, 4) && ok(d->dv, d->ld_dv, 4) && ok(q3, ld3, 8) && ok(k3, ld3, 8) && ok(v3, ld3, 8) &&
                ok(do3, ld_do3, 8) && lo_off % 8 == 0 && lo_off >= (long)d->H * 64 && lo_off_do3 % 8 == 0 && lo_off_do3 >= (long)d->H * 64 && d->lse &&
                d->delta,
            VFM_E_ALIGN, "vfm_attn_bwd_x3: operands must be 16-byte aligned; q3 / k3 / v3 (leading dimension ld3) and do3 (ld_do3) are split bf16 "
                         "operands: hi at column h*64, lo at lo_off + h*64");
  VFM_CHECK(d->nq_main + d->nq_extra > 0 && d->nk_main + d->nk_extra > 0, VFM_E_SHAPE, "vfm_attn_bwd_x3: empty sequence");
  const AttnP p = to_p(d);
  X3P x;
  x.q3 = (const bf16_t*)q3, x.k3 = (const bf16_t*)k3, x.v3 = (const bf16_t*)v3, x.do3 = (const bf16_t*)do3, x.ld3 = ld3, x.lo_off = (int)lo_off;
  x.ld_g = ld_do3, x.lo_off_g = (int)lo_off_do3;
  const int nq = d->nq_main + d->nq_extra, nk = d->nk_main + d->nk_extra;
  constexpr int SM_DQ = 8 * TILE_BYTES, SM_DKV = 2 * (4 * TILE_BYTES + 1024);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_attn_x3_dq, hipFuncAttributeMaxDynamicSharedMemorySize, SM_DQ);
    (void)hipFuncSetAttribute((const void*)k_attn_x3_dkv, hipFuncAttributeMaxDynamicSharedMemorySize, SM_DKV);
    attr = true;
  }
  hipLaunchKernelGGL(k_attn_x3_dq, dim3(cdiv(nq, 128), d->B * d->H), dim3(256), SM_DQ, (hipStream_t)stream, p, x);
  VFM_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_attn_x3_dkv, dim3(cdiv(nk, 128), d->B * d->H), dim3(256), SM_DKV, (hipStream_t)stream, p, x);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
