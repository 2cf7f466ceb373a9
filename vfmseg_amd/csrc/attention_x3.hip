// Split-bf16 ("bf16 x 3") flash attention forward for head dim 64, gfx950: fp32 in, fp32 out, every product on the bf16 MFMA.
//
// The parity configuration's attention (exact fp32: one thread per query on the VALU, attention_f32.hip) ran at 22 TFLOP/s and took
// 92 of the 127 ms of a 1024^2 prediction once the GEMMs had moved to split operands.  Same idea here: an fp32 operand x is carried as
// hi = bf16(x), lo = bf16(x - hi) (16 significant bits), and a product a b as a_hi b_hi + a_hi b_lo + a_lo b_hi with fp32 accumulation:
//     S^T   = Kh Qh^T + Kl Qh^T + Kh Ql^T                   (Q split in registers from the fp32 rows; K arrives split: vfm_split3)
//     O^T  += Vh^T Ph^T + Vl^T Ph^T + Vh^T Pl^T             (P = exp2(...) in fp32, split in registers; V arrives split)
// softmax statistics, the exponentials and the output stay fp32.  Structure of attention_bf16.hip's forward (32 queries per wave on the
// MFMA lane index, scores' accumulator = the second product's B operand, 64-key LDS tiles by LDS-DMA, lazy rescale), without its
// [cls] special cases: the extra token is simply the ragged last tile / block.  Two stages x (Kh, Kl, Vh, Vl) = 64 KiB: two blocks per CU.
// k / v: bf16 [rows, ld] with the hi half of head h at columns h*64.., the lo half at lo_off + h*64.. (lo_off = H*64: the [hi | lo | hi]
// layout vfm_split3(pattern 1) writes for K = H*64).
#include "attn_bf16_dev.h"

namespace {

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hi[e] = (vfm_h)x[e];
    lo[e] = (vfm_h)(x[e] - (float)hi[e]);
  }
}

// o3 != null: the output is (also) written as the split-bf16 A operand of the projection GEMM that consumes it ([hi | hi | lo], planes
// `plane` columns apart: vfm_split3 pattern 0) - the fp32 copy is skipped when p.o is null
__global__ void __launch_bounds__(256, 2) k_attn_x3_fwd(AttnP p, int lo_off, bf16_t* o3, long ld3, long plane) {
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
  bf16x8 qh[4], ql[4];
  {
    const float* qp = (const float*)p.q + qrow * p.ldq + col0 + 8 * h;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float4 a = *reinterpret_cast<const float4*>(qp + 16 * kk), c4 = *reinterpret_cast<const float4*>(qp + 16 * kk + 4);
      const float x[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
      split8(x, qh[kk], ql[kk]);
    }
  }
  const bf16_t* Kb = (const bf16_t*)p.k;
  const bf16_t* Vb = (const bf16_t*)p.v;
  const float c = p.scale * LOG2E;
  f32x16 oacc[2] = {zero16(), zero16()};
  float m = -INFINITY, l = 0.f;
  const int nt = (nk + TROWS - 1) / TROWS;
  auto stage = [&](int buf, int t) {
    char* st = smem + buf * 4 * TILE_BYTES;
    stage_tile<4>(Kb, p.ldk, col0, b, t * TROWS, nk, p.nk_main, p.B, st, wave, lane);
    stage_tile<4>(Kb, p.ldk, col0 + lo_off, b, t * TROWS, nk, p.nk_main, p.B, st + TILE_BYTES, wave, lane);
    stage_tile<4>(Vb, p.ldv, col0, b, t * TROWS, nk, p.nk_main, p.B, st + 2 * TILE_BYTES, wave, lane);
    stage_tile<4>(Vb, p.ldv, col0 + lo_off, b, t * TROWS, nk, p.nk_main, p.B, st + 3 * TILE_BYTES, wave, lane);
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
    f32x16 sacc[2] = {zero16(), zero16()};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 fh = row_frag(kth, kb, kk, lane), fl = row_frag(ktl, kb, kk, lane);
        sacc[kb] = MFMA(fl, qh[kk], sacc[kb]);   // small terms first
        sacc[kb] = MFMA(fh, ql[kk], sacc[kb]);
        sacc[kb] = MFMA(fh, qh[kk], sacc[kb]);
      }
    if ((t == nt - 1) && (nk % TROWS != 0)) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * TROWS + kb * 32 + acc_row(r, h) >= nk) sacc[kb][r] = -INFINITY;
    }
    float mx;
    {
      float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) m4[r & 3] = fmaxf(m4[r & 3], sacc[kb][r]);
      mx = half_max(fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3])));
    }
    // lazy rescale as in the bf16 kernel (p <= 2^8: exact enough for the 16-bit split of P and the fp32 sums)
    const bool need = (mx - m) * c > 8.0f;
    const float mn = need ? mx : m;
    const float alpha = need ? __builtin_amdgcn_exp2f((m - mn) * c) : 1.0f;
    const float mnc = mn * c;
    float rs4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], c, -mnc));
        sacc[kb][r] = pv;
        rs4[r & 3] += pv;
      }
    l = l * alpha + ((rs4[0] + rs4[1]) + (rs4[2] + rs4[3]));
    m = mn;
    if (__ballot(alpha != 1.0f) != 0ull) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] *= alpha;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = sacc[kb][8 * s + e];
        bf16x8 ph, pl;
        split8(x, ph, pl);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 vh = tr_frag(vth, kb, s, j, lane), vl = tr_frag(vtl, kb, s, j, lane);
          oacc[j] = MFMA(vl, ph, oacc[j]);
          oacc[j] = MFMA(vh, pl, oacc[j]);
          oacc[j] = MFMA(vh, ph, oacc[j]);
        }
      }
    __builtin_amdgcn_s_barrier();   // every wave is done with this stage before the next iteration's DMA overwrites the other one's successor
  }
  l = half_sum(l);
  const float mult = 1.f / l;
  if (qvalid && h == 0 && p.lse) p.lse[((long)b * p.H + hh) * nq + qi] = m * p.scale + __logf(l);
  if (qvalid) {
    float* out = p.o ? (float*)p.o + qrow * p.ldo + col0 + 4 * h : nullptr;
    bf16_t* out3 = o3 ? o3 + qrow * ld3 + col0 + 4 * h : nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v0 = oacc[j][4 * g] * mult, v1 = oacc[j][4 * g + 1] * mult, v2 = oacc[j][4 * g + 2] * mult, v3 = oacc[j][4 * g + 3] * mult;
        if (out) *reinterpret_cast<float4*>(out + 32 * j + 8 * g) = make_float4(v0, v1, v2, v3);
        if (out3) {
          const bf16_t h0 = f32_to_bf16(v0), h1 = f32_to_bf16(v1), h2 = f32_to_bf16(v2), h3 = f32_to_bf16(v3);
          const ushort4 hi = {h0, h1, h2, h3};
          const ushort4 lo = {f32_to_bf16(v0 - bf16_to_f32(h0)), f32_to_bf16(v1 - bf16_to_f32(h1)), f32_to_bf16(v2 - bf16_to_f32(h2)),
                              f32_to_bf16(v3 - bf16_to_f32(h3))};
          bf16_t* q3 = out3 + 32 * j + 8 * g;
          *reinterpret_cast<ushort4*>(q3) = hi;
          *reinterpret_cast<ushort4*>(q3 + plane) = hi;
          *reinterpret_cast<ushort4*>(q3 + 2 * plane) = lo;
        }
      }
  }
}

}  // namespace

static int attn_fwd_x3_impl(const vfm_attn_desc* d, long lo_off, void* o3, long ld3, long plane, void* stream) {
  auto ok = [](const void* ptr, long ld, long al) { return ((uintptr_t)ptr % 16 == 0) && (ld % al == 0); };
  VFM_CHECK(d->d == 64 && d->dt == VFM_F32, VFM_E_UNSUPPORTED, "vfm_attn_fwd_x3: head dim 64, fp32 q / o");
  VFM_CHECK((d->o || o3) && (!o3 || ((uintptr_t)o3 % 8 == 0 && plane >= (long)d->H * 64 && plane % 4 == 0 && ld3 >= 3 * plane && ld3 % 4 == 0)), VFM_E_ALIGN,
            "vfm_attn_fwd_x3: o3 [rows, >= 3 plane] bf16 with plane >= H*64, 8-byte aligned (or an fp32 o)");
  VFM_CHECK(ok(d->q, d->ldq, 4) && (!d->o || ok(d->o, d->ldo, 4)) && ok(d->k, d->ldk, 8) && ok(d->v, d->ldv, 8) && lo_off % 8 == 0 && lo_off >= (long)d->H * 64,
            VFM_E_ALIGN, "vfm_attn_fwd_x3: operands must be 16-byte aligned; k / v are split bf16 operands (hi at column h*64, lo at lo_off + h*64)");
  VFM_CHECK(d->nq_main + d->nq_extra > 0 && d->nk_main + d->nk_extra > 0, VFM_E_SHAPE, "vfm_attn_fwd_x3: empty sequence");
  const AttnP p = to_p(d);
  const int nq = d->nq_main + d->nq_extra;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_attn_x3_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * TILE_BYTES);
    attr = true;
  }
  hipLaunchKernelGGL(k_attn_x3_fwd, dim3(cdiv(nq, 128), d->B * d->H), dim3(256), 8 * TILE_BYTES, (hipStream_t)stream, p, (int)lo_off, (bf16_t*)o3, ld3,
                     plane);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
extern "C" int vfm_attn_fwd_x3(const vfm_attn_desc* d, long lo_off, void* stream) {
  VFM_CHECK(d && d->o, VFM_E_INVAL, "vfm_attn_fwd_x3: null output");
  return attn_fwd_x3_impl(d, lo_off, nullptr, 0, 0, stream);
}
extern "C" int vfm_attn_fwd_x3_split(const vfm_attn_desc* d, long lo_off, void* o3, long ld3, long plane, void* stream) {
  VFM_CHECK(d && o3, VFM_E_INVAL, "vfm_attn_fwd_x3_split: null split output");
  return attn_fwd_x3_impl(d, lo_off, o3, ld3, plane, stream);
}
