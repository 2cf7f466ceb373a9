// Shared GEMM epilogue (see vfm_gemm in include/vfmseg_hip.h for the order of operations).
#pragma once
#include "common.h"

struct EpiParams {
  void* C; int c_dt; long ldc;
  float alpha;
  const float* bias; long bias_mod;
  const float* colscale;
  const void* residual; int r_dt; long ldr;
  int ep_mode; const void* aux; int aux_dt; long ld_aux;
  void* C2; int c2_dt; long ldc2;
  long c_plane;   // c_dt == VFM_SPLIT3: columns between the three planes of the split-bf16 image
  int nt;   // nontemporal stores of C / C2 in the vector epilogues (large write-once outputs: see g_nt_bytes in gemm_bf16.hip)
};

static inline EpiParams make_epi(const vfm_gemm_desc* d) {
  EpiParams e;
  e.C = d->C; e.c_dt = d->c_dt; e.ldc = d->ldc;
  e.alpha = d->alpha;
  e.bias = d->bias; e.bias_mod = d->bias_mod > 0 ? d->bias_mod : d->N;
  e.colscale = d->colscale;
  e.residual = d->residual; e.r_dt = d->r_dt; e.ldr = d->ldr;
  e.ep_mode = d->ep_mode; e.aux = d->aux; e.aux_dt = d->aux_dt; e.ld_aux = d->ld_aux;
  e.C2 = d->C2; e.c2_dt = d->c2_dt; e.ldc2 = d->ldc2;
  e.c_plane = d->c_plane;
  extern long g_nt_bytes;
  const long out_bytes = d->M * d->N * (d->c_dt == VFM_BF16 ? 2 : (d->c_dt == VFM_SPLIT3 ? 6 : 4)) * (d->C2 ? 2 : 1) * (d->batch > 0 ? d->batch : 1);
  e.nt = g_nt_bytes > 0 && out_bytes >= g_nt_bytes;
  return e;
}

// one output element; zoff = batch offset into C (elements)
__device__ __forceinline__ void epi_store(const EpiParams& e, long zoff, long m, long n, float acc) {
  float v = e.alpha * acc;
  if (e.bias) v += e.bias[n % e.bias_mod];
  if (e.C2) st_any(e.C2, zoff + m * e.ldc2 + n, e.c2_dt, e.ep_mode == VFM_EP_GELU_DGELU ? gelu_grad_f(v) : v);
  switch (e.ep_mode) {
    case VFM_EP_GELU:
    case VFM_EP_GELU_DGELU: v = gelu_f(v); break;
    case VFM_EP_RELU: v = fmaxf(v, 0.f); break;
    case VFM_EP_MUL_GELU_GRAD: v *= gelu_grad_f(ld_any(e.aux, m * e.ld_aux + n, e.aux_dt)); break;
    case VFM_EP_MUL: v *= ld_any(e.aux, m * e.ld_aux + n, e.aux_dt); break;
    case VFM_EP_QGELU: v = qgelu_f(v); break;
    case VFM_EP_MUL_QGELU_GRAD: v *= qgelu_grad_f(ld_any(e.aux, m * e.ld_aux + n, e.aux_dt)); break;
    default: break;
  }
  if (e.colscale) v *= e.colscale[n];
  if (e.residual) v += ld_any(e.residual, zoff + m * e.ldr + n, e.r_dt);
  if (e.c_dt == VFM_SPLIT3) {
    bf16_t* cp = (bf16_t*)e.C + zoff + m * e.ldc + n;
    const bf16_t hi = f32_to_bf16(v);
    cp[0] = hi, cp[e.c_plane] = hi, cp[2 * e.c_plane] = f32_to_bf16(v - bf16_to_f32(hi));
    return;
  }
  st_any(e.C, zoff + m * e.ldc + n, e.c_dt, v);
}
