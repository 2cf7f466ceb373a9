// Shared by the attention kernels: parameter block and the token-row map of the cls-last layout.
#pragma once
#include "common.h"

struct AttnP {
  const void* q; const void* k; const void* v; void* o;
  int dt; long ldq, ldk, ldv, ldo;
  int B, H;
  int nq_main, nq_extra, nk_main, nk_extra;
  float scale;
  float* lse;
  const void* dout; long ld_do;
  void* dq; void* dk; void* dv; long ld_dq, ld_dk, ld_dv;
  float* delta;
  float* cls_scratch;   // backward: fp32 [B*H][3][64] partial dq / dk / dv rows of the extra ([cls]) token, or null
  int xcd;              // 1: all blocks of an (image, head) pair on one XCD (vfm_tune "attn_xcd", default 1)
  int il;               // vfm_tune("attn_il"): bit 0 / 1 = the forward / dQ kernel requests its next tile between the MFMAs of Q K^T (attention_bf16.hip)
};

__device__ __forceinline__ long tok_row(int b, int i, int n_main, int B) {
  return i < n_main ? (long)b * n_main + i : (long)B * n_main + b;  // the [cls] token lives after all patch tokens
}


static inline AttnP to_p(const vfm_attn_desc* d) {
  AttnP p;
  p.q = d->q; p.k = d->k; p.v = d->v; p.o = d->o;
  p.dt = d->dt; p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.ldo = d->ldo;
  p.B = d->B; p.H = d->H;
  p.nq_main = d->nq_main; p.nq_extra = d->nq_extra; p.nk_main = d->nk_main; p.nk_extra = d->nk_extra;
  p.scale = d->scale; p.lse = d->lse;
  p.dout = d->dout; p.ld_do = d->ld_do;
  p.dq = d->dq; p.dk = d->dk; p.dv = d->dv; p.ld_dq = d->ld_dq; p.ld_dk = d->ld_dk; p.ld_dv = d->ld_dv;
  p.delta = d->delta;
  p.cls_scratch = nullptr;
  extern int g_attn_xcd;
  p.xcd = g_attn_xcd;
  extern int g_attn_il;
  p.il = g_attn_il;
  return p;
}

