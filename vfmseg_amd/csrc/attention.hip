// vfm_attn_fwd / vfm_attn_bwd dispatch: exact-fp32 kernels (attention_f32.hip) or bf16 MFMA flash kernels
// (attention_bf16.hip).
#include "common.h"

int vfm_attn_f32_fwd_impl(const vfm_attn_desc* d, hipStream_t s);
int vfm_attn_f32_bwd_impl(const vfm_attn_desc* d, hipStream_t s);
int vfm_attn_bf16_fwd_impl(const vfm_attn_desc* d, hipStream_t s);
int vfm_attn_bf16_bwd_impl(const vfm_attn_desc* d, hipStream_t s);

static int check(const vfm_attn_desc* d, bool bwd) {
  VFM_CHECK(d && d->q && d->k && d->v && d->o, VFM_E_INVAL, "vfm_attn: null operand");
  VFM_CHECK(d->B > 0 && d->H > 0 && d->d > 0, VFM_E_SHAPE, "vfm_attn: B/H/d");
  VFM_CHECK(d->nq_extra >= 0 && d->nq_extra <= 1 && d->nk_extra >= 0 && d->nk_extra <= 1, VFM_E_SHAPE, "vfm_attn: extra tokens must be 0 or 1");
  VFM_CHECK(d->nq_main + d->nq_extra > 0 && d->nk_main + d->nk_extra > 0, VFM_E_SHAPE, "vfm_attn: empty sequence");
  VFM_CHECK(d->dt == VFM_F32 || d->dt == VFM_BF16, VFM_E_INVAL, "vfm_attn: dtype");
  const long hd = (long)d->H * d->d;
  VFM_CHECK(d->ldq >= hd && d->ldk >= hd && d->ldv >= hd && d->ldo >= hd, VFM_E_SHAPE, "vfm_attn: leading dims");
  if (bwd) {
    VFM_CHECK(d->dout && d->dq && d->dk && d->dv && d->lse && d->delta, VFM_E_INVAL, "vfm_attn_bwd: null operand");
  }
  return VFM_OK;
}

extern "C" int vfm_attn_fwd(const vfm_attn_desc* d, void* stream) {
  int rc = check(d, false);
  if (rc) return rc;
  if (d->dt == VFM_BF16 && d->d == 64) return vfm_attn_bf16_fwd_impl(d, (hipStream_t)stream);
  return vfm_attn_f32_fwd_impl(d, (hipStream_t)stream);
}
extern "C" int vfm_attn_bwd(const vfm_attn_desc* d, void* stream) {
  int rc = check(d, true);
  if (rc) return rc;
  if (d->dt == VFM_BF16 && d->d == 64) return vfm_attn_bf16_bwd_impl(d, (hipStream_t)stream);
  return vfm_attn_f32_bwd_impl(d, (hipStream_t)stream);
}
