// bf16 MFMA flash attention (forward + backward).  Placeholder dispatch to the generic kernels until the MFMA
// kernels below are complete.
#include "common.h"

int vfm_attn_f32_fwd_impl(const vfm_attn_desc* d, hipStream_t s);
int vfm_attn_f32_bwd_impl(const vfm_attn_desc* d, hipStream_t s);

int vfm_attn_bf16_fwd_impl(const vfm_attn_desc* d, hipStream_t s) { return vfm_attn_f32_fwd_impl(d, s); }
int vfm_attn_bf16_bwd_impl(const vfm_attn_desc* d, hipStream_t s) { return vfm_attn_f32_bwd_impl(d, s); }
