/* libvfmseg_hip - C ABI of the MI355X (gfx950) kernels behind the VFM-segmentation hot path.
 *
 * The reference (tpy001/VFMSeg) has no FFI of its own: its extension boundary is the mmengine
 * registry + nn.Module method signatures.  This library sits *below* that surface;
 * every entry point replaces one family of third-party CUDA kernels the reference launches through
 * torch / xformers / peft, cited per function as reference file:line.
 *
 * Conventions
 *  - every pointer is a caller-owned DEVICE pointer (e.g. torch.Tensor.data_ptr()); no ownership transfer
 *  - kernels are asynchronous on `stream` (a hipStream_t passed as void*), never synchronise, never allocate
 *  - return 0 on success, negative VFM_E_* otherwise; vfm_last_error() gives a thread-local message
 *  - matrices are row-major with explicit leading dimensions in ELEMENTS
 *  - dtypes: VFM_F32 / VFM_BF16 activations and weights, fp32 accumulation and statistics, int64 labels
 *  - the SAME sources are built twice: libvfmseg_hip.so (16-bit type = bf16, the throughput configuration) and
 *    libvfmseg_hip_f16.so (16-bit type = IEEE fp16: the autocast dtype of the reference's `--amp`, tools/train.py:87-102).
 *    Both export this header; in the fp16 library dtype code VFM_BF16 means fp16 storage and the MFMA is v_mfma_f32_32x32x16_f16.
 *    vfm_half_kind() tells which one was loaded.
 *  - token-major ("NHWC") activations: a feature map [B,H,W,C] is the matrix [B*H*W, C]
 */
#ifndef VFMSEG_HIP_H
#define VFMSEG_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VFM_F32 = 0, VFM_BF16 = 1, VFM_U8 = 2, VFM_I64 = 3,
       VFM_SPLIT3 = 4 /* only as vfm_gemm's c_dt (bf16 inputs): C is the split-bf16 image of the result - bf16 [M, >= 3 c_plane], hi at column n,
                         hi again at c_plane + n, lo = bf16(v - hi) at 2 c_plane + n (vfm_split3 pattern 0): the A operand of the next GEMM of the
                         bf16x3 mode, written by the producer instead of by a vfm_split3 pass */ };
enum { VFM_OK = 0, VFM_E_INVAL = -1, VFM_E_SHAPE = -2, VFM_E_ALIGN = -3, VFM_E_HIP = -4, VFM_E_UNSUPPORTED = -5 };
/* GEMM epilogue modes */
enum { VFM_EP_NONE = 0, VFM_EP_GELU = 1, VFM_EP_RELU = 2, VFM_EP_MUL_GELU_GRAD = 3, VFM_EP_MUL = 4,
       VFM_EP_QGELU = 5 /* CLIP QuickGELU x*sigmoid(1.702x), clip.py:18-20 */, VFM_EP_MUL_QGELU_GRAD = 6,
       VFM_EP_GELU_DGELU = 7 /* C = gelu(v) and C2 = gelu'(v) (NOT the pre-activation): the backward GEMM then uses VFM_EP_MUL */ };
/* activation fused into norm kernels */
enum { VFM_ACT_NONE = 0, VFM_ACT_GELU = 1, VFM_ACT_RELU = 2, VFM_ACT_QGELU = 3 };

const char* vfm_last_error(void);
int vfm_abi_version(void);
int vfm_half_kind(void); /* 0: the 16-bit type of this library is bf16, 1: IEEE fp16 */

/* ---- elementwise / layout ------------------------------------------------------------------ */
/* dst[r,c] = src[r,c] * (colscale ? colscale[c] : 1)          (casts; LayerScale prologue for dgrad) */
int vfm_cast(const void* src, int src_dt, long ld_src, void* dst, int dst_dt, long ld_dst, long rows, long cols,
             const float* colscale, void* stream);
/* Split-bf16 ("bf16 x 3") form of an fp32 GEMM operand: src[rows, K] fp32 with element strides (stride_r, stride_c) -> dst[rows, 3 Kp] bf16,
 * Kp = ceil64(K): pattern 0 = [hi | hi | lo] (A side), 1 = [hi | lo | hi] (B side), hi = bf16(x), lo = bf16(x - hi), zero pad columns.
 * vfm_gemm over K' = 3 Kp of two such operands is the fp32 product to ~2^-16 relative per term (fp32 accumulate) at the bf16 MFMA rate / 3:
 * the parity configuration's GEMMs without the 157-TFLOP/s fp32 MFMA (replaces cuBLAS fp32 Linear of the reference's CPU / fp32 path;
 * rein/models/backbones/dino_layers/attention.py:51-58, mlp.py:34-40 in fp32). */
int vfm_split3(const float* src, long stride_r, long stride_c, void* dst, long ld_dst, long rows, long K, int pattern, void* stream);
/* dst[c, r] = src[r, c] for r<rows, c<cols; dst has ld_dst >= rows, columns rows..pad_rows-1 are zero-filled.
 * Used to feed weight-gradient GEMMs (reduction over tokens) to the NT GEMM. */
int vfm_transpose(const void* src, int src_dt, long ld_src, void* dst, int dst_dt, long ld_dst, long rows, long cols,
                  long pad_rows, void* stream);
/* generic 4-D strided gather/scatter + cast over the index space [n0,n1,n2,n3] (weight packing / un-packing):
 * dst[i0*d0+i1*d1+i2*d2+i3*d3] = src[i0*s0+i1*s1+i2*s2+i3*s3]; accumulate=1 adds into an fp32 dst */
int vfm_strided_copy(const void* src, int src_dt, void* dst, int dst_dt, long n0, long n1, long n2, long n3, long s0,
                     long s1, long s2, long s3, long d0, long d1, long d2, long d3, int accumulate, void* stream);
/* A table of such copies in ONE launch (fp32 sources; entries in DEVICE memory; max_elems = the largest entry's element count):
 * the per-step re-pack of all trainable decoder weights (Conv2d / ConvTranspose2d / Linear parameters of linear_head.py:36-48,
 * VFMHead.py:28-49, Transformer.py:95-177) into their GEMM operand layouts. */
typedef struct vfm_copy_job {
  const float* src; void* dst; long dst_dt; long accumulate;   /* accumulate != 0: dst (fp32) += src */
  long n[4], s[4], d[4];
  long nsum, sum_stride;   /* nsum > 1: the value copied is the sum of nsum source slices sum_stride elements apart (the
                            * partial results of a reduction cut into nsum batch entries); 0 / 1: a plain copy */
} vfm_copy_job;
int vfm_strided_copy_batch(const vfm_copy_job* table_dev, int njobs, long max_elems, void* stream);
/* y = a*x + b*y elementwise over n fp32 values (gradient accumulation, residual adds) */
int vfm_axpby(const float* x, float a, float* y, float b, long n, void* stream);
/* y[i] *= *scalar (device scalar; keeps the loss-scale multiply on the GPU) */
int vfm_scale_by_device_scalar(float* y, const float* scalar, long n, void* stream);
/* out[c] (+)= sum_r x[r,c]  (bias gradients); deterministic two-stage reduction, ws >= 64*cols floats */
int vfm_colsum(const void* x, int dt, long ld, long rows, long cols, float* out, int accumulate, float* ws, void* stream);
/* Re-pack of the LoRA factors (peft lora.Linear, lora_backbone.py:16-23) of ALL adapter sites into their K-concatenated GEMM
 * operands, one launch: a[:r,:K] = A, at[:K,:r] = A^T, w[:N,Kw:Kw+r] = B, wt[Kw:Kw+r,:N] = B^T (wt may be null).  The
 * table is an array of nsites entries in DEVICE memory; dst dtype dt; max_elems = max over sites of r*(K+N). */
typedef struct vfm_lora_site {
  const float* A; const float* B;     /* fp32 [r,K], [N,r] */
  void* a; void* at; void* w; void* wt;
  long r, K, N, Kw, ld_a, ld_at, ld_w, ld_wt;
} vfm_lora_site;
int vfm_lora_pack(const vfm_lora_site* table_dev, int nsites, long max_elems, int dt, void* stream);
/* Split-K combine fused with the scatter into the parameter-gradient layout:
 * dst[p*sp + q*sq] (+)= alpha * sum_k slabs[k][p][q], p < rows_used (slabs fp32 [kch, P, Q]).  LoRA dA/dB (peft lora.Linear
 * backward) from the transposed-B weight-gradient GEMM of vfm_gemm. */
int vfm_slab_reduce(const float* slabs, int kch, long P, long Q, long rows_used, float alpha, float* dst, long sp, long sq,
                    int accumulate, void* stream);
/* Bernoulli keep-mask multipliers: out[i] = (u_i >= p) ? 1/(1-p) : 0, counter-based hash RNG (seed, offset) */
int vfm_dropout_mask(void* out, int dt, long n, float p, uint64_t seed, uint64_t offset, void* stream);
/* dst[r,c] = src[r,c] * mask[(r / rows_per_group) * mask_ld + c]   (Dropout2d: one multiplier per (image, channel));
 * rows_per_group == 1 gives plain elementwise dropout */
int vfm_mul_mask(const void* src, int src_dt, long ld_src, const void* mask, int mask_dt, long mask_ld,
                 long rows_per_group, void* dst, int dst_dt, long ld_dst, long rows, long cols, void* stream);
/* GEGLU (Transformer.py:52-59): out[r, c] = h[r, c] * gelu(h[r, C + c]),  h is [rows, 2C] */
int vfm_geglu_fwd(const void* h, int h_dt, long ld_h, void* out, int out_dt, long ld_out, long rows, long C, void* stream);
int vfm_geglu_bwd(const void* h, int h_dt, long ld_h, const void* dout, int do_dt, long ld_do, void* dh, int dh_dt,
                  long ld_dh, long rows, long C, void* stream);
/* SwiGLU (eva_02.py:235-242): out[r,c] = silu(h[r,c]) * h[r, C+c],  h is [rows, 2C] */
int vfm_swiglu_fwd(const void* h, int h_dt, long ld_h, void* out, int out_dt, long ld_out, long rows, long C, void* stream);
int vfm_swiglu_bwd(const void* h, int h_dt, long ld_h, const void* dout, int do_dt, long ld_do, void* dh, int dh_dt,
                   long ld_dh, long rows, long C, void* stream);
/* 2-D rotary embedding in place (eva_02.py:119-160, 362-369): y = x*cos + rotate_half(x)*sin on columns [0,ncols) = heads
 * of width d of rows whose token index is row % np; cos/sin fp32 [np, d]; inverse=1 applies the transpose (backward) */
int vfm_rope(void* x, int dt, long ld, long rows, int np, int ncols, int d, const float* cos_t, const float* sin_t,
             int inverse, void* stream);
/* out[r,c] = dy[r,c] * act'(pre[r,c])  (act: VFM_ACT_GELU on the saved pre-activation; VFM_ACT_RELU may be given the
 * post-activation, the sign test is identical) */
int vfm_act_grad_mul(const void* dy, int dy_dt, long ld_dy, const void* pre, int pre_dt, long ld_pre, void* out, int out_dt,
                     long ld_out, long rows, long cols, int act, void* stream);
/* query masking (Transformer.py:263-268): out[r,:] = keep[r] ? x[r,:] : token[:] ; bwd splits the gradient */
int vfm_mask_token_fwd(const float* x, const uint8_t* keep, const float* token, float* out, long rows, long C, void* stream);
int vfm_mask_token_bwd(const float* dout, const uint8_t* keep, float* dx, float* dtoken, float* ws /* >= 64*C floats */,
                       long rows, long C, void* stream);

/* ---- normalisation --------------------------------------------------------------------------- */
/* LayerNorm over the last dim (block.py:63,75; Transformer.py:167-169). x fp32 [rows, C]; y in out_dt; optional second
 * output y2 = y * mask2 (LoRA input dropout, peft lora.Linear) ; stats = (mean, rstd) fp32 [rows,2] */
int vfm_layernorm_fwd(const float* x, long ld_x, const float* w, const float* b, float eps, void* y, int y_dt, long ld_y,
                      float* stats, long rows, long C, void* stream);
/* bf16x3 mode: LayerNorm whose output goes (also) out as the split-bf16 A operand of the GEMM that consumes it - y3 bf16 [rows, >= 3 plane]:
 * hi = bf16(y) at column c, hi again at plane + c, lo = bf16(y - hi) at 2 plane + c (what vfm_split3 pattern 0 writes; plane >= C, a
 * multiple of 64 for the GEMM) - so that no separate vfm_split3 pass reads y back.  y (fp32) may be NULL: then only the split image is
 * written.  C in {256, 512, 1024, 1280, 2048}.  (rein/models/backbones/dino_layers/block.py:63,75 in the fp32 parity configuration.) */
int vfm_layernorm_fwd_split3(const float* x, long ld_x, const float* w, const float* b, float eps, float* y, long ld_y, void* y3, long ld3,
                             long plane, float* stats, long rows, long C, void* stream);
/* LayerNorm forward fused with the LoRA-branch dropout (peft lora.Linear: lora_dropout(x)): besides y (bf16) it
 * writes mask[row, c] = 0 or 1/(1-p) (bf16; the values vfm_dropout_mask gives for element offset + row*C + c) and
 * y_drop = y * mask.  C % 256 == 0, 16-byte aligned rows. */
int vfm_layernorm_dropout_fwd(const float* x, long ld_x, const float* w, const float* b, float eps, void* y, long ld_y,
                              float* stats, void* y_drop, long ld_yd, void* mask, long ld_mask, float p, uint64_t seed,
                              uint64_t offset, long rows, long C, void* stream);
/* LN backward that also emits t_out = bf16(dx_new * t_scale[c]) - the LayerScale-weighted operand of the next dgrad GEMM
 * (block.py:99-114: x + ls(f(norm(x)))).  Vectorised path only (C % 256 == 0), no dw/db. */
int vfm_layernorm_bwd_scaled(const void* dy, int dy_dt, long ld_dy, const float* x, long ld_x, const float* w,
                             const float* stats, float* dx, long ld_dx, int accumulate_dx, void* t_out, long ld_t,
                             const float* t_scale, long rows, long C, void* stream);
/* dx (+)= LN backward; dw/db accumulate into fp32 [C] when non-null (two-stage, ws >= 2*128*C floats) */
int vfm_layernorm_bwd(const void* dy, int dy_dt, long ld_dy, const float* x, long ld_x, const float* w,
                      const float* stats, float* dx, long ld_dx, int accumulate_dx, float* dw, float* db, float* ws,
                      long rows, long C, void* stream);
/* GroupNorm on token-major maps [B, P, C] (P pixels): groups of C/G channels, statistics over (P, C/G).
 * (linear_head.py:36-40 ConvModule GN; VFMHead.py:28-49; Transformer.py:91-92 eps 1e-6). Fused activation.
 * stats fp32 [B, G, 2]. y in y_dt; */
int vfm_groupnorm_fwd(const float* x, const float* w, const float* b, float eps, int G, int act, void* y, int y_dt,
                      float* stats, float* ws, long B, long P, long C, void* stream);
/* ws for both: >= B*G*2 + B*64*2*C floats */
int vfm_groupnorm_bwd(const void* dy, int dy_dt, const float* x, const float* w, const float* b, const float* stats,
                      int G, int act, float* dx, float* dw, float* db, float* ws, long B, long P, long C, void* stream);
/* BatchNorm (nn.SyncBatchNorm, linear_head.py:44) on [rows, C]: partial moments -> (sum, sumsq) fp32 [2,C] */
int vfm_bn_moments(const float* x, long rows, long C, float* sums, float* ws, void* stream);
/* (sum, sumsq) over `count` rows (after the DP all-reduce of the sums, if any) -> mean_var fp32 [2,C] (biased var);
 * updates running stats like nn.SyncBatchNorm (momentum, unbiased var) when non-null */
int vfm_bn_finalize(const float* sums, float count, float* mean_var, float* running_mean, float* running_var,
                    float momentum, long C, void* stream);
/* y = act((x-mean)*rstd*w+b); mean_var fp32 [2,C] (biased var) */
int vfm_bn_apply(const float* x, const float* mean_var, const float* w, const float* b, float eps, int act, void* y,
                 int y_dt, long rows, long C, void* stream);
/* sums_dy fp32 [2,C] = (sum dz, sum dz*xhat) where dz = dy*act'(.)  (the quantities SyncBN all-reduces in backward) */
int vfm_bn_bwd_reduce(const void* dy, int dy_dt, const float* x, const float* mean_var, const float* w, const float* b,
                      float eps, int act, float* sums_dy, float* ws, long rows, long C, void* stream);
/* dx from the (possibly all-reduced) sums; total_rows = global row count */
int vfm_bn_bwd_apply(const void* dy, int dy_dt, const float* x, const float* mean_var, const float* w, const float* b,
                     float eps, int act, const float* sums_dy, float total_rows, float* dx, long rows, long C,
                     void* stream);

/* ---- GEMM ------------------------------------------------------------------------------------- */
/* C[M,N] = epilogue( alpha * sum_k A[m,k] * B[n,k] )       (torch.nn.Linear / 1x1 conv / ConvT2x2 / dgrad / wgrad)
 *   in_dt VFM_BF16: MFMA bf16 path, A bf16 [M,K] with K contiguous, K % 64 == 0 (pad with zeros); B either [N,K]
 *                   (sb_k == 1) or [K,N] (sb_n == 1, N % 8 == 0: weight-gradient GEMMs consume activations in place)
 *   in_dt VFM_F32 : exact-fp32 MFMA path, arbitrary element strides (sa_m,sa_k),(sb_n,sb_k), any K
 * epilogue order:  v = alpha*acc + bias[n % bias_mod];  if C2: C2 = v (pre-activation, saved for backward)
 *                  ep_mode: GELU/RELU -> v = act(v);  MUL_GELU_GRAD -> v *= gelu'(aux[m,n]);  MUL -> v *= aux[m,n];
 *                           GELU_DGELU -> C2 = gelu'(v) instead of v, then v = gelu(v)   (mlp.py:34-40 forward saving what its
 *                           backward multiplies by)
 *                  v *= colscale[n];  v += residual[m,n];  C = v   (residual may alias C: in-place accumulate)
 * batch: z in [0,batch): operand offsets z*stride (elements).  (attention.py:51-53,58,80; mlp.py:34-40; linear_head.py:36-48)
 */
typedef struct vfm_gemm_desc {
  const void* A; const void* B; void* C;
  int in_dt; int c_dt;
  long M, N, K;
  long sa_m, sa_k, sb_n, sb_k, ldc;
  float alpha;
  const float* bias; long bias_mod;
  const float* colscale;
  const void* residual; int r_dt; long ldr;
  int ep_mode; const void* aux; int aux_dt; long ld_aux;
  void* C2; int c2_dt; long ldc2;
  long batch, stride_a, stride_b, stride_c;
  long c_plane; /* c_dt == VFM_SPLIT3: columns between the planes of the split image (>= N, a multiple of 8); ldc >= 3 c_plane counts bf16 elements */
  long kb_rows; /* bf16, B given as [K,N] (sb_n == 1): number of valid rows of B (<= K; A must be zero beyond); 0 = K.  With batch > 1 a
                 * positive value counts the rows of ONE matrix that the batches slice along K (split-K); a negative value -r gives
                 * every batch r valid rows of its own (independent problems, e.g. one weight gradient per layer) */
} vfm_gemm_desc;
int vfm_gemm(const vfm_gemm_desc* d, void* stream);
/* tuning / experiment knobs (bench.py --tune KEY=INT).  bf16 GEMM dispatch: "gemm_cfg" forces a tile configuration (-1 = heuristic;
 * 10/18 small tiles, 17 128x128 two-stage, 30/31 ping-pong, 32/33 256x256 ring, 34 128x128 five-chunk ring, 35/36 deeper rings),
 * "gemm_use_pp" bit mask of the kernels the heuristic may pick (default 40), "gemm_split_tail", "gemm_fold_tail", "gemm_bt64",
 * "gemm_batch_tiles".  Unknown keys return VFM_E_INVAL. */
int vfm_tune(const char* key, int value);

/* ---- attention -------------------------------------------------------------------------------- */
/* softmax(q k^T * scale) v per (image, head)  - xformers memory_efficient_attention (attention.py:73-89,
 * Transformer.py:140-156).  Token rows of image b: sequence index i < n_main -> row b*n_main + i,
 * else (the [cls] token, stored after all patch tokens) row B*n_main + b.  q/k/v element (row, h*d + j) at
 * ptr[row*ld + h*d + j].  lse fp32 [B, H, nq] (log-sum-exp of the scaled scores) is saved for backward. */
typedef struct vfm_attn_desc {
  const void* q; const void* k; const void* v; void* o;
  int dt; long ldq, ldk, ldv, ldo;
  int B, H, d;
  int nq_main, nq_extra, nk_main, nk_extra;
  float scale;
  float* lse;
  /* backward only */
  const void* dout; long ld_do;
  void* dq; void* dk; void* dv; long ld_dq, ld_dk, ld_dv;
  float* delta; /* backward workspace: fp32 [B,H,nq] followed by [B*H*192] floats that must be ZERO on entry (they are zero again on
                 * return): scratch for the extra-token rows, gathered with fp32 atomics */
} vfm_attn_desc;
int vfm_attn_fwd(const vfm_attn_desc* d, void* stream);
int vfm_attn_bwd(const vfm_attn_desc* d, void* stream);
/* Split-bf16 ("bf16 x 3") forward, head dim 64: q / o fp32 (dt = VFM_F32), k / v SPLIT bf16 operands as vfm_split3(pattern 1) writes them
 * for K = H*64 - the hi half of head h at columns h*64.., the lo half at lo_off + h*64.. (lo_off >= H*64).  S = q k^T and O = P v are
 * computed as hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation, softmax and output in fp32: the exact-fp32 attention of the
 * parity configuration (rein/models/backbones/dino_layers/attention.py:73-89 in fp32) to ~1e-5 at MFMA speed.  lse as vfm_attn_fwd. */
int vfm_attn_fwd_x3(const vfm_attn_desc* d, long lo_off, void* stream);
/* ... with the output (also) as the split-bf16 A operand of the projection GEMM (o3 bf16 [rows, >= 3 plane], layout as vfm_layernorm_fwd_split3);
 * d->o may be NULL: then only the split image is written. */
int vfm_attn_fwd_x3_split(const vfm_attn_desc* d, long lo_off, void* o3, long ld3, long plane, void* stream);
/* ... and its backward: q / k / v / o / dout and dq / dk / dv fp32 (dt = VFM_F32), lse from the forward, delta as vfm_attn_bwd (only its
 * first B*H*nq floats are used).  q3 / k3 / v3 (bf16 [rows, ld3]) and do3 (bf16 [rows, ld_do3]) = the vfm_split3(pattern 1) images of q / k / v / dout, hi
 * half of head h at columns h*64.., lo half at lo_off (lo_off_do3) + h*64.. (q3 / k3 / v3 may be column offsets into ONE split of a packed qkv buffer): the streamed operands of the two kernels (dQ streams K, V; dK/dV streams Q, dO); the
 * stationary rows are read in fp32 and split in registers, P and dS are split from the fp32 accumulators.  Five products, each as
 * hi*hi + hi*lo + lo*hi on the bf16 MFMA: the exact-fp32 attention backward of the parity configuration
 * (rein/models/backbones/dino_layers/attention.py:73-89 under autograd, in fp32) to ~1e-5 at MFMA speed. */
int vfm_attn_bwd_x3(const vfm_attn_desc* d, const void* q3, const void* k3, const void* v3, long ld3, long lo_off, const void* do3,
                    long ld_do3, long lo_off_do3, void* stream);

/* ---- SAM (ViTDet) windowed attention with decomposed relative-position bias (sam_vit.py:273-430) --------------
 * The bias q.Rh[qh,kh] + q.Rw[qw,kw] is folded into augmented operands so the attention is batched GEMMs + softmax:
 *   q_aug = [scale*q | q.Rh[qh,:] | q.Rw[qw,:] | 0],  k_aug = [k | onehot(kh) | onehot(kw) | 0]   (width Dq, multiple of 64)
 * vfm_sam_relpos_table : get_rel_pos - (re-interpolated) table gathered at q-k+(S-1) -> fp32 [S,S,d]
 * vfm_sam_attn_prep    : window_partition (padded tokens = projection bias) + head split + augmentation
 * vfm_softmax_rows     : fp32 scores -> probabilities (zero-padded columns)
 * vfm_sam_attn_merge   : window_unpartition + head merge back to token-major */
int vfm_sam_relpos_table(const float* rel_pos, int L, int d, int S, float* out, void* stream);
int vfm_sam_attn_prep(const void* qkv, int dt, long ld, const float* bias, const float* rh, const float* rw, void* q_aug,
                      void* k_aug, void* v_win, int nimg, int G, int S, int H, int d, int Dq, int NP,
                      int rows_per_batch /* of q_aug / k_aug, >= S*S */, float scale, void* stream);
int vfm_softmax_rows(const float* scores, long ld_s, void* out, int out_dt, long ld_o, long rows, int n, int npad, void* stream);
int vfm_sam_attn_merge(const void* o_win, int dt, void* out, long ld, int nimg, int G, int S, int H, int d, int NP, void* stream);
/* Backward of the SAM attention form (sam_vit.py:273-298, :392-430 under autograd).  bwd_prep: window-partitioned, head-dim
 * zero-padded (dp) operands dO [nb,NP,dp] and its transpose, V [nb,NP,dp], (scale q)^T [nb,dp,NP]; softmax_rows_batched /
 * softmax_rows_bwd: row softmax and its backward over batches of rows_per_batch rows of which valid_rows are real (the rest
 * are zero-filled); bwd_merge: dQaug [nb,NP,Dq], dK^T, dV^T [nb,dp,NP] -> token-major dqkv (rel-pos chain rule folded in). */
int vfm_sam_attn_bwd_prep(const void* dao, long ld_dao, const void* qkv, long ld_qkv, const float* bias, int dt, void* dow,
                          void* dowT, void* vp, void* qsT, int nimg, int G, int S, int H, int d, int dp, int NP, float scale,
                          void* stream);
int vfm_softmax_rows_batched(const float* scores, long ld_s, void* out, int out_dt, long ld_o, long rows, int n, int npad,
                             int rows_per_batch, int valid_rows, void* stream);
int vfm_softmax_rows_bwd(const void* p, const float* dp, long ld_dp, void* ds, int dt, long ld_p, long rows, int n, int npad,
                         int rows_per_batch, int valid_rows, void* stream);
int vfm_sam_attn_bwd_merge(const void* dqa, const void* dkT, const void* dvT, int dt, const float* rh, const float* rw, void* dqkv,
                           long ld, int nimg, int G, int S, int H, int d, int dp, int NP, int Dq, float scale, void* stream);

/* Flash-style forward of the same attention (inference; head dim 80 = SAM ViT-H): token-major qkv [nimg*G*G, 3*H*d] -> token-major
 * out [nimg*G*G, H*d] in ONE launch - windows of S = 14 on the zero-padded grid (padded tokens: k / v = projection bias) or global
 * attention (S = G = 32).  tbl_h / tbl_w: bf16 [2*SP, d] (SP = 16 for S = 14, 32 for S = 32) relative-index tables,
 * tbl[j] = rel_pos(re-interpolated)[j] = Rh[qh, kh] for qh - kh + S - 1 = j, rows >= 2S-1 zero.  No score matrix in memory.
 * Alignment: qkv, out, tables and bias 16 bytes; ld, ldo multiples of 8 elements (rows move as 16-byte pieces).  VFM_E_ALIGN otherwise. */
int vfm_sam_attn_flash_fwd(const void* qkv, long ld, const float* bias, const void* tbl_h, const void* tbl_w, void* out, long ldo,
                           int nimg, int G, int S, int H, int d, float scale, void* stream);
/* Training form of the same launch: also writes, per (image, window, head) and window token (row pitch NWINP = 256 for S = 14,
 * 1024 for S = 32; vfm_sam_attn_flash_stat_rows gives the row count), lse = log2-sum-exp of the scaled scores (fp32) and
 * qext = the bias columns [Bh / scale | Bw / scale] of the query operand (bf16 [rows, 2*SP]). */
int vfm_sam_attn_flash_fwd_train(const void* qkv, long ld, const float* bias, const void* tbl_h, const void* tbl_w, void* out, long ldo,
                                 int nimg, int G, int S, int H, int d, float scale, float* lse, void* qext, void* stream);
long vfm_sam_attn_flash_stat_rows(int nimg, int G, int S, int H);
/* Backward of the above with frozen tables (autograd of sam_vit.py:273-298, 392-430): out / dout [nimg*G*G, H*d] bf16 (same
 * pitch ldo), lse / qext from the training forward, dsum = fp32 scratch [rows] -> dqkv [nimg*G*G, 3*H*d] bf16, every element
 * written (gradients of padded window tokens are dropped, as window_unpartition / F.pad do).  Two launches, no score matrix. */
int vfm_sam_attn_flash_bwd(const void* qkv, long ld, const float* bias, const void* tbl_h, const void* tbl_w, const void* out,
                           const void* dout, long ldo, const float* lse, const void* qext, float* dsum, void* dqkv, long ldg,
                           int nimg, int G, int S, int H, int d, float scale, void* stream);

/* ---- ViT input / output ------------------------------------------------------------------------ */
/* im2col of non-overlapping PxP patches (patch_embed.py:65-77): img fp32 NCHW [B,3,H,W] (crop window y0,x0,
 * row/plane strides in elements) -> A [B*(h/P)*(w/P), 3*P*P] in out_dt, k = c*P*P + py*P + px (conv weight order) */
int vfm_patchify(const float* img, long stride_b, long stride_c, long stride_y, int y0, int x0, int h, int w, int P,
                 void* out, int out_dt, long ld_out, int B, void* stream);
/* tokens = [patch rows + pos[1+i]] then [cls + pos[0]] rows at the end (dino_v2.py:217-228, cls-last layout) */
int vfm_assemble_tokens(const float* patch_tok, const float* cls, const float* pos, float* x, int B, int np, int C,
                        void* stream);

/* ---- resize / loss ------------------------------------------------------------------------------ */
/* bilinear, align_corners=False (F.interpolate semantics: src=(dst+0.5)*scale-0.5 clamped at 0), fp32.
 * in: [B,Hi,Wi,C] (in_nchw=0) or [B,C,Hi,Wi] (in_nchw=1). The *virtual* output is Hv x Wv; only the window
 * (y0,x0,hc,wc) is produced.  out layout: out_mode 0 = NHWC [B,hc,wc,ldc_out] (C valid, rest zero),
 * 1 = NCHW [B,C,hc,wc], 2 = NHWC with 2-level 2x2 blocked pixel order (b, y/4, x/4, (y/2)%2, (x/2)%2, y%2, x%2, C)
 * so that stride-2 2x2 convolutions become GEMMs on views (VFMHead.py:38-45).  (Ms_VFM_encoder_decoder.py:129-133,
 * 160-167; VFMHead.py:63-67; linear_head.py:76-80) */
int vfm_resize_bilinear(const void* in, int in_dt, int in_nchw, int B, int Hi, int Wi, int C, long in_ld_c, void* out,
                        int out_dt, int out_mode, long out_ld_c, int Hv, int Wv, int y0, int x0, int hc, int wc,
                        void* stream);
/* bicubic (A=-0.75, align_corners=False, explicit source scales as F.interpolate(scale_factor=...) passes them) on a
 * token-major fp32 map [Hi,Wi,C] -> [Ho,Wo,C]: DINOv2 pos-embed re-interpolation (dino_v2.py:184-215) */
int vfm_resize_bicubic(const float* in, int Hi, int Wi, int C, float* out, int Ho, int Wo, float scale_y, float scale_x,
                       void* stream);
/* nearest (F.interpolate mode='nearest': src=floor(dst*scale)) + crop for int64 label maps (get_lr_seg/get_hr_seg,
 * Ms_VFM_encoder_decoder.py:148-158). in [B,Hi,Wi] -> out [B,hc,wc] of the virtual Hv x Wv map */
int vfm_label_resize(const int64_t* in, int B, int Hi, int Wi, int64_t* out, int Hv, int Wv, int y0, int x0, int hc,
                     int wc, void* stream);
/* blocked <-> raster pixel order for small maps: x [B, H*W (blocked, levels), C] -> y [B,H,W,C] (inverse when inv=1) */
int vfm_unblock(const float* x, float* y, int B, int H, int W, int C, int levels, int inverse, void* stream);
/* fused bilinear-upsample + softmax cross-entropy (ignore_index) + top-1 accuracy + d(logits_low)
 * (mmseg CrossEntropyLoss(avg_non_ignore=False) + accuracy via linear_head.py:72-113 / VFMHead.py:91-133):
 *   logits_low fp32 NHWC [B,h,w,C] ; label int64 [B,H,W] ; loss = sum_valid(-log p[label]) / (B*H*W)
 *   out: loss_parts fp32 [B*h*w] (summed by caller-visible vfm_reduce_sum), counts int32 [2] = (hits, valid),
 *   dlogits fp32 [B,h,w,C] = d loss / d logits_low (may be null for inference).  The full-resolution logits never
 *   touch HBM. */
int vfm_upsample_ce(const float* logits_low, const int64_t* label, int B, int h, int w, int C, int H, int W,
                    int ignore_index, float* loss_parts, int32_t* counts, float* dlogits, void* stream);
/* After vfm_upsample_ce: loss[0] = scale * sum(loss_parts[0..n)), acc[0] = 100 * counts[0] / (counts[1] + eps) (mmseg `accuracy`
 * over the valid pixels, linear_head.py:95-108), then counts is reset to (0, 0) for the next call. */
int vfm_ce_finish(const float* loss_parts, long n, float scale, int32_t* counts, float eps, float* loss, float* acc, void* stream);
/* deterministic sum of n floats -> out[0] (out[0] *= scale) */
int vfm_reduce_sum(const float* x, long n, float scale, float* out, void* stream);

/* ---- inference helpers --------------------------------------------------------------------------- */
/* mmseg SegDataPreProcessor (lora_dinov2_ms_masked.py:4-12) for one decoded uint8 CHW image (device memory): BGR->RGB,
 * (x - mean[c]) / std[c] (mean3 / std3: host arrays of 3, given in the OUTPUT channel order), right/bottom padding with pad_val. */
int vfm_preprocess_u8(const uint8_t* img, int H, int W, float* out, int Hp, int Wp, const float* mean3, const float* std3,
                      int bgr_to_rgb, float pad_val, void* stream);
/* confidence gate (Ms_VFM_encoder_decoder.py:446-448): frac[0] = mean_pixels( max softmax(logits) > thr ) over the
 * window of an NCHW map [B,C,H,W]; counts int32 [1] zeroed by the caller */
int vfm_conf_gate(const float* logits, int B, int C, int H, int W, int y0, int x0, int hc, int wc, float thr,
                  int32_t* count, void* stream);
/* preds[:, :, y0:y0+hc, x0:x0+wc] += bilinear_resize(crop_logits NHWC [B,h,w,C] or NCHW) ; count += 1
 * (slide accumulate, Ms_VFM_encoder_decoder.py:453-459) */
int vfm_slide_accumulate(const float* crop, int crop_nchw, int B, int h, int w, int C, float* preds, float* count,
                         int H, int W, int y0, int x0, int hc, int wc, void* stream);
/* The same merge as ONE gather pass: preds[b,c,y,x] = (sum over the windows j that hold (y,x), in table order, of the bilinear sample of
 * window j's logits) / (their number).  preds is only written (no zero fill, no count map, no finalize pass); same per-sample
 * arithmetic and summation order as vfm_slide_accumulate + vfm_slide_finalize.  wins: HOST array of nwin <= 16 descriptors; C <= 32. */
typedef struct vfm_slide_win {
  const float* crop; /* window logits: NHWC [B,h,w,C] (nchw = 0) or NCHW [B,C,h,w] (nchw = 1), device pointer */
  int nchw, h, w;    /* layout and resolution of the window logits */
  int y0, x0, hc, wc;/* where the window sits in the [H,W] map and its size there */
} vfm_slide_win;
int vfm_slide_gather(const vfm_slide_win* wins, int nwin, int B, int C, float* preds, int H, int W, void* stream);
/* vfm_conf_gate for nwin <= 16 windows at once: boxes = HOST int[nwin*4] {y0, x0, hc, wc}; counts int32 [nwin] zeroed by the caller.  The
 * predicate is evaluated once per pixel and counted for every window that holds it (Ms_VFM_encoder_decoder.py:446-448 per window). */
int vfm_conf_gate_windows(const float* logits, int B, int C, int H, int W, const int* boxes, int nwin, float thr, int32_t* counts,
                          void* stream);
/* seg = preds / count ; pred = argmax_c  (NCHW) */
int vfm_slide_finalize(float* preds, const float* count, uint8_t* argmax, int B, int C, int H, int W, void* stream);

/* ---- evaluation ------------------------------------------------------------------------------------ */
/* Confusion histogram behind mmseg IoUMetric.intersect_and_union, which rein/dg_metrics.py:46-52 (DGIoUMetric.process) calls per
 * sample: for every pixel with label != ignore_index, hist[row * num_classes + pred] += 1 where row = label if 0 <= label <
 * num_classes else num_classes (labels outside the class range still count into the prediction areas, as torch.histc on
 * pred[mask] does).  hist: int64 [(num_classes+1) * num_classes], ACCUMULATED into (zero it once per evaluation);
 * area_intersect = diag, area_pred_label = column sums, area_label = row sums of the first num_classes rows.
 * pred uint8 [n], label int64 or uint8 [n], both 16-byte aligned; num_classes <= 64. */
int vfm_confusion_hist(const uint8_t* pred, const void* label, int label_dt, long n, int num_classes, int ignore_index,
                       int64_t* hist, void* stream);

/* ---- optimiser ------------------------------------------------------------------------------------- */
/* fused multi-tensor AdamW over one flat fp32 buffer (torch.optim.AdamW semantics; groups from
 * peft_optimizer_constructor.py:25-147): segment s covers [seg_start[s], seg_start[s+1]) with lr*seg_lr_mult[s],
 * seg_wd[s]. lr and step are passed by value each iteration (PolyLR on the host).  zero_grad != 0 also clears g in the same
 * pass (OptimWrapper.update_params: step then zero_grad); vec4 != 0 promises that every seg_start is a multiple of 4 (and the
 * buffers 16-byte aligned): one float4 per lane. */
int vfm_adamw(float* p, float* g, float* m, float* v, long n, const long* seg_start, const float* seg_lr_mult,
              const float* seg_wd, int n_seg, float lr, float beta1, float beta2, float eps, int step, float grad_scale,
              int zero_grad, int vec4, void* stream);
/* ... for loss-scaled training (mmengine AmpOptimWrapper / torch GradScaler, tools/train.py:87-102) with the scaler's state on the DEVICE, so
 * that the host never waits for a backward pass:  *skip != 0 (the step's gradients hold an inf / NaN) leaves p / m / v untouched (g is still
 * cleared when zero_grad != 0);  amp_state = float[4] {loss scale, growth tracker, optimiser steps taken, steps skipped}: when given, the
 * gradients are multiplied by grad_scale / amp_state[0] and the bias corrections use step amp_state[2] + 1 (`step` is ignored).
 * Both may be NULL (= vfm_adamw).  vfm_amp_update is GradScaler.update() on that state: after a skipped step the scale is multiplied by
 * `backoff` and the tracker cleared; after a good one the step count and the tracker advance and every `interval` good steps the scale is
 * multiplied by `growth` (dynamic == 0: the scale stays, the counters still move). */
int vfm_adamw_guarded(float* p, float* g, float* m, float* v, long n, const long* seg_start, const float* seg_lr_mult,
                      const float* seg_wd, int n_seg, float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                      int zero_grad, int vec4, const int* skip, const float* amp_state, void* stream);
int vfm_amp_update(const int* flag, float* amp_state, float growth, float backoff, int interval, int dynamic, void* stream);

/* ---- launch plans -------------------------------------------------------------------------------------- */
/* A recorded sequence of the calls above, replayed in order on one stream by ONE call.  The hot path's backbone is a fixed launch
 * sequence per step (18 launches per transformer block: rein/models/backbones/dino_v2.py:259-291 forward_features and its backward) over
 * buffers whose addresses do not change between steps; issuing it from C costs ~3 us of host time per launch where the Python binding
 * pays ~14 us - with eight ranks of a node each enqueuing ~800 launches per step, host time is what a data-parallel step waits for
 * (configs/_base_/default_runtime.py:5, tools/dist_train.sh:9-17).  This is a host-side replay, not a graph capture: every launch goes
 * through the same entry point with the same arguments as when it is issued one by one, so results are bit-identical.
 * Per-replay scalars: `seed` and `rng_base` feed the dropout of VFM_OP_LN_DROPOUT_FWD entries (their `offset` is relative to rng_base).
 * prof_kind / flops: the class and algorithmic FLOPs the profiling sampler (below) files the launch under (0 = not sampled). */
enum { VFM_OP_GEMM = 0, VFM_OP_LN_FWD = 1, VFM_OP_LN_DROPOUT_FWD = 2, VFM_OP_LN_BWD_SCALED = 3, VFM_OP_ATTN_FWD = 4, VFM_OP_ATTN_BWD = 5,
       VFM_OP_CAST = 6, VFM_OP_STRIDED_COPY = 7 };
enum { VFM_PROF_NONE = 0, VFM_PROF_GEMM = 1, VFM_PROF_GEMM_TN = 2, VFM_PROF_ATTN_FWD = 3, VFM_PROF_ATTN_BWD = 4 };
typedef struct vfm_plan_op {
  int kind;
  int prof_kind;
  double flops;
  union {
    vfm_gemm_desc gemm;                                             /* vfm_gemm */
    vfm_attn_desc attn;                                             /* vfm_attn_fwd / vfm_attn_bwd */
    struct { const float* x; long ld_x; const float* w; const float* b; float eps; void* y; int y_dt; long ld_y; float* stats;
             long rows, C; } ln_fwd;                                /* vfm_layernorm_fwd */
    struct { const float* x; long ld_x; const float* w; const float* b; float eps; void* y; long ld_y; float* stats; void* y_drop;
             long ld_yd; void* mask; long ld_mask; float p; uint64_t offset; long rows, C; } ln_drop;   /* vfm_layernorm_dropout_fwd */
    struct { const void* dy; int dy_dt; long ld_dy; const float* x; long ld_x; const float* w; const float* stats; float* dx;
             long ld_dx; int accumulate_dx; void* t_out; long ld_t; const float* t_scale; long rows, C; } ln_bwd;   /* vfm_layernorm_bwd_scaled */
    struct { const void* src; int src_dt; long ld_src; void* dst; int dst_dt; long ld_dst; long rows, cols; const float* colscale; } cast;
    struct { const void* src; int src_dt; void* dst; int dst_dt; long n[4]; long ss[4]; long ds[4]; int accumulate; } copy;   /* vfm_strided_copy */
  } u;
} vfm_plan_op;
/* Runs ops[0..n) in order; stops at the first failing entry and returns its code (vfm_last_error names the index). */
int vfm_run_plan(const vfm_plan_op* ops, int n, uint64_t seed, uint64_t rng_base, void* stream);
/* Profiling sampler for launches issued by vfm_run_plan (the measurement hook bench.py uses for the `roofline` block: HIP events on the
 * stream the kernels are launched on).  every > 0: each every-th entry with prof_kind != 0 is bracketed by an event pair; 0 = off.
 * vfm_prof_read waits for the recorded events and writes up to `cap` records {prof_kind, flops, milliseconds} as three doubles each;
 * returns the number of records (negative on error) and clears them. */
int vfm_prof_config(int every);
int vfm_prof_read(double* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
