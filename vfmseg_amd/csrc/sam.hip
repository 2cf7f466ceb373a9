// SAM (ViTDet-style) attention helpers: decomposed relative-position bias folded into augmented Q/K so that the
// biased, windowed attention becomes plain batched GEMMs + a row softmax (reference: rein/models/backbones/sam_vit.py
// :273-298 Attention.forward, :301-356 window (un)partition, :359-430 get_rel_pos / add_decomposed_rel_pos).
//   score[q,k] = scale*q.k + q.Rh[qh,kh] + q.Rw[qw,kw]  =  [scale*q | q.Rh[qh,:] | q.Rw[qw,:]] . [k | e_kh | e_kw]
#include "common.h"

// out[S,S,d] = table[(i - j) + (S - 1)] of the (linearly re-interpolated, F.interpolate(mode='linear')) rel_pos [L,d]
__global__ void k_sam_relpos(const float* __restrict__ rel, int L, int d, int S, float* __restrict__ out) {
  const int maxrel = 2 * S - 1;
  const long total = (long)S * S * d;
  const float scale = (float)L / (float)maxrel;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % d);
    const long t = idx / d;
    const int j = (int)(t % S), i = (int)(t / S);
    const int r = i - j + (S - 1);
    float v;
    if (L == maxrel) {
      v = rel[(long)r * d + c];
    } else {
      float src = scale * (r + 0.5f) - 0.5f;
      if (src < 0.f) src = 0.f;
      int i0 = (int)src;
      if (i0 > L - 1) i0 = L - 1;
      const int i1 = i0 + (i0 < L - 1 ? 1 : 0);
      const float l1 = src - i0;
      v = (1.f - l1) * rel[(long)i0 * d + c] + l1 * rel[(long)i1 * d + c];
    }
    out[idx] = v;
  }
}
extern "C" int vfm_sam_relpos_table(const float* rel_pos, int L, int d, int S, float* out, void* stream) {
  const long total = (long)S * S * d;
  if (total == 0) return VFM_OK;
  hipLaunchKernelGGL(k_sam_relpos, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rel_pos, L, d, S, out);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// qkv token-major [nimg*G*G, 3C] -> per (image, window, head) batches of S*S tokens:
//   q_aug [nb, S*S, Dq], k_aug [nb, S*S, Dq], v_win [nb, NP, d]  (NP >= S*S rows, extra rows stay zero)
// window tokens outside the G x G grid are the zero-padded tokens of window_partition: their qkv is the projection bias.
struct SamPrepP {
  const void* qkv; int dt; long ld;
  const float* bias;          // [3C] projection bias (value of padded tokens), may be null (= 0)
  const float* rh; const float* rw;  // [S,S,d]
  void* qa; void* ka; void* vw;
  int nimg, G, S, nwin_side, H, d, C, Dq, NP;
  float scale;
};
__global__ void k_sam_prep(SamPrepP p) {
  const int S2 = p.S * p.S;
  const int nwin = p.nwin_side * p.nwin_side;
  const long nb = (long)p.nimg * nwin * p.H;
  const long total = nb * S2 * p.Dq;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % p.Dq);
    long t = idx / p.Dq;
    const int tok = (int)(t % S2);
    const long bz = t / S2;
    const int h = (int)(bz % p.H);
    const long iw = bz / p.H;
    const int win = (int)(iw % nwin);
    const int img = (int)(iw / nwin);
    const int iy = tok / p.S, ix = tok % p.S;
    const int gy = (win / p.nwin_side) * p.S + iy, gx = (win % p.nwin_side) * p.S + ix;
    const bool inside = gy < p.G && gx < p.G;
    const long row = ((long)img * p.G + gy) * p.G + gx;
    auto qv = [&](int c) -> float {
      return inside ? ld_any(p.qkv, row * p.ld + h * p.d + c, p.dt) : (p.bias ? p.bias[h * p.d + c] : 0.f);
    };
    float qo, ko;
    if (j < p.d) {
      qo = qv(j) * p.scale;
      ko = inside ? ld_any(p.qkv, row * p.ld + p.C + h * p.d + j, p.dt) : (p.bias ? p.bias[p.C + h * p.d + j] : 0.f);
      const float vv = inside ? ld_any(p.qkv, row * p.ld + 2 * p.C + h * p.d + j, p.dt) : (p.bias ? p.bias[2 * p.C + h * p.d + j] : 0.f);
      st_any(p.vw, (bz * p.NP + tok) * p.d + j, p.dt, vv);
    } else if (j < p.d + p.S) {
      const int kh = j - p.d;
      const float* r = p.rh + ((long)iy * p.S + kh) * p.d;
      float a = 0.f;
      for (int c = 0; c < p.d; ++c) a += qv(c) * r[c];
      qo = a;
      ko = (kh == iy) ? 1.f : 0.f;
    } else if (j < p.d + 2 * p.S) {
      const int kw = j - p.d - p.S;
      const float* r = p.rw + ((long)ix * p.S + kw) * p.d;
      float a = 0.f;
      for (int c = 0; c < p.d; ++c) a += qv(c) * r[c];
      qo = a;
      ko = (kw == ix) ? 1.f : 0.f;
    } else {
      qo = 0.f;
      ko = 0.f;
    }
    st_any(p.qa, idx, p.dt, qo);
    st_any(p.ka, idx, p.dt, ko);
  }
}
extern "C" int vfm_sam_attn_prep(const void* qkv, int dt, long ld, const float* bias, const float* rh, const float* rw, void* q_aug,
                                 void* k_aug, void* v_win, int nimg, int G, int S, int H, int d, int Dq, int NP, float scale,
                                 void* stream) {
  VFM_CHECK(S > 0 && G > 0 && Dq >= d + 2 * S && NP >= S * S, VFM_E_SHAPE, "vfm_sam_attn_prep: shape");
  SamPrepP p;
  p.qkv = qkv; p.dt = dt; p.ld = ld; p.bias = bias; p.rh = rh; p.rw = rw; p.qa = q_aug; p.ka = k_aug; p.vw = v_win;
  p.nimg = nimg; p.G = G; p.S = S; p.nwin_side = (G + S - 1) / S; p.H = H; p.d = d; p.C = H * d; p.Dq = Dq; p.NP = NP; p.scale = scale;
  const long total = (long)nimg * p.nwin_side * p.nwin_side * H * S * S * Dq;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipLaunchKernelGGL(k_sam_prep, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// row softmax: fp32 scores [rows, n] (ld_s) -> probabilities in out_dt [rows, npad] (ld_o), columns n..npad-1 = 0
__global__ void k_softmax_rows(const float* __restrict__ s, long ld_s, void* __restrict__ out, int out_dt, long ld_o, long rows, int n,
                               int npad) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sr = s + row * ld_s;
  float m = -INFINITY;
  for (int c = lane; c < n; c += 64) m = fmaxf(m, sr[c]);
  m = wave_max(m);
  float z = 0.f;
  for (int c = lane; c < n; c += 64) z += __expf(sr[c] - m);
  z = wave_sum(z);
  const float inv = 1.f / z;
  for (int c = lane; c < npad; c += 64) st_any(out, row * ld_o + c, out_dt, c < n ? __expf(sr[c] - m) * inv : 0.f);
}
extern "C" int vfm_softmax_rows(const float* scores, long ld_s, void* out, int out_dt, long ld_o, long rows, int n, int npad,
                                void* stream) {
  VFM_CHECK(npad >= n && ld_o >= npad && ld_s >= n, VFM_E_SHAPE, "vfm_softmax_rows: shape");
  if (rows == 0) return VFM_OK;
  hipLaunchKernelGGL(k_softmax_rows, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, scores, ld_s, out, out_dt, ld_o, rows, n, npad);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// o_win [nb, NP, d] -> token-major out [nimg*G*G, ld] columns h*d.. (window_unpartition + head merge, padded tokens dropped)
__global__ void k_sam_merge(const void* __restrict__ ow, int dt, void* __restrict__ out, long ld, int nimg, int G, int S, int nws,
                            int H, int d, int NP) {
  const long total = (long)nimg * G * G * H * d;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % d);
    long t = idx / d;
    const int h = (int)(t % H);
    t /= H;
    const int gx = (int)(t % G);
    t /= G;
    const int gy = (int)(t % G);
    const int img = (int)(t / G);
    const int win = (gy / S) * nws + gx / S, tok = (gy % S) * S + gx % S;
    const long bz = ((long)img * nws * nws + win) * H + h;
    const long row = ((long)img * G + gy) * G + gx;
    st_any(out, row * ld + h * d + c, dt, ld_any(ow, (bz * NP + tok) * d + c, dt));
  }
}
extern "C" int vfm_sam_attn_merge(const void* o_win, int dt, void* out, long ld, int nimg, int G, int S, int H, int d, int NP,
                                  void* stream) {
  const long total = (long)nimg * G * G * H * d;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_sam_merge, dim3(grid), dim3(256), 0, (hipStream_t)stream, o_win, dt, out, ld, nimg, G, S, (G + S - 1) / S, H, d, NP);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
