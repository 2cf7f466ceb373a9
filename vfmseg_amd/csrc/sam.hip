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
  int rpb;  // rows per batch of q_aug / k_aug (>= S*S; rows S*S..rpb-1 are left untouched: the caller zero-fills them)
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
    const long oidx = (bz * p.rpb + tok) * p.Dq + j;
    st_any(p.qa, oidx, p.dt, qo);
    st_any(p.ka, oidx, p.dt, ko);
  }
}
// Same outputs, one workgroup per (batch bz, window row iy) [blockIdx.y = 0] or (bz, window column ix) [blockIdx.y = 1]:
//   pass 0 copies q (scaled) / k / v of the S tokens of that row, writes k_aug's one-hot columns and the zero padding, and
//          computes the row term  q . Rh[iy, kh, :]  for the S x S (ix, kh) pairs from LDS copies of the S query vectors and of
//          the [S, d] slice Rh[iy] that all of them share;
//   pass 1 does the same for the column term  q . Rw[ix, kw, :]  with the S queries of column ix and the slice Rw[ix].
// The element-per-thread form above re-reads the query vector and a table row from global memory for every output element
// (d scalar loads each): 830 us per call on SAM-H's global blocks; this form reads every input once, coalesced.
#define SAM_PREP_MAXS 32
#define SAM_PREP_MAXD 128
__global__ void __launch_bounds__(256) k_sam_prep_rows(SamPrepP p) {
  __shared__ float qs[SAM_PREP_MAXS][SAM_PREP_MAXD + 1];
  __shared__ float rs[SAM_PREP_MAXS][SAM_PREP_MAXD + 1];
  const int S = p.S, d = p.d, S2 = S * S;
  const int nwin = p.nwin_side * p.nwin_side;
  const int pass = blockIdx.y;
  const long blk = blockIdx.x;          // bz * S + line
  const int line = (int)(blk % S);      // iy (pass 0) or ix (pass 1)
  const long bz = blk / S;
  const int h = (int)(bz % p.H);
  const long iw = bz / p.H;
  const int win = (int)(iw % nwin), img = (int)(iw / nwin);
  const int tid = threadIdx.x;
  auto token = [&](int i, int& tok, bool& inside, long& row) {  // i-th query of this line
    const int iy = pass == 0 ? line : i, ix = pass == 0 ? i : line;
    tok = iy * S + ix;
    const int gy = (win / p.nwin_side) * S + iy, gx = (win % p.nwin_side) * S + ix;
    inside = gy < p.G && gx < p.G;
    row = ((long)img * p.G + gy) * p.G + gx;
  };
  // ---- the S query vectors of the line -> LDS (and, in pass 0, the q / k / v copies)
  for (int e = tid; e < S * d; e += 256) {
    const int i = e / d, c = e % d;
    int tok; bool inside; long row;
    token(i, tok, inside, row);
    const float qv = inside ? ld_any(p.qkv, row * p.ld + h * d + c, p.dt) : (p.bias ? p.bias[h * d + c] : 0.f);
    qs[i][c] = qv;
    if (pass == 0) {
      const float kv = inside ? ld_any(p.qkv, row * p.ld + p.C + h * d + c, p.dt) : (p.bias ? p.bias[p.C + h * d + c] : 0.f);
      const float vv = inside ? ld_any(p.qkv, row * p.ld + 2 * p.C + h * d + c, p.dt) : (p.bias ? p.bias[2 * p.C + h * d + c] : 0.f);
      const long o = (bz * p.rpb + tok) * p.Dq + c;
      st_any(p.qa, o, p.dt, qv * p.scale);
      st_any(p.ka, o, p.dt, kv);
      st_any(p.vw, (bz * p.NP + tok) * d + c, p.dt, vv);
    }
  }
  // ---- the table slice shared by the line: Rh[iy] or Rw[ix], [S, d]
  const float* tab = (pass == 0 ? p.rh : p.rw) + (long)line * S * d;
  for (int e = tid; e < S * d; e += 256) rs[e / d][e % d] = tab[e];
  if (pass == 0) {  // k_aug one-hot columns and the zero padding of both augmented operands
    const int extra = p.Dq - d;
    for (int e = tid; e < S * extra; e += 256) {
      const int i = e / extra, j = d + e % extra;
      const int tok = line * S + i;  // (iy = line, ix = i)
      const long o = (bz * p.rpb + tok) * p.Dq + j;
      float ko = 0.f;
      if (j < d + S) ko = (j - d == line) ? 1.f : 0.f;
      else if (j < d + 2 * S) ko = (j - d - S == i) ? 1.f : 0.f;
      st_any(p.ka, o, p.dt, ko);
      if (j >= d + 2 * S) st_any(p.qa, o, p.dt, 0.f);
    }
  }
  __syncthreads();
  // ---- S x S dot products of length d from LDS: lane -> table row (stride d + 1 floats: conflict-free), query broadcast
  for (int e = tid; e < S2; e += 256) {
    const int i = e / S, kk = e % S;
    float a = 0.f;
#pragma unroll 8
    for (int c = 0; c < d; ++c) a = fmaf(qs[i][c], rs[kk][c], a);
    const int tok = pass == 0 ? line * S + i : i * S + line;
    st_any(p.qa, (bz * p.rpb + tok) * p.Dq + d + pass * S + kk, p.dt, a);
  }
}

// bf16 form of k_sam_prep_rows with two channels per lane: 4-byte loads and stores instead of 2-byte ones (the kernel is bound
// by the number of memory instructions, not by bytes).  d, S, Dq, ld even; all bases 4-byte aligned.
__global__ void __launch_bounds__(256) k_sam_prep_rows_bf16x2(SamPrepP p) {
  __shared__ float qs[SAM_PREP_MAXS][SAM_PREP_MAXD + 1];
  __shared__ float rs[SAM_PREP_MAXS][SAM_PREP_MAXD + 1];
  const int S = p.S, d = p.d, d2 = d >> 1, S2h = S * (S >> 1);
  const int nwin = p.nwin_side * p.nwin_side;
  const int pass = blockIdx.y;
  const long blk = blockIdx.x;
  const int line = (int)(blk % S);
  const long bz = blk / S;
  const int h = (int)(bz % p.H);
  const long iw = bz / p.H;
  const int win = (int)(iw % nwin), img = (int)(iw / nwin);
  const int tid = threadIdx.x;
  const int wy = (win / p.nwin_side) * S, wx = (win % p.nwin_side) * S;
  const bf16_t* qkv = (const bf16_t*)p.qkv;
  bf16_t* qa = (bf16_t*)p.qa;
  bf16_t* ka = (bf16_t*)p.ka;
  bf16_t* vw = (bf16_t*)p.vw;
  auto ld2 = [&](long idx) -> float2 {
    const ushort2 u = *reinterpret_cast<const ushort2*>(qkv + idx);
    return make_float2(bf16_to_f32(u.x), bf16_to_f32(u.y));
  };
  auto st2 = [&](bf16_t* base, long idx, float a, float b) {
    const ushort2 u = {f32_to_bf16(a), f32_to_bf16(b)};
    *reinterpret_cast<ushort2*>(base + idx) = u;
  };
  for (int e = tid; e < S * d2; e += 256) {
    const int i = e / d2, c = (e - i * d2) * 2;
    const int iy = pass == 0 ? line : i, ix = pass == 0 ? i : line;
    const int tok = iy * S + ix;
    const int gy = wy + iy, gx = wx + ix;
    const bool inside = gy < p.G && gx < p.G;
    const long base = (((long)img * p.G + gy) * p.G + gx) * p.ld + h * d + c;
    const float2 qv = inside ? ld2(base) : (p.bias ? make_float2(p.bias[h * d + c], p.bias[h * d + c + 1]) : make_float2(0.f, 0.f));
    qs[i][c] = qv.x;
    qs[i][c + 1] = qv.y;
    if (pass == 0) {
      const float2 kv = inside ? ld2(base + p.C) : (p.bias ? make_float2(p.bias[p.C + h * d + c], p.bias[p.C + h * d + c + 1]) : make_float2(0.f, 0.f));
      const float2 vv = inside ? ld2(base + 2 * p.C)
                               : (p.bias ? make_float2(p.bias[2 * p.C + h * d + c], p.bias[2 * p.C + h * d + c + 1]) : make_float2(0.f, 0.f));
      const long o = (bz * p.rpb + tok) * p.Dq + c;
      st2(qa, o, qv.x * p.scale, qv.y * p.scale);
      st2(ka, o, kv.x, kv.y);
      st2(vw, (bz * p.NP + tok) * d + c, vv.x, vv.y);
    }
  }
  const float* tab = (pass == 0 ? p.rh : p.rw) + (long)line * S * d;
  for (int e = tid; e < S * d2; e += 256) {
    const int k = e / d2, c = (e - k * d2) * 2;
    const float2 t = *reinterpret_cast<const float2*>(tab + k * d + c);
    rs[k][c] = t.x;
    rs[k][c + 1] = t.y;
  }
  if (pass == 0) {
    const int extra2 = (p.Dq - d) >> 1;
    for (int e = tid; e < S * extra2; e += 256) {
      const int i = e / extra2, j = d + (e - i * extra2) * 2;
      const long o = (bz * p.rpb + line * S + i) * p.Dq + j;
      float k0 = 0.f, k1 = 0.f;
      if (j < d + S) k0 = (j - d == line) ? 1.f : 0.f, k1 = (j + 1 - d == line) ? 1.f : 0.f;
      else if (j < d + 2 * S) k0 = (j - d - S == i) ? 1.f : 0.f, k1 = (j + 1 - d - S == i) ? 1.f : 0.f;
      st2(ka, o, k0, k1);
      if (j >= d + 2 * S) st2(qa, o, 0.f, 0.f);
    }
  }
  __syncthreads();
  for (int e = tid; e < S2h; e += 256) {
    const int i = e / (S >> 1), kk = (e - i * (S >> 1)) * 2;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
    for (int c = 0; c < d; ++c) {
      const float qv = qs[i][c];
      a0 = fmaf(qv, rs[kk][c], a0);
      a1 = fmaf(qv, rs[kk + 1][c], a1);
    }
    const int tok = pass == 0 ? line * S + i : i * S + line;
    st2(qa, (bz * p.rpb + tok) * p.Dq + d + pass * S + kk, a0, a1);
  }
}

extern "C" int vfm_sam_attn_prep(const void* qkv, int dt, long ld, const float* bias, const float* rh, const float* rw, void* q_aug,
                                 void* k_aug, void* v_win, int nimg, int G, int S, int H, int d, int Dq, int NP, int rows_per_batch,
                                 float scale, void* stream) {
  VFM_CHECK(S > 0 && G > 0 && Dq >= d + 2 * S && NP >= S * S && rows_per_batch >= S * S, VFM_E_SHAPE, "vfm_sam_attn_prep: shape");
  SamPrepP p;
  p.qkv = qkv; p.dt = dt; p.ld = ld; p.bias = bias; p.rh = rh; p.rw = rw; p.qa = q_aug; p.ka = k_aug; p.vw = v_win;
  p.nimg = nimg; p.G = G; p.S = S; p.nwin_side = (G + S - 1) / S; p.H = H; p.d = d; p.C = H * d; p.Dq = Dq; p.NP = NP; p.rpb = rows_per_batch; p.scale = scale;
  const long total = (long)nimg * p.nwin_side * p.nwin_side * H * S * S * Dq;
  if (total == 0) return VFM_OK;
  if (S <= SAM_PREP_MAXS && d <= SAM_PREP_MAXD) {
    const long nb = (long)nimg * p.nwin_side * p.nwin_side * H;
    const bool x2 = dt == VFM_BF16 && d % 2 == 0 && S % 2 == 0 && Dq % 2 == 0 && ld % 2 == 0 && (H * d) % 2 == 0 && (uintptr_t)qkv % 4 == 0 &&
                    (uintptr_t)q_aug % 4 == 0 && (uintptr_t)k_aug % 4 == 0 && (uintptr_t)v_win % 4 == 0 && (uintptr_t)rh % 8 == 0 && (uintptr_t)rw % 8 == 0;
    if (x2) hipLaunchKernelGGL(k_sam_prep_rows_bf16x2, dim3((unsigned)(nb * S), 2), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(k_sam_prep_rows, dim3((unsigned)(nb * S), 2), dim3(256), 0, (hipStream_t)stream, p);
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
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
  if (n <= 1024) {  // the row lives in registers: one read, one exponential per element
    float v[16];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = i * 64 + lane;
      v[i] = c < n ? sr[c] : -INFINITY;
      m = fmaxf(m, v[i]);
    }
    m = wave_max(m);
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      v[i] = __expf(v[i] - m);  // exp(-inf) = 0 for the columns past n
      z += v[i];
    }
    const float inv = 1.f / wave_sum(z);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = i * 64 + lane;
      if (c < npad) st_any(out, row * ld_o + c, out_dt, v[i] * inv);
    }
    return;
  }
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
__global__ void k_sam_merge_bf16x2(const bf16_t* __restrict__ ow, bf16_t* __restrict__ out, long ld, int nimg, int G, int S, int nws,
                                   int H, int d, int NP) {
  const int d2 = d >> 1;
  const long total = (long)nimg * G * G * H * d2;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % d2) * 2;
    long t = idx / d2;
    const int h = (int)(t % H);
    t /= H;
    const int gx = (int)(t % G);
    t /= G;
    const int gy = (int)(t % G);
    const int img = (int)(t / G);
    const int win = (gy / S) * nws + gx / S, tok = (gy % S) * S + gx % S;
    const long bz = ((long)img * nws * nws + win) * H + h;
    const long row = ((long)img * G + gy) * G + gx;
    *reinterpret_cast<ushort2*>(out + row * ld + h * d + c) = *reinterpret_cast<const ushort2*>(ow + (bz * NP + tok) * d + c);
  }
}
extern "C" int vfm_sam_attn_merge(const void* o_win, int dt, void* out, long ld, int nimg, int G, int S, int H, int d, int NP,
                                  void* stream) {
  const long total = (long)nimg * G * G * H * d;
  if (total == 0) return VFM_OK;
  if (dt == VFM_BF16 && d % 2 == 0 && ld % 2 == 0 && (uintptr_t)o_win % 4 == 0 && (uintptr_t)out % 4 == 0) {
    const long t2 = total / 2;
    const int grid2 = (int)((t2 + 255) / 256 > 16384 ? 16384 : (t2 + 255) / 256);
    hipLaunchKernelGGL(k_sam_merge_bf16x2, dim3(grid2), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)o_win, (bf16_t*)out, ld, nimg, G, S,
                       (G + S - 1) / S, H, d, NP);
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_sam_merge, dim3(grid), dim3(256), 0, (hipStream_t)stream, o_win, dt, out, ld, nimg, G, S, (G + S - 1) / S, H, d, NP);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// ------------------------------------------------------------------------------------------------------------ backward
// Backward of the same attention form (training with a SAM backbone: lora_sam_ms_masked.py).  With P = softmax(S),
//   dP = dO V^T,  dS = P o (dP - rowsum(P o dP)),  dV^T = dO^T P,  dK^T = (scale q)^T dS,  dQaug = dS Kaug,
//   dq = scale dQaug[:d] + sum_kh dQaug[d+kh] Rh[qh,kh,:] + sum_kw dQaug[d+S+kw] Rw[qw,kw,:]
// every product is a batched MFMA GEMM whose [K,N] operand (P, dS, Kaug) is consumed in place; only the small per-window
// operands are laid out transposed here.  The head dimension (80) is zero-padded to dp (128) where it is a GEMM K.
struct SamBwdPrepP {
  const void* dao; long ld_dao;   // token-major gradient of the attention output [nimg*G*G, >= C]
  const void* qkv; long ld_qkv;   // token-major qkv [nimg*G*G, 3C]
  const float* bias;
  int dt;
  void* dow; void* dowT; void* vp; void* qsT;   // [nb,NP,dp], [nb,dp,NP], [nb,NP,dp], [nb,dp,NP]
  int nimg, G, S, nws, H, d, C, dp, NP;
  float scale;
};
__global__ void k_sam_bwd_prep(SamBwdPrepP p) {
  const int S2 = p.S * p.S, nwin = p.nws * p.nws;
  const long nb = (long)p.nimg * nwin * p.H;
  const long total = nb * p.NP * p.dp;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % p.dp);
    long t = idx / p.dp;
    const int tok = (int)(t % p.NP);
    const long bz = t / p.NP;
    const int h = (int)(bz % p.H);
    const long iw = bz / p.H;
    const int win = (int)(iw % nwin), img = (int)(iw / nwin);
    const int iy = tok / p.S, ix = tok % p.S;
    const int gy = (win / p.nws) * p.S + iy, gx = (win % p.nws) * p.S + ix;
    const bool tokv = tok < S2, inside = tokv && gy < p.G && gx < p.G, cv = c < p.d;
    const long row = ((long)img * p.G + gy) * p.G + gx;
    float g = 0.f, vv = 0.f, qs = 0.f;
    if (cv && tokv) {
      if (inside) {
        g = ld_any(p.dao, row * p.ld_dao + h * p.d + c, p.dt);
        vv = ld_any(p.qkv, row * p.ld_qkv + 2 * p.C + h * p.d + c, p.dt);
        qs = ld_any(p.qkv, row * p.ld_qkv + h * p.d + c, p.dt) * p.scale;
      } else {  // zero-padded window token: q, k, v = projection bias, no gradient flows back
        vv = p.bias ? p.bias[2 * p.C + h * p.d + c] : 0.f;
        qs = (p.bias ? p.bias[h * p.d + c] : 0.f) * p.scale;
      }
    }
    st_any(p.dow, idx, p.dt, g);
    st_any(p.vp, idx, p.dt, vv);
    const long tidx = (bz * p.dp + c) * p.NP + tok;
    st_any(p.dowT, tidx, p.dt, g);
    st_any(p.qsT, tidx, p.dt, qs);
  }
}
// Same outputs, one workgroup per (batch bz, 32 tokens): the row-major copies are written as the values arrive (c fastest),
// the two transposed ones go through an LDS tile so that they are written token-fastest (the per-element form writes them
// with a stride of NP elements between neighbouring lanes: 318 us per call).
#define SAM_BP_MAXDP 128
__global__ void __launch_bounds__(256) k_sam_bwd_prep_tiles(SamBwdPrepP p) {
  __shared__ float tg[SAM_BP_MAXDP][33], tq[SAM_BP_MAXDP][33];
  const int S2 = p.S * p.S, nwin = p.nws * p.nws;
  const int tiles = p.NP / 32;
  const long bz = blockIdx.x / tiles;
  const int tok0 = (int)(blockIdx.x % tiles) * 32;
  const int h = (int)(bz % p.H);
  const long iw = bz / p.H;
  const int win = (int)(iw % nwin), img = (int)(iw / nwin);
  for (int e = threadIdx.x; e < 32 * p.dp; e += 256) {
    const int tl = e / p.dp, c = e % p.dp;
    const int tok = tok0 + tl;
    const int iy = tok / p.S, ix = tok % p.S;
    const int gy = (win / p.nws) * p.S + iy, gx = (win % p.nws) * p.S + ix;
    const bool tokv = tok < S2, inside = tokv && gy < p.G && gx < p.G, cv = c < p.d;
    const long row = ((long)img * p.G + gy) * p.G + gx;
    float g = 0.f, vv = 0.f, qs = 0.f;
    if (cv && tokv) {
      if (inside) {
        g = ld_any(p.dao, row * p.ld_dao + h * p.d + c, p.dt);
        vv = ld_any(p.qkv, row * p.ld_qkv + 2 * p.C + h * p.d + c, p.dt);
        qs = ld_any(p.qkv, row * p.ld_qkv + h * p.d + c, p.dt) * p.scale;
      } else {
        vv = p.bias ? p.bias[2 * p.C + h * p.d + c] : 0.f;
        qs = (p.bias ? p.bias[h * p.d + c] : 0.f) * p.scale;
      }
    }
    const long idx = (bz * p.NP + tok) * p.dp + c;
    st_any(p.dow, idx, p.dt, g);
    st_any(p.vp, idx, p.dt, vv);
    tg[c][tl] = g;
    tq[c][tl] = qs;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * p.dp; e += 256) {
    const int c = e >> 5, tl = e & 31;
    const long tidx = (bz * p.dp + c) * p.NP + tok0 + tl;
    st_any(p.dowT, tidx, p.dt, tg[c][tl]);
    st_any(p.qsT, tidx, p.dt, tq[c][tl]);
  }
}

extern "C" int vfm_sam_attn_bwd_prep(const void* dao, long ld_dao, const void* qkv, long ld_qkv, const float* bias, int dt, void* dow,
                                     void* dowT, void* vp, void* qsT, int nimg, int G, int S, int H, int d, int dp, int NP,
                                     float scale, void* stream) {
  VFM_CHECK(S > 0 && G > 0 && dp >= d && NP >= S * S, VFM_E_SHAPE, "vfm_sam_attn_bwd_prep: shape");
  SamBwdPrepP p;
  p.dao = dao; p.ld_dao = ld_dao; p.qkv = qkv; p.ld_qkv = ld_qkv; p.bias = bias; p.dt = dt;
  p.dow = dow; p.dowT = dowT; p.vp = vp; p.qsT = qsT;
  p.nimg = nimg; p.G = G; p.S = S; p.nws = (G + S - 1) / S; p.H = H; p.d = d; p.C = H * d; p.dp = dp; p.NP = NP; p.scale = scale;
  const long total = (long)nimg * p.nws * p.nws * H * NP * dp;
  if (total == 0) return VFM_OK;
  if (NP % 32 == 0 && dp <= SAM_BP_MAXDP) {
    const long nb = (long)nimg * p.nws * p.nws * H;
    hipLaunchKernelGGL(k_sam_bwd_prep_tiles, dim3((unsigned)(nb * (NP / 32))), dim3(256), 0, (hipStream_t)stream, p);
    VFM_LAUNCH_CHECK();
    return VFM_OK;
  }
  const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipLaunchKernelGGL(k_sam_bwd_prep, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// batched row softmax for training: rows are grouped in batches of `rpb` rows of which the first `valid` are real; the
// others are written as zeros (they are K rows of later transposed-operand GEMMs).  ds != null: softmax BACKWARD instead,
// ds[r,c] = p[r,c] * (dp[r,c] - sum_j p[r,j] dp[r,j]).
__global__ void k_softmax_rows_b(const float* __restrict__ s, long ld_s, void* __restrict__ out, int out_dt, long ld_o, long rows,
                                 int n, int npad, int rpb, int valid, const void* __restrict__ pin, void* __restrict__ ds) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bool live = (int)(row % rpb) < valid;
  const float* sr = s + row * ld_s;
  if (!ds && n <= 1024) {  // forward, row in registers (see k_softmax_rows)
    float v[16];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = i * 64 + lane;
      v[i] = (live && c < n) ? sr[c] : -INFINITY;
      m = fmaxf(m, v[i]);
    }
    m = live ? wave_max(m) : 0.f;
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      v[i] = __expf(v[i] - m);
      z += v[i];
    }
    const float inv = live ? 1.f / wave_sum(z) : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = i * 64 + lane;
      if (c < npad) st_any(out, row * ld_o + c, out_dt, live ? v[i] * inv : 0.f);
    }
    return;
  }
  if (!ds) {
    float m = -INFINITY, z = 1.f;
    if (live) {
      for (int c = lane; c < n; c += 64) m = fmaxf(m, sr[c]);
      m = wave_max(m);
      z = 0.f;
      for (int c = lane; c < n; c += 64) z += __expf(sr[c] - m);
      z = wave_sum(z);
    }
    const float inv = 1.f / z;
    for (int c = lane; c < npad; c += 64) st_any(out, row * ld_o + c, out_dt, (live && c < n) ? __expf(sr[c] - m) * inv : 0.f);
  } else if (n <= 1024) {  // softmax backward with the row of p and of dp in registers
    float pv[16], dv[16];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = i * 64 + lane;
      const bool ok = live && c < n;
      pv[i] = ok ? ld_any(pin, row * ld_o + c, out_dt) : 0.f;
      dv[i] = ok ? sr[c] : 0.f;
      dot = fmaf(pv[i], dv[i], dot);
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = i * 64 + lane;
      if (c < npad) st_any(ds, row * ld_o + c, out_dt, pv[i] * (dv[i] - dot));
    }
  } else {
    float dot = 0.f;
    if (live) {
      for (int c = lane; c < n; c += 64) dot += ld_any(pin, row * ld_o + c, out_dt) * sr[c];
      dot = wave_sum(dot);
    }
    for (int c = lane; c < npad; c += 64)
      st_any(ds, row * ld_o + c, out_dt, (live && c < n) ? ld_any(pin, row * ld_o + c, out_dt) * (sr[c] - dot) : 0.f);
  }
}
extern "C" int vfm_softmax_rows_batched(const float* scores, long ld_s, void* out, int out_dt, long ld_o, long rows, int n, int npad,
                                        int rows_per_batch, int valid_rows, void* stream) {
  VFM_CHECK(npad >= n && ld_o >= npad && ld_s >= n && rows_per_batch >= valid_rows && valid_rows > 0, VFM_E_SHAPE, "vfm_softmax_rows_batched: shape");
  if (rows == 0) return VFM_OK;
  hipLaunchKernelGGL(k_softmax_rows_b, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, scores, ld_s, out, out_dt, ld_o, rows, n,
                     npad, rows_per_batch, valid_rows, (const void*)nullptr, (void*)nullptr);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
extern "C" int vfm_softmax_rows_bwd(const void* p, const float* dp, long ld_dp, void* ds, int dt, long ld_p, long rows, int n, int npad,
                                    int rows_per_batch, int valid_rows, void* stream) {
  VFM_CHECK(npad >= n && ld_p >= npad && ld_dp >= n && rows_per_batch >= valid_rows && valid_rows > 0, VFM_E_SHAPE, "vfm_softmax_rows_bwd: shape");
  if (rows == 0) return VFM_OK;
  hipLaunchKernelGGL(k_softmax_rows_b, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dp, ld_dp, (void*)nullptr, dt, ld_p, rows, n,
                     npad, rows_per_batch, valid_rows, p, ds);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

// dQaug [nb, NP, Dq] (fp32 or bf16), dkT / dvT [nb, dp, NP] -> token-major dqkv [nimg*G*G, 3C] (padded tokens dropped)
struct SamBwdMergeP {
  const void* dqa; const void* dkT; const void* dvT; int dt;
  const float* rh; const float* rw;
  void* dqkv; long ld;
  int nimg, G, S, nws, H, d, C, dp, NP, Dq;
  float scale;
};
__global__ void k_sam_bwd_merge(SamBwdMergeP p) {
  const long total = (long)p.nimg * p.G * p.G * p.H * p.d;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % p.d);
    long t = idx / p.d;
    const int h = (int)(t % p.H);
    t /= p.H;
    const int gx = (int)(t % p.G);
    t /= p.G;
    const int gy = (int)(t % p.G);
    const int img = (int)(t / p.G);
    const int iy = gy % p.S, ix = gx % p.S;
    const int win = (gy / p.S) * p.nws + gx / p.S, tok = iy * p.S + ix;
    const long bz = ((long)img * p.nws * p.nws + win) * p.H + h;
    const long qrow = (bz * p.NP + tok) * p.Dq;
    float dq = ld_any(p.dqa, qrow + c, p.dt) * p.scale;
    const float* rh = p.rh + (long)iy * p.S * p.d + c;
    const float* rw = p.rw + (long)ix * p.S * p.d + c;
    for (int k = 0; k < p.S; ++k) {
      dq += ld_any(p.dqa, qrow + p.d + k, p.dt) * rh[(long)k * p.d];
      dq += ld_any(p.dqa, qrow + p.d + p.S + k, p.dt) * rw[(long)k * p.d];
    }
    const long tidx = (bz * p.dp + c) * p.NP + tok;
    const long row = ((long)img * p.G + gy) * p.G + gx;
    st_any(p.dqkv, row * p.ld + h * p.d + c, p.dt, dq);
    st_any(p.dqkv, row * p.ld + p.C + h * p.d + c, p.dt, ld_any(p.dkT, tidx, p.dt));
    st_any(p.dqkv, row * p.ld + 2 * p.C + h * p.d + c, p.dt, ld_any(p.dvT, tidx, p.dt));
  }
}
extern "C" int vfm_sam_attn_bwd_merge(const void* dqa, const void* dkT, const void* dvT, int dt, const float* rh, const float* rw,
                                      void* dqkv, long ld, int nimg, int G, int S, int H, int d, int dp, int NP, int Dq, float scale,
                                      void* stream) {
  SamBwdMergeP p;
  p.dqa = dqa; p.dkT = dkT; p.dvT = dvT; p.dt = dt; p.rh = rh; p.rw = rw; p.dqkv = dqkv; p.ld = ld;
  p.nimg = nimg; p.G = G; p.S = S; p.nws = (G + S - 1) / S; p.H = H; p.d = d; p.C = H * d; p.dp = dp; p.NP = NP; p.Dq = Dq; p.scale = scale;
  const long total = (long)nimg * G * G * H * d;
  if (total == 0) return VFM_OK;
  const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipLaunchKernelGGL(k_sam_bwd_merge, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
