// Exact-fp32 attention (parity path): one thread per query (forward, dQ) or per key (dK, dV), K/V or Q/dO tiles
// staged in LDS and broadcast-read.  VALU-bound by design; the bf16 MFMA flash kernels live in attention_bf16.hip.
#include "attn_common.h"

#define AT_TILE 32

template <int D>
__global__ void __launch_bounds__(256) k_attn_f32_fwd(AttnP p) {
  __shared__ float Ks[AT_TILE][D];
  __shared__ float Vs[AT_TILE][D];
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int nq = p.nq_main + p.nq_extra, nk = p.nk_main + p.nk_extra;
  const int qi = blockIdx.x * 256 + threadIdx.x;
  const bool active = qi < nq;
  const long qrow = tok_row(b, active ? qi : 0, p.nq_main, p.B);
  float q[D], o[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    q[j] = ld_any(p.q, qrow * p.ldq + h * D + j, p.dt) * p.scale;
    o[j] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < nk; k0 += AT_TILE) {
    __syncthreads();
    for (int e = threadIdx.x; e < AT_TILE * D; e += 256) {
      const int jj = e / D, c = e % D;
      const int kj = k0 + jj;
      if (kj < nk) {
        const long krow = tok_row(b, kj, p.nk_main, p.B);
        Ks[jj][c] = ld_any(p.k, krow * p.ldk + h * D + c, p.dt);
        Vs[jj][c] = ld_any(p.v, krow * p.ldv + h * D + c, p.dt);
      } else {
        Ks[jj][c] = 0.f;
        Vs[jj][c] = 0.f;
      }
    }
    __syncthreads();
    float s[AT_TILE];
    float tm = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < AT_TILE; ++jj) {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < D; ++c) a += q[c] * Ks[jj][c];
      s[jj] = (k0 + jj < nk) ? a : -INFINITY;
      tm = fmaxf(tm, s[jj]);
    }
    const float mn = fmaxf(m, tm);
    const float alpha = __expf(m - mn);
    l *= alpha;
#pragma unroll
    for (int c = 0; c < D; ++c) o[c] *= alpha;
#pragma unroll
    for (int jj = 0; jj < AT_TILE; ++jj) {
      const float pj = __expf(s[jj] - mn);
      l += pj;
#pragma unroll
      for (int c = 0; c < D; ++c) o[c] += pj * Vs[jj][c];
    }
    m = mn;
  }
  if (active) {
    const float inv = 1.f / l;
    const long orow = tok_row(b, qi, p.nq_main, p.B);
#pragma unroll
    for (int j = 0; j < D; ++j) st_any(p.o, orow * p.ldo + h * D + j, p.dt, o[j] * inv);
    if (p.lse) p.lse[((long)b * p.H + h) * nq + qi] = m + __logf(l);
  }
}

// delta[b,h,i] = sum_j dO[i,j] * O[i,j]
template <int D>
__global__ void k_attn_delta(AttnP p) {
  const int nq = p.nq_main + p.nq_extra;
  const long total = (long)p.B * p.H * nq;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int qi = (int)(i % nq);
    const int bh = (int)(i / nq), b = bh / p.H, h = bh % p.H;
    const long row = tok_row(b, qi, p.nq_main, p.B);
    float a = 0.f;
    for (int j = 0; j < D; ++j) a += ld_any(p.dout, row * p.ld_do + h * D + j, p.dt) * ld_any(p.o, row * p.ldo + h * D + j, p.dt);
    p.delta[i] = a;
  }
}

template <int D>
__global__ void __launch_bounds__(256) k_attn_f32_dq(AttnP p) {
  __shared__ float Ks[AT_TILE][D];
  __shared__ float Vs[AT_TILE][D];
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int nq = p.nq_main + p.nq_extra, nk = p.nk_main + p.nk_extra;
  const int qi = blockIdx.x * 256 + threadIdx.x;
  const bool active = qi < nq;
  const long qrow = tok_row(b, active ? qi : 0, p.nq_main, p.B);
  float q[D], d_o[D], dq[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    q[j] = ld_any(p.q, qrow * p.ldq + h * D + j, p.dt) * p.scale;
    d_o[j] = ld_any(p.dout, qrow * p.ld_do + h * D + j, p.dt);
    dq[j] = 0.f;
  }
  const float lse = active ? p.lse[((long)b * p.H + h) * nq + qi] : 0.f;
  const float dl = active ? p.delta[((long)b * p.H + h) * nq + qi] : 0.f;
  for (int k0 = 0; k0 < nk; k0 += AT_TILE) {
    __syncthreads();
    for (int e = threadIdx.x; e < AT_TILE * D; e += 256) {
      const int jj = e / D, c = e % D;
      const int kj = k0 + jj;
      if (kj < nk) {
        const long krow = tok_row(b, kj, p.nk_main, p.B);
        Ks[jj][c] = ld_any(p.k, krow * p.ldk + h * D + c, p.dt);
        Vs[jj][c] = ld_any(p.v, krow * p.ldv + h * D + c, p.dt);
      } else {
        Ks[jj][c] = 0.f;
        Vs[jj][c] = 0.f;
      }
    }
    __syncthreads();
    for (int jj = 0; jj < AT_TILE; ++jj) {
      if (k0 + jj >= nk) break;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < D; ++c) {
        s += q[c] * Ks[jj][c];
        dp += d_o[c] * Vs[jj][c];
      }
      const float ds = __expf(s - lse) * (dp - dl);
#pragma unroll
      for (int c = 0; c < D; ++c) dq[c] += ds * Ks[jj][c];
    }
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < D; ++j) st_any(p.dq, qrow * p.ld_dq + h * D + j, p.dt, dq[j] * p.scale);
  }
}

// per-key thread: DV=true accumulates dV = P^T dO ; DV=false accumulates dK = dS^T Q * scale
template <int D, bool DV>
__global__ void __launch_bounds__(256) k_attn_f32_dkv(AttnP p) {
  __shared__ float Qs[AT_TILE][D];
  __shared__ float Os[AT_TILE][D];
  __shared__ float Ls[AT_TILE], Ds[AT_TILE];
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int nq = p.nq_main + p.nq_extra, nk = p.nk_main + p.nk_extra;
  const int kj = blockIdx.x * 256 + threadIdx.x;
  const bool active = kj < nk;
  const long krow = tok_row(b, active ? kj : 0, p.nk_main, p.B);
  float kk[D], vv[DV ? 1 : D], acc[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    kk[j] = ld_any(p.k, krow * p.ldk + h * D + j, p.dt);
    if (!DV) vv[j] = ld_any(p.v, krow * p.ldv + h * D + j, p.dt);
    acc[j] = 0.f;
  }
  for (int q0 = 0; q0 < nq; q0 += AT_TILE) {
    __syncthreads();
    for (int e = threadIdx.x; e < AT_TILE * D; e += 256) {
      const int ii = e / D, c = e % D;
      const int qi = q0 + ii;
      if (qi < nq) {
        const long qrow = tok_row(b, qi, p.nq_main, p.B);
        Qs[ii][c] = ld_any(p.q, qrow * p.ldq + h * D + c, p.dt) * p.scale;
        Os[ii][c] = ld_any(p.dout, qrow * p.ld_do + h * D + c, p.dt);
      } else {
        Qs[ii][c] = 0.f;
        Os[ii][c] = 0.f;
      }
    }
    if (threadIdx.x < AT_TILE) {
      const int qi = q0 + threadIdx.x;
      Ls[threadIdx.x] = qi < nq ? p.lse[((long)b * p.H + h) * nq + qi] : INFINITY;
      Ds[threadIdx.x] = qi < nq ? p.delta[((long)b * p.H + h) * nq + qi] : 0.f;
    }
    __syncthreads();
    for (int ii = 0; ii < AT_TILE; ++ii) {
      if (q0 + ii >= nq) break;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < D; ++c) s += Qs[ii][c] * kk[c];
      const float pj = __expf(s - Ls[ii]);
      if (DV) {
#pragma unroll
        for (int c = 0; c < D; ++c) acc[c] += pj * Os[ii][c];
      } else {
        float dp = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) dp += Os[ii][c] * vv[c];
        const float ds = pj * (dp - Ds[ii]);
#pragma unroll
        for (int c = 0; c < D; ++c) acc[c] += ds * Qs[ii][c];  // Qs already carries the scale
      }
    }
  }
  if (active) {
    if (DV) {
#pragma unroll
      for (int j = 0; j < D; ++j) st_any(p.dv, krow * p.ld_dv + h * D + j, p.dt, acc[j]);
    } else {
#pragma unroll
      for (int j = 0; j < D; ++j) st_any(p.dk, krow * p.ld_dk + h * D + j, p.dt, acc[j]);
    }
  }
}

int vfm_attn_f32_fwd_impl(const vfm_attn_desc* d, hipStream_t s) {
  const AttnP p = to_p(d);
  const int nq = d->nq_main + d->nq_extra;
  dim3 grid(cdiv(nq, 256), d->B * d->H);
  if (d->d == 64) hipLaunchKernelGGL(k_attn_f32_fwd<64>, grid, dim3(256), 0, s, p);
  else if (d->d == 80) hipLaunchKernelGGL(k_attn_f32_fwd<80>, grid, dim3(256), 0, s, p);
  else VFM_FAIL(VFM_E_UNSUPPORTED, "vfm_attn: head dim %d", d->d);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}

int vfm_attn_f32_bwd_impl(const vfm_attn_desc* d, hipStream_t s) {
  const AttnP p = to_p(d);
  const int nq = d->nq_main + d->nq_extra, nk = d->nk_main + d->nk_extra;
  VFM_CHECK(d->d == 64, VFM_E_UNSUPPORTED, "vfm_attn_bwd(f32): head dim %d", d->d);
  const long total = (long)d->B * d->H * nq;
  hipLaunchKernelGGL(k_attn_delta<64>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
  hipLaunchKernelGGL(k_attn_f32_dq<64>, dim3(cdiv(nq, 256), d->B * d->H), dim3(256), 0, s, p);
  hipLaunchKernelGGL((k_attn_f32_dkv<64, true>), dim3(cdiv(nk, 256), d->B * d->H), dim3(256), 0, s, p);
  hipLaunchKernelGGL((k_attn_f32_dkv<64, false>), dim3(cdiv(nk, 256), d->B * d->H), dim3(256), 0, s, p);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
