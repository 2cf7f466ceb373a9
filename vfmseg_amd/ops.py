"""Tensor-level wrappers over the C ABI (one Python function per entry point, no autograd here).

All tensors must live on the GPU; every call is asynchronous on torch's current HIP stream.
"""
import ctypes as C
import os

import torch

from .precision import is_half
from . import lib as L
from .lib import (ACT_GELU, ACT_NONE, ACT_QGELU, ACT_RELU, BF16, EP_GELU, EP_GELU_DGELU, EP_MUL, EP_MUL_GELU_GRAD,  # noqa: F401
                  EP_MUL_QGELU_GRAD, EP_NONE, EP_QGELU, EP_RELU, F32)

_ws_cache = {}

# Measurement hook (bench.py installs it; None in production): PROFILE(kind, flops, region) -> finish() | None is called around the
# launches of the two MFMA kernel families so that HIP events bracket exactly one kernel on the stream it is launched on.
# REGION names who is launching ("backbone" inside the ViT engines, "heads" elsewhere).
PROFILE = None
REGION = ["heads"]


class region:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        REGION.append(self.name)

    def __exit__(self, *a):
        REGION.pop()


def workspace(nfloats, device, tag="default"):
    """Grow-only fp32 scratch per (device, tag, STREAM); kernels never allocate.  Per stream: the decoder heads' weight gradients run on a
    side stream (functional.wgrad_stream) - kernels that may run concurrently must not share scratch."""
    key = (str(device), tag, L.stream())
    t = _ws_cache.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty(max(int(nfloats), 1 << 16), dtype=torch.float32, device=device)
        _ws_cache[key] = t
    return t


def ld_pad(cols, dtype):
    """Extra columns for a row-major buffer whose rows would otherwise be a multiple of 8 KiB apart: with such a stride the
    128 rows of a GEMM tile fall on a few L2 channels and the LDS-DMA operand stream loses ~10 % (measured at K = 4096:
    ld 4096 -> 744 TF, ld 4224 -> 815 TF; 128- or 16-byte pads are worse than none)."""
    esz = torch.empty(0, dtype=dtype).element_size()
    return (256 // esz) if (cols * esz) % 8192 == 0 else 0


def empty_ld(rows, cols, dtype, device, zero=False):
    """[rows, cols] view of a buffer whose leading dimension avoids the 8-KiB stride (see ld_pad)."""
    pad = ld_pad(cols, dtype)
    make = torch.zeros if zero else torch.empty
    if pad == 0:
        return make(rows, cols, dtype=dtype, device=device)
    return make(rows, cols + pad, dtype=dtype, device=device)[:, :cols]


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D view"
    return t.stride(0)


def cast(src, dst, colscale=None):
    """dst[r,c] = src[r,c] * colscale[c]; src/dst 2-D row-major views (may be column slices)."""
    lib = L.load()
    L.check(lib.vfm_cast(L.ptr(src), L.dt_of(src), _ld(src), L.ptr(dst), L.dt_of(dst), _ld(dst), src.shape[0], src.shape[1],
                         L.ptr(colscale), L.stream()), "vfm_cast")
    return dst


def transpose(src, dst, pad_rows=None):
    """dst[c, r] = src[r, c]; dst [cols, >=pad_rows]; columns rows..pad_rows-1 zero-filled."""
    lib = L.load()
    rows, cols = src.shape
    pad_rows = rows if pad_rows is None else pad_rows
    L.check(lib.vfm_transpose(L.ptr(src), L.dt_of(src), _ld(src), L.ptr(dst), L.dt_of(dst), _ld(dst), rows, cols, pad_rows,
                              L.stream()), "vfm_transpose")
    return dst


def strided_copy(src, dst, shape, sstr, dstr, accumulate=False):
    lib = L.load()
    n = list(shape) + [1] * (4 - len(shape))
    s = list(sstr) + [0] * (4 - len(sstr))
    d = list(dstr) + [0] * (4 - len(dstr))
    # leading singleton padding must be at the FRONT so that the fastest index stays last
    k = 4 - len(shape)
    n = [1] * k + list(shape)
    s = [0] * k + list(sstr)
    d = [0] * k + list(dstr)
    L.check(lib.vfm_strided_copy(L.ptr(src), L.dt_of(src), L.ptr(dst), L.dt_of(dst), *n, *s, *d, int(accumulate), L.stream()),
            "vfm_strided_copy")
    return dst


class CopyBatch:
    """Device table for vfm_strided_copy_batch: jobs = [(src fp32 tensor, dst tensor, shape, src strides, dst strides[, accumulate
    [, nsum, sum_stride]])], each up to 4-D; run() performs all of them in one launch.  The tensors are kept alive by the table.
    Jobs of one table run concurrently: two jobs must not accumulate into the same destination - sum the partials of one
    reduction with nsum / sum_stride instead."""

    def __init__(self, jobs):
        import struct
        buf = bytearray()
        self.keep, self.max_elems, self.n = [], 0, len(jobs)
        for job in jobs:
            src, dst, shape, sstr, dstr = job[:5]
            acc = int(bool(job[5])) if len(job) > 5 else 0    # optional 6th item: accumulate into an fp32 dst
            nsum, sstride = (int(job[6]), int(job[7])) if len(job) > 7 else (1, 0)
            assert src.dtype == torch.float32 and src.is_cuda and dst.is_cuda and (not acc or dst.dtype == torch.float32)
            k = 4 - len(shape)
            n, s_, d_ = [1] * k + list(shape), [0] * k + list(sstr), [0] * k + list(dstr)
            buf += struct.pack("2Q16q", src.data_ptr(), dst.data_ptr(), L.dt_of(dst), acc, *n, *s_, *d_, nsum, sstride)
            self.keep += [src, dst]
            e = 1
            for v in shape:
                e *= v
            self.max_elems = max(self.max_elems, e)
        self.table = torch.frombuffer(buf, dtype=torch.uint8).clone().to(jobs[0][1].device) if jobs else None

    def run(self):
        if self.n:
            lib = L.load()
            L.check(lib.vfm_strided_copy_batch(L.ptr(self.table), self.n, self.max_elems, L.stream()), "vfm_strided_copy_batch")


def permute_copy(src, perm, dst):
    """dst (contiguous, any dtype) = src.permute(perm) ; src may be any strided <=4-D tensor."""
    v = src.permute(*perm)
    strided_copy(src, dst, list(v.shape), list(v.stride()), list(torch.empty(v.shape, device="meta").stride()))
    return dst


def axpby(x, a, y, b):
    lib = L.load()
    assert x.is_contiguous() and y.is_contiguous() and x.dtype == y.dtype == torch.float32 and x.numel() == y.numel()
    L.check(lib.vfm_axpby(L.ptr(x), float(a), L.ptr(y), float(b), x.numel(), L.stream()), "vfm_axpby")
    return y


def scale_by_device_scalar(y, scalar):
    lib = L.load()
    assert y.is_contiguous() and y.dtype == torch.float32
    L.check(lib.vfm_scale_by_device_scalar(L.ptr(y), L.ptr(scalar), y.numel(), L.stream()), "vfm_scale_by_device_scalar")
    return y


def colsum(x, out, accumulate=False):
    lib = L.load()
    rows, cols = x.shape
    ws = workspace(64 * cols, x.device)
    L.check(lib.vfm_colsum(L.ptr(x), L.dt_of(x), _ld(x), rows, cols, L.ptr(out), int(accumulate), L.ptr(ws), L.stream()),
            "vfm_colsum")
    return out


def dropout_mask(out, p, seed, offset=0):
    lib = L.load()
    assert out.is_contiguous()
    L.check(lib.vfm_dropout_mask(L.ptr(out), L.dt_of(out), out.numel(), float(p), int(seed), int(offset), L.stream()),
            "vfm_dropout_mask")
    return out


def mul_mask(src, mask, dst, rows_per_group=1):
    lib = L.load()
    L.check(lib.vfm_mul_mask(L.ptr(src), L.dt_of(src), _ld(src), L.ptr(mask), L.dt_of(mask), _ld(mask), rows_per_group,
                             L.ptr(dst), L.dt_of(dst), _ld(dst), src.shape[0], src.shape[1], L.stream()), "vfm_mul_mask")
    return dst


def geglu_fwd(h, out):
    lib = L.load()
    rows, c2 = h.shape
    L.check(lib.vfm_geglu_fwd(L.ptr(h), L.dt_of(h), _ld(h), L.ptr(out), L.dt_of(out), _ld(out), rows, c2 // 2, L.stream()),
            "vfm_geglu_fwd")
    return out


def geglu_bwd(h, dout, dh):
    lib = L.load()
    rows, c2 = h.shape
    L.check(lib.vfm_geglu_bwd(L.ptr(h), L.dt_of(h), _ld(h), L.ptr(dout), L.dt_of(dout), _ld(dout), L.ptr(dh), L.dt_of(dh),
                              _ld(dh), rows, c2 // 2, L.stream()), "vfm_geglu_bwd")
    return dh


def swiglu_fwd(h, out, C_):
    lib = L.load()
    L.check(lib.vfm_swiglu_fwd(L.ptr(h), L.dt_of(h), _ld(h), L.ptr(out), L.dt_of(out), _ld(out), h.shape[0], C_, L.stream()),
            "vfm_swiglu_fwd")
    return out


def swiglu_bwd(h, dout, dh, C_):
    lib = L.load()
    L.check(lib.vfm_swiglu_bwd(L.ptr(h), L.dt_of(h), _ld(h), L.ptr(dout), L.dt_of(dout), _ld(dout), L.ptr(dh), L.dt_of(dh),
                               _ld(dh), h.shape[0], C_, L.stream()), "vfm_swiglu_bwd")
    return dh


def rope(x, rows, np_, ncols, d, cos_t, sin_t, inverse=False):
    """In place on columns [0, ncols) of the first `rows` rows of the 2-D view x."""
    lib = L.load()
    L.check(lib.vfm_rope(L.ptr(x), L.dt_of(x), _ld(x), rows, np_, ncols, d, L.ptr(cos_t), L.ptr(sin_t), int(inverse), L.stream()),
            "vfm_rope")
    return x


def act_grad_mul(dy, pre, out, act):
    lib = L.load()
    L.check(lib.vfm_act_grad_mul(L.ptr(dy), L.dt_of(dy), _ld(dy), L.ptr(pre), L.dt_of(pre), _ld(pre), L.ptr(out),
                                 L.dt_of(out), _ld(out), dy.shape[0], dy.shape[1], act, L.stream()), "vfm_act_grad_mul")
    return out


def mask_token_fwd(x, keep, token, out):
    lib = L.load()
    L.check(lib.vfm_mask_token_fwd(L.ptr(x), L.ptr(keep), L.ptr(token), L.ptr(out), x.shape[0], x.shape[1], L.stream()),
            "vfm_mask_token_fwd")
    return out


def mask_token_bwd(dout, keep, dx, dtoken):
    lib = L.load()
    ws = workspace(64 * dout.shape[1], dout.device)
    L.check(lib.vfm_mask_token_bwd(L.ptr(dout), L.ptr(keep), L.ptr(dx), L.ptr(dtoken), L.ptr(ws), dout.shape[0], dout.shape[1],
                                   L.stream()), "vfm_mask_token_bwd")


def _split_key(t):
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version)


def _register_split_out(t, t3):
    """t3 is the split-bf16 image (vfm_split3 pattern 0) of the fp32 tensor t, written by t's producer: the GEMM that consumes t finds it on
    the tensor OBJECT (ops.split3) instead of running vfm_split3.  When the producer skipped the fp32 copy, t holds no data at all."""
    try:
        t._vfm_split3_out = (_split_key(t), t3)
    except AttributeError:
        pass


def layernorm_fwd(x, w, b, eps, y, stats=None, split_out=None):
    """split_out (bf16x3 mode, y fp32, C % 256 == 0): "also" = y and its split-bf16 image, "only" = the image alone (y is then NOT written:
    for outputs whose one consumer is a GEMM's A operand); the image is registered on y for ops.split3."""
    lib = L.load()
    rows, c = x.shape
    if split_out and _split3_on() and y.dtype == torch.float32 and c % 256 == 0 and c // 256 in (1, 2, 4, 5, 8) and y.shape[1] == c:
        kp = (c + 63) // 64 * 64
        y3 = torch.empty(rows, 3 * kp, dtype=L.half_dtype(), device=x.device)
        L.check(lib.vfm_layernorm_fwd_split3(L.ptr(x), _ld(x), L.ptr(w), L.ptr(b), float(eps), None if split_out == "only" else L.ptr(y), _ld(y),
                                             L.ptr(y3), y3.stride(0), kp, L.ptr(stats), rows, c, L.stream()), "vfm_layernorm_fwd_split3")
        _register_split_out(y, y3)
        return y
    L.check(lib.vfm_layernorm_fwd(L.ptr(x), _ld(x), L.ptr(w), L.ptr(b), float(eps), L.ptr(y), L.dt_of(y), _ld(y),
                                  L.ptr(stats), rows, c, L.stream()), "vfm_layernorm_fwd")
    return y


def layernorm_dropout_fwd(x, w, b, eps, y, stats, y_drop, mask, p, seed, offset=0):
    """LN forward that also writes the dropout multiplier `mask` and y_drop = y * mask (bf16 outputs, C % 256 == 0)."""
    lib = L.load()
    rows, c = x.shape
    assert is_half(y.dtype) and is_half(y_drop.dtype) and is_half(mask.dtype)
    L.check(lib.vfm_layernorm_dropout_fwd(L.ptr(x), _ld(x), L.ptr(w), L.ptr(b), float(eps), L.ptr(y), _ld(y), L.ptr(stats),
                                          L.ptr(y_drop), _ld(y_drop), L.ptr(mask), _ld(mask), float(p), int(seed), int(offset),
                                          rows, c, L.stream()), "vfm_layernorm_dropout_fwd")
    return y


def layernorm_bwd_scaled(dy, x, w, stats, dx, t_out, t_scale, accumulate_dx=False):
    """LN backward into dx (fp32) that also emits t_out = bf16(dx_new * t_scale[c])."""
    lib = L.load()
    rows, c = x.shape
    assert is_half(t_out.dtype) and t_scale.dtype == torch.float32
    L.check(lib.vfm_layernorm_bwd_scaled(L.ptr(dy), L.dt_of(dy), _ld(dy), L.ptr(x), _ld(x), L.ptr(w), L.ptr(stats), L.ptr(dx),
                                         _ld(dx), int(accumulate_dx), L.ptr(t_out), _ld(t_out), L.ptr(t_scale), rows, c,
                                         L.stream()), "vfm_layernorm_bwd_scaled")
    return dx


def layernorm_bwd(dy, x, w, stats, dx, accumulate_dx=False, dw=None, db=None):
    lib = L.load()
    rows, c = x.shape
    ws = workspace(2 * 128 * c, x.device) if (dw is not None or db is not None) else None
    L.check(lib.vfm_layernorm_bwd(L.ptr(dy), L.dt_of(dy), _ld(dy), L.ptr(x), _ld(x), L.ptr(w), L.ptr(stats), L.ptr(dx),
                                  _ld(dx), int(accumulate_dx), L.ptr(dw), L.ptr(db), L.ptr(ws), rows, c, L.stream()),
            "vfm_layernorm_bwd")
    return dx


def groupnorm_fwd(x, w, b, eps, groups, act, y, stats, B, P):
    """x fp32 [B*P, C] contiguous; stats fp32 [B, G, 2]."""
    lib = L.load()
    c = x.shape[1]
    ws = workspace(B * groups * 2 + B * 64 * 2 * c, x.device)
    L.check(lib.vfm_groupnorm_fwd(L.ptr(x), L.ptr(w), L.ptr(b), float(eps), groups, act, L.ptr(y), L.dt_of(y), L.ptr(stats),
                                  L.ptr(ws), B, P, c, L.stream()), "vfm_groupnorm_fwd")
    return y


def groupnorm_bwd(dy, x, w, b, stats, groups, act, dx, dw, db, B, P):
    lib = L.load()
    c = x.shape[1]
    ws = workspace(B * groups * 2 + B * 64 * 2 * c, x.device)
    L.check(lib.vfm_groupnorm_bwd(L.ptr(dy), L.dt_of(dy), L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(stats), groups, act, L.ptr(dx),
                                  L.ptr(dw), L.ptr(db), L.ptr(ws), B, P, c, L.stream()), "vfm_groupnorm_bwd")
    return dx


def bn_moments(x, sums):
    lib = L.load()
    rows, c = x.shape
    ws = workspace(64 * 2 * c, x.device)
    L.check(lib.vfm_bn_moments(L.ptr(x), rows, c, L.ptr(sums), L.ptr(ws), L.stream()), "vfm_bn_moments")
    return sums


def bn_finalize(sums, count, mean_var, running_mean=None, running_var=None, momentum=0.1):
    lib = L.load()
    L.check(lib.vfm_bn_finalize(L.ptr(sums), float(count), L.ptr(mean_var), L.ptr(running_mean), L.ptr(running_var),
                                float(momentum), sums.shape[1], L.stream()), "vfm_bn_finalize")
    return mean_var


def bn_apply(x, mean_var, w, b, eps, act, y):
    lib = L.load()
    rows, c = x.shape
    L.check(lib.vfm_bn_apply(L.ptr(x), L.ptr(mean_var), L.ptr(w), L.ptr(b), float(eps), act, L.ptr(y), L.dt_of(y), rows, c,
                             L.stream()), "vfm_bn_apply")
    return y


def bn_bwd_reduce(dy, x, mean_var, w, b, eps, act, sums_dy):
    lib = L.load()
    rows, c = x.shape
    ws = workspace(64 * 2 * c, x.device)
    L.check(lib.vfm_bn_bwd_reduce(L.ptr(dy), L.dt_of(dy), L.ptr(x), L.ptr(mean_var), L.ptr(w), L.ptr(b), float(eps), act,
                                  L.ptr(sums_dy), L.ptr(ws), rows, c, L.stream()), "vfm_bn_bwd_reduce")
    return sums_dy


def bn_bwd_apply(dy, x, mean_var, w, b, eps, act, sums_dy, total_rows, dx):
    lib = L.load()
    rows, c = x.shape
    L.check(lib.vfm_bn_bwd_apply(L.ptr(dy), L.dt_of(dy), L.ptr(x), L.ptr(mean_var), L.ptr(w), L.ptr(b), float(eps), act,
                                 L.ptr(sums_dy), float(total_rows), L.ptr(dx), rows, c, L.stream()), "vfm_bn_bwd_apply")
    return dx


def split3(x, pattern, trans=False, cache=False):
    """fp32 operand x [rows, K] ([K, rows] if trans) -> bf16 [rows, 3 ceil64(K)] split form (vfm_split3).  cache=True keeps the result on
    the tensor OBJECT (weights packed once by the engines: same object every call) and rebuilds it when the tensor's version counter
    moves (LoRA-merged weights are re-packed in place); activations are fresh objects and never hit."""
    if pattern == 0 and not trans:   # an image its producer wrote (LayerNorm / attention / GEMM epilogue in the bf16x3 mode)
        hit = getattr(x, "_vfm_split3_out", None)
        if hit is not None and hit[0] == _split_key(x):
            return hit[1]
    if cache:
        from .optim import PARAM_EPOCH   # the fused AdamW / LoRA re-pack kernels rewrite packed operands behind torch's version counters
        key = (PARAM_EPOCH[0], x._version, x.data_ptr(), pattern, trans, tuple(x.shape), tuple(x.stride()))
        hit = getattr(x, "_vfm_split3", None)
        if hit is not None and hit[0] == key:
            return hit[1]
    rows, K = (x.shape[1], x.shape[0]) if trans else (x.shape[0], x.shape[1])
    sr, sc = (x.stride(1), x.stride(0)) if trans else (x.stride(0), x.stride(1))
    kp = (K + 63) // 64 * 64
    out = torch.empty(rows, 3 * kp, dtype=L.half_dtype(), device=x.device)
    L.check(L.load().vfm_split3(L.ptr(x), sr, sc, L.ptr(out), out.stride(0), rows, K, pattern, L.stream()), "vfm_split3")
    if cache:
        try:
            x._vfm_split3 = (key, out)
        except AttributeError:
            pass
    return out


def gemm(a, b, c, *, alpha=1.0, bias=None, bias_mod=0, colscale=None, residual=None, ep_mode=EP_NONE, aux=None, c2=None,
         trans_a=False, trans_b=False, kb_rows=0, c_split=None):
    """c[M,N] = epilogue(alpha * A @ B^T).  a: [M,K] (or [K,M] if trans_a), b: [N,K] (or [K,N] if trans_b); 2-D views
    (or 3-D batched with equal batch).  bf16 inputs require K-contiguous operands with K % 64 == 0.
    In the bf16x3 mode (precision.split3()) an fp32 x fp32 2-D product is computed by the bf16 MFMA kernels on split operands;
    c_split="only" (that mode, c fp32, N % 64 == 0, no residual / c2) makes the epilogue write the split-bf16 image of the result
    INSTEAD of c (c_dt VFM_SPLIT3) and registers it on c for ops.split3: for a result whose one consumer is the next GEMM's A operand."""
    if a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2 and not kb_rows and _split3_on():
        a3 = split3(a, 0, trans=trans_a)
        b3 = split3(b, 1, trans=trans_b, cache=not b.requires_grad)
        c3 = None
        if (c_split == "only" and c.dtype == torch.float32 and c.dim() == 2 and c.shape[1] % 64 == 0 and residual is None and c2 is None
                and ep_mode in (EP_NONE, EP_GELU)):
            c3 = torch.empty(c.shape[0], 3 * c.shape[1], dtype=L.half_dtype(), device=c.device)
        gemm(a3, b3, c, alpha=alpha, bias=bias, bias_mod=bias_mod, colscale=colscale, residual=residual, ep_mode=ep_mode, aux=aux, c2=c2, c_split=c3)
        if c3 is not None:
            _register_split_out(c, c3)
        return c
    lib = L.load()
    d = gemm_desc(a, b, c, alpha=alpha, bias=bias, bias_mod=bias_mod, colscale=colscale, residual=residual, ep_mode=ep_mode, aux=aux, c2=c2,
                  trans_a=trans_a, trans_b=trans_b, kb_rows=kb_rows)
    if torch.is_tensor(c_split):   # the image replaces c as the output
        d.C, d.c_dt, d.ldc, d.c_plane = L.ptr(c_split), L.SPLIT3, c_split.stride(0), c.shape[1]
    fin = PROFILE("gemm", 2.0 * d.M * d.N * d.K * d.batch, REGION[-1]) if (PROFILE is not None and is_half(a.dtype)) else None
    L.check(lib.vfm_gemm(C.byref(d), L.stream()), "vfm_gemm")
    if fin is not None:
        fin()
    return c


def gemm_desc(a, b, c, *, alpha=1.0, bias=None, bias_mod=0, colscale=None, residual=None, ep_mode=EP_NONE, aux=None, c2=None,
              trans_a=False, trans_b=False, kb_rows=0, d=None):
    """The vfm_gemm_desc of gemm(a, b, c, ...) (filled into `d` when given: an entry of a launch plan)."""
    d = L.GemmDesc() if d is None else d
    batched = a.dim() == 3
    a2 = a[0] if batched else a
    b2 = b[0] if batched else b
    c2d = c[0] if batched else c
    if trans_a:
        K, M = a2.shape
        d.sa_m, d.sa_k = a2.stride(1), a2.stride(0)
    else:
        M, K = a2.shape
        d.sa_m, d.sa_k = a2.stride(0), a2.stride(1)
    if trans_b:
        Kb, N = b2.shape
        d.sb_n, d.sb_k = b2.stride(1), b2.stride(0)
    else:
        N, Kb = b2.shape
        d.sb_n, d.sb_k = b2.stride(0), b2.stride(1)
    if kb_rows:
        assert trans_b and Kb == kb_rows and K >= Kb, "kb_rows: b holds the first kb_rows of the K (token) dimension"
        d.kb_rows = int(kb_rows)
    else:
        assert K == Kb, f"gemm K mismatch {K} vs {Kb}"
    assert a.dtype == b.dtype
    assert c2d.shape[0] == M and c2d.shape[1] == N and c2d.stride(1) == 1
    d.A, d.B, d.C = L.ptr(a), L.ptr(b), L.ptr(c)
    d.in_dt, d.c_dt = L.dt_of(a), L.dt_of(c)
    d.M, d.N, d.K = M, N, K
    d.ldc = c2d.stride(0)
    d.alpha = float(alpha)
    d.bias, d.bias_mod = L.ptr(bias), int(bias_mod)
    d.colscale = L.ptr(colscale)
    if residual is not None:
        r2 = residual[0] if batched else residual
        d.residual, d.r_dt, d.ldr = L.ptr(residual), L.dt_of(residual), r2.stride(0)
    d.ep_mode = ep_mode
    if aux is not None:
        d.aux, d.aux_dt, d.ld_aux = L.ptr(aux), L.dt_of(aux), aux.stride(0)
    if c2 is not None:
        d.C2, d.c2_dt, d.ldc2 = L.ptr(c2), L.dt_of(c2), c2.stride(0)
    if batched:
        d.batch, d.stride_a, d.stride_b, d.stride_c = a.shape[0], a.stride(0), b.stride(0), c.stride(0)
    else:
        d.batch = 1
    return d


class Plan:
    """A launch plan (include/vfmseg_hip.h "launch plans"): entries are appended with the same tensor arguments the one-by-one wrappers
    take, run() replays them through vfm_run_plan in ONE ctypes call.  The tensors are kept alive by the plan; their addresses must stay
    what they were when the entry was appended (persistent buffers), except for fields patched through `entry(i)` before a run."""

    def __init__(self, region="backbone"):
        self.ops, self.keep, self.arr, self.region = [], [], None, region

    def _new(self, kind, prof=L.PROF_NONE, flops=0.0, *tensors):
        op = L.PlanOp()
        op.kind, op.prof_kind, op.flops = kind, prof, float(flops)
        self.ops.append(op)
        self.keep.extend(t for t in tensors if t is not None)
        self.arr = None
        return op

    def gemm(self, a, b, c, **kw):
        op = self._new(L.OP_GEMM, L.PROF_GEMM if is_half(a.dtype) else L.PROF_NONE, 0.0, a, b, c, kw.get("bias"), kw.get("colscale"),
                       kw.get("residual"), kw.get("aux"), kw.get("c2"))
        d = gemm_desc(a, b, c, d=op.u.gemm, **kw)
        op.flops = 2.0 * d.M * d.N * d.K * d.batch
        return len(self.ops) - 1

    def layernorm_fwd(self, x, w, b, eps, y, stats):
        op = self._new(L.OP_LN_FWD, L.PROF_NONE, 0.0, x, w, b, y, stats)
        u = op.u.ln_fwd
        u.x, u.ld_x, u.w, u.b, u.eps, u.y, u.y_dt, u.ld_y, u.stats = L.ptr(x), _ld(x), L.ptr(w), L.ptr(b), float(eps), L.ptr(y), L.dt_of(y), _ld(y), L.ptr(stats)
        u.rows, u.C = x.shape
        return len(self.ops) - 1

    def layernorm_dropout_fwd(self, x, w, b, eps, y, stats, y_drop, mask, p, offset):
        """`offset` is relative to the rng_base given to run()."""
        assert is_half(y.dtype) and is_half(y_drop.dtype) and is_half(mask.dtype)
        op = self._new(L.OP_LN_DROPOUT_FWD, L.PROF_NONE, 0.0, x, w, b, y, stats, y_drop, mask)
        u = op.u.ln_drop
        u.x, u.ld_x, u.w, u.b, u.eps, u.y, u.ld_y, u.stats = L.ptr(x), _ld(x), L.ptr(w), L.ptr(b), float(eps), L.ptr(y), _ld(y), L.ptr(stats)
        u.y_drop, u.ld_yd, u.mask, u.ld_mask, u.p, u.offset = L.ptr(y_drop), _ld(y_drop), L.ptr(mask), _ld(mask), float(p), int(offset)
        u.rows, u.C = x.shape
        return len(self.ops) - 1

    def layernorm_bwd_scaled(self, dy, x, w, stats, dx, t_out, t_scale, accumulate_dx=False):
        assert is_half(t_out.dtype) and t_scale.dtype == torch.float32
        op = self._new(L.OP_LN_BWD_SCALED, L.PROF_NONE, 0.0, dy, x, w, stats, dx, t_out, t_scale)
        u = op.u.ln_bwd
        u.dy, u.dy_dt, u.ld_dy, u.x, u.ld_x, u.w, u.stats = L.ptr(dy), L.dt_of(dy), _ld(dy), L.ptr(x), _ld(x), L.ptr(w), L.ptr(stats)
        u.dx, u.ld_dx, u.accumulate_dx, u.t_out, u.ld_t, u.t_scale = L.ptr(dx), _ld(dx), int(accumulate_dx), L.ptr(t_out), _ld(t_out), L.ptr(t_scale)
        u.rows, u.C = x.shape
        return len(self.ops) - 1

    def attn_fwd(self, q, k, v, o, lse, B, H, d, nq_main, nq_extra, nk_main, nk_extra, scale):
        fl = 4.0 * B * H * (nq_main + nq_extra) * (nk_main + nk_extra) * d
        op = self._new(L.OP_ATTN_FWD, L.PROF_ATTN_FWD if is_half(q.dtype) else L.PROF_NONE, fl, q, k, v, o, lse)
        C.memmove(C.byref(op.u.attn), C.byref(_attn_desc(q, k, v, o, B, H, d, nq_main, nq_extra, nk_main, nk_extra, scale, lse)), C.sizeof(L.AttnDesc))
        return len(self.ops) - 1

    def attn_bwd(self, q, k, v, o, lse, dout, dq, dk, dv, B, H, d, nq_main, nq_extra, nk_main, nk_extra, scale):
        fl = 10.0 * B * H * (nq_main + nq_extra) * (nk_main + nk_extra) * d
        delta = _attn_bwd_delta(q, B, H, nq_main + nq_extra)
        op = self._new(L.OP_ATTN_BWD, L.PROF_ATTN_BWD if is_half(q.dtype) else L.PROF_NONE, fl, q, k, v, o, lse, dout, dq, dk, dv, delta)
        a = _attn_desc(q, k, v, o, B, H, d, nq_main, nq_extra, nk_main, nk_extra, scale, lse)
        a.dout, a.ld_do = L.ptr(dout), dout.stride(0)
        a.dq, a.dk, a.dv = L.ptr(dq), L.ptr(dk), L.ptr(dv)
        a.ld_dq, a.ld_dk, a.ld_dv = dq.stride(0), dk.stride(0), dv.stride(0)
        a.delta = L.ptr(delta)
        C.memmove(C.byref(op.u.attn), C.byref(a), C.sizeof(L.AttnDesc))
        return len(self.ops) - 1

    def cast(self, src, dst, colscale=None):
        op = self._new(L.OP_CAST, L.PROF_NONE, 0.0, src, dst, colscale)
        u = op.u.cast
        u.src, u.src_dt, u.ld_src, u.dst, u.dst_dt, u.ld_dst = L.ptr(src), L.dt_of(src), _ld(src), L.ptr(dst), L.dt_of(dst), _ld(dst)
        u.rows, u.cols = src.shape
        u.colscale = L.ptr(colscale)
        return len(self.ops) - 1

    def strided_copy(self, src, dst, shape, sstr, dstr, accumulate=False):
        op = self._new(L.OP_STRIDED_COPY, L.PROF_NONE, 0.0, src, dst)
        u = op.u.copy
        k = 4 - len(shape)
        u.src, u.src_dt, u.dst, u.dst_dt = L.ptr(src), L.dt_of(src), L.ptr(dst), L.dt_of(dst)
        for i, (n, s_, d_) in enumerate(zip([1] * k + list(shape), [0] * k + list(sstr), [0] * k + list(dstr))):
            u.n[i], u.ss[i], u.ds[i] = n, s_, d_
        u.accumulate = int(accumulate)
        return len(self.ops) - 1

    def _array(self):
        if self.arr is None:
            self.arr = (L.PlanOp * len(self.ops))()
            for i, op in enumerate(self.ops):
                C.memmove(C.byref(self.arr[i]), C.byref(op), C.sizeof(L.PlanOp))
        return self.arr

    def entry(self, i):
        """The i-th entry of the array run() hands to the library (for per-run patches of pointer fields)."""
        return self._array()[i]

    def run(self, seed=0, rng_base=0, start=0, count=None):
        arr = self._array()
        n = len(self.ops) - start if count is None else count
        first = C.cast(C.byref(arr, start * C.sizeof(L.PlanOp)), C.POINTER(L.PlanOp))
        L.check(L.load().vfm_run_plan(first, n, int(seed), int(rng_base), L.stream()), "vfm_run_plan")

    def __len__(self):
        return len(self.ops)


def prof_config(every):
    L.check(L.load().vfm_prof_config(int(every)), "vfm_prof_config")


def prof_read(cap=65536):
    """[(kind name, flops, ms)] of the launches the plan sampler timed since the last read (waits for their events)."""
    buf = (C.c_double * (3 * cap))()
    n = L.load().vfm_prof_read(buf, cap)
    if n < 0:
        L.check(n, "vfm_prof_read")
    return [(L.PROF_NAMES.get(int(buf[3 * i]), "other"), buf[3 * i + 1], buf[3 * i + 2]) for i in range(n)]


class LoraPackTable:
    """Device-resident table for vfm_lora_pack: one entry per adapter site (A, B fp32 parameters and the packed operands they
    feed).  Built once per packing (pointers are stable: parameters live in the optimiser's flat buffer, operands in the
    engine's packed dict); `run()` re-packs every site with one launch."""

    def __init__(self, sites, dt_tensor):
        import struct
        buf = bytearray()
        self.max_elems = 0
        self.keep = []
        for (A, B, a, at, w, wt, r, K, N, Kw) in sites:
            assert A.dtype == torch.float32 and B.dtype == torch.float32 and A.is_contiguous() and B.is_contiguous()
            self.keep += [A, B, a, at, w, wt]
            buf += struct.pack("6Q8q", A.data_ptr(), B.data_ptr(), a.data_ptr(), at.data_ptr(), w.data_ptr(),
                               wt.data_ptr() if wt is not None else 0, r, K, N, Kw, a.stride(0), at.stride(0), w.stride(0),
                               wt.stride(0) if wt is not None else 0)
            self.max_elems = max(self.max_elems, r * (K + N))
        self.n = len(sites)
        self.dt = L.dt_of(dt_tensor)
        dev = sites[0][2].device
        self.table = torch.frombuffer(buf, dtype=torch.uint8).clone().to(dev)
        self.ptrs = [s_[0].data_ptr() for s_ in sites] + [s_[1].data_ptr() for s_ in sites]

    def valid_for(self, sites):
        return self.n == len(sites) and self.ptrs == [s_[0].data_ptr() for s_ in sites] + [s_[1].data_ptr() for s_ in sites]

    def run(self):
        lib = L.load()
        L.check(lib.vfm_lora_pack(L.ptr(self.table), self.n, self.max_elems, self.dt, L.stream()), "vfm_lora_pack")


def slab_reduce(slabs, rows_used, dst, sp, sq, alpha=1.0, accumulate=False):
    """dst[p*sp + q*sq] (+)= alpha * sum_k slabs[k, p, q] for p < rows_used (slabs fp32 [kch, P, Q], dst fp32)."""
    lib = L.load()
    kch, P, Q = slabs.shape
    assert slabs.is_contiguous() and slabs.dtype == torch.float32 and dst.dtype == torch.float32
    L.check(lib.vfm_slab_reduce(L.ptr(slabs), kch, P, Q, int(rows_used), float(alpha), L.ptr(dst), int(sp), int(sq), int(accumulate),
                                L.stream()), "vfm_slab_reduce")
    return dst


def gemm_splitk_bt(at, y, slabs, kch):
    """Split-K weight gradient: slabs[z] = at[:, z*ck:(z+1)*ck] @ y[z*ck:(z+1)*ck, :]  (z < kch, ck = at.shape[1] / kch).
    at [P, Mp] bf16 (tokens zero-padded to Mp), y [M, Q] bf16 consumed in place as the [K, N] operand (rows >= M are
    clamped in the kernel - `at` is zero there), slabs fp32 [kch, P, Q]."""
    lib = L.load()
    d = L.GemmDesc()
    P, mp = at.shape
    M, Q = y.shape
    ck = mp // kch
    assert ck * kch == mp and ck % 64 == 0 and slabs.shape == (kch, P, Q)
    d.A, d.B, d.C = L.ptr(at), L.ptr(y), L.ptr(slabs)
    d.in_dt, d.c_dt = L.dt_of(at), L.dt_of(slabs)
    d.M, d.N, d.K = P, Q, ck
    d.sa_m, d.sa_k = at.stride(0), 1
    d.sb_n, d.sb_k = 1, y.stride(0)
    d.ldc = slabs.stride(1)
    d.alpha = 1.0
    d.batch, d.stride_a, d.stride_b, d.stride_c = kch, ck, ck * y.stride(0), slabs.stride(0)
    d.kb_rows = int(M)
    fin = PROFILE("gemm_tn", 2.0 * P * Q * mp, REGION[-1]) if PROFILE is not None else None
    L.check(lib.vfm_gemm(C.byref(d), L.stream()), "vfm_gemm")
    if fin is not None:
        fin()
    return slabs


def gemm_splitk_tn(xs, y, slabs, kch):
    """Split-K weight gradient with BOTH operands token-major: slabs[z] = xs[z*ck:(z+1)*ck, :]^T @ y[z*ck:(z+1)*ck, :]
    (xs [M, P], y [M, Q] bf16 views consumed in place, P % 8 == 0; token rows >= M read as zeros), slabs fp32 [kch, P, Q],
    ck = ceil64(M) / kch."""
    lib = L.load()
    d = L.GemmDesc()
    M, P = xs.shape
    M2, Q = y.shape
    assert M == M2 and xs.stride(1) == 1 and y.stride(1) == 1
    mp = (M + 63) // 64 * 64
    ck = mp // kch
    assert ck * kch == mp and ck % 64 == 0 and slabs.shape == (kch, P, Q)
    d.A, d.B, d.C = L.ptr(xs), L.ptr(y), L.ptr(slabs)
    d.in_dt, d.c_dt = L.dt_of(xs), L.dt_of(slabs)
    d.M, d.N, d.K = P, Q, ck
    d.sa_m, d.sa_k = 1, xs.stride(0)
    d.sb_n, d.sb_k = 1, y.stride(0)
    d.ldc = slabs.stride(1)
    d.alpha = 1.0
    d.batch, d.stride_a, d.stride_b, d.stride_c = kch, ck * xs.stride(0), ck * y.stride(0), slabs.stride(0)
    d.kb_rows = int(M)
    fin = PROFILE("gemm_tn", 2.0 * P * Q * mp, REGION[-1]) if PROFILE is not None else None
    L.check(lib.vfm_gemm(C.byref(d), L.stream()), "vfm_gemm")
    if fin is not None:
        fin()
    return slabs


def gemm_tn_batched(xs, y, out, valid_rows, alpha=1.0):
    """out[l] = alpha * xs[l]^T @ y[l] for independent problems l (one weight gradient per layer in ONE launch): xs [L, M, P],
    y [L, M, Q] bf16 views (unit column stride, P % 8 == 0, Q % 8 == 0), out fp32 [L, P, Q]; token rows >= valid_rows read as zeros."""
    lib = L.load()
    d = L.GemmDesc()
    Lb, M, P = xs.shape
    _, M2, Q = y.shape
    assert M == M2 and xs.stride(2) == 1 and y.stride(2) == 1 and out.shape == (Lb, P, Q) and out.is_contiguous()
    d.A, d.B, d.C = L.ptr(xs), L.ptr(y), L.ptr(out)
    d.in_dt, d.c_dt = L.dt_of(xs), L.dt_of(out)
    d.M, d.N, d.K = P, Q, (valid_rows + 63) // 64 * 64
    d.sa_m, d.sa_k = 1, xs.stride(1)
    d.sb_n, d.sb_k = 1, y.stride(1)
    d.ldc = Q
    d.alpha = float(alpha)
    d.batch, d.stride_a, d.stride_b, d.stride_c = Lb, xs.stride(0), y.stride(0), P * Q
    d.kb_rows = -int(valid_rows)
    fin = PROFILE("gemm_tn", 2.0 * P * Q * valid_rows * Lb, REGION[-1]) if PROFILE is not None else None
    L.check(lib.vfm_gemm(C.byref(d), L.stream()), "vfm_gemm")
    if fin is not None:
        fin()
    return out


def _split3_on():
    from .precision import split3 as _s3
    return _s3()


def tune(key, value):
    lib = L.load()
    L.check(lib.vfm_tune(key.encode(), int(value)), "vfm_tune")


def _attn_desc(q, k, v, o, B, H, d, nq_main, nq_extra, nk_main, nk_extra, scale, lse):
    a = L.AttnDesc()
    a.q, a.k, a.v, a.o = L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o)
    a.dt = L.dt_of(q)
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(0), k.stride(0), v.stride(0), o.stride(0)
    a.B, a.H, a.d = B, H, d
    a.nq_main, a.nq_extra, a.nk_main, a.nk_extra = nq_main, nq_extra, nk_main, nk_extra
    a.scale = float(scale)
    a.lse = L.ptr(lse)
    return a


def _qkv_split3(q, k, v, hd):
    """q, k, v fp32 [rows, hd] that are the three column thirds of ONE packed [rows, 3 hd] buffer (self-attention: the QKV projection's
    output) -> the vfm_split3(pattern 1) image of the whole buffer, bf16 [rows, 9 hd]: hi of q / k / v at columns 0 / hd / 2 hd, lo at
    3 hd + the same.  Kept on the buffer object (views share their base) so that the backward reuses the forward's split; None when the
    operands are not laid out that way."""
    base = q._base if q._base is not None else None
    if (base is None or k._base is not base or v._base is not base or q.stride(1) != 1 or q.stride(0) != 3 * hd or k.stride(0) != 3 * hd
            or v.stride(0) != 3 * hd or k.storage_offset() - q.storage_offset() != hd or v.storage_offset() - q.storage_offset() != 2 * hd
            or q.storage_offset() % (3 * hd) != 0):
        return None
    rows = q.shape[0]
    hit = getattr(base, "_vfm_qkv3", None)
    key = (base._version, q.storage_offset(), rows)
    if hit is not None and hit[0] == key:
        return hit[1]
    whole = q.as_strided((rows, 3 * hd), (3 * hd, 1), q.storage_offset())
    full = split3(whole, 1)
    try:
        base._vfm_qkv3 = (key, full)
    except AttributeError:
        pass
    return full


_attn_bwd_ws = {}


def _attn_bwd_delta(q, B, H, nq):
    """delta[B, H, nq] + the zero-on-entry scratch of the extra-token rows (the kernels leave it zeroed): one buffer per shape"""
    key = (str(q.device), B, H, nq)
    delta = _attn_bwd_ws.get(key)
    if delta is None:
        delta = _attn_bwd_ws[key] = torch.zeros(B * H * nq + B * H * 192, dtype=torch.float32, device=q.device)
    return delta


def attn_fwd(q, k, v, o, lse, B, H, d, nq_main, nq_extra, nk_main, nk_extra, scale, keep_split=False, o_split=None):
    """q,k,v,o: 2-D row-major views [rows, >=H*d] (column slices of a packed qkv buffer are fine).  keep_split (bf16x3 mode): a backward
    pass will follow - split the whole packed qkv buffer once and keep the image for it (a prediction splits only k and v)."""
    lib = L.load()
    if q.dtype == torch.float32 and d == 64 and _split3_on() and k.shape[1] == H * d and v.shape[1] == H * d:
        # bf16x3 mode: K and V as split bf16 operands ([hi | lo | hi] per row), Q split in the kernel, fp32 softmax and output
        full = _qkv_split3(q, k, v, H * d) if keep_split else None
        if full is not None:
            k3, v3, lo = full[:, H * d:], full[:, 2 * H * d:], 3 * H * d
        else:
            k3, v3, lo = split3(k, 1), split3(v, 1), H * d
        a = _attn_desc(q, k3, v3, o, B, H, d, nq_main, nq_extra, nk_main, nk_extra, scale, lse)
        a.dt = L.dt_of(q)
        if o_split and o.shape[1] == H * d and o.stride(1) == 1:
            # the output as the split operand of the projection GEMM that consumes it (o_split "only": no fp32 copy at all)
            o3 = torch.empty(o.shape[0], 3 * H * d, dtype=L.half_dtype(), device=o.device)
            if o_split == "only":
                a.o = None
            L.check(lib.vfm_attn_fwd_x3_split(C.byref(a), lo, L.ptr(o3), o3.stride(0), H * d, L.stream()), "vfm_attn_fwd_x3_split")
            _register_split_out(o, o3)
            return o
        L.check(lib.vfm_attn_fwd_x3(C.byref(a), lo, L.stream()), "vfm_attn_fwd_x3")
        return o
    a = _attn_desc(q, k, v, o, B, H, d, nq_main, nq_extra, nk_main, nk_extra, scale, lse)
    fin = None
    if PROFILE is not None and is_half(q.dtype):   # QK^T + PV: 4 B H Nq Nk d
        fin = PROFILE("attn_fwd", 4.0 * B * H * (nq_main + nq_extra) * (nk_main + nk_extra) * d, REGION[-1])
    L.check(lib.vfm_attn_fwd(C.byref(a), L.stream()), "vfm_attn_fwd")
    if fin is not None:
        fin()
    return o


def attn_bwd(q, k, v, o, lse, dout, dq, dk, dv, B, H, d, nq_main, nq_extra, nk_main, nk_extra, scale):
    lib = L.load()
    a = _attn_desc(q, k, v, o, B, H, d, nq_main, nq_extra, nk_main, nk_extra, scale, lse)
    a.dout, a.ld_do = L.ptr(dout), dout.stride(0)
    a.dq, a.dk, a.dv = L.ptr(dq), L.ptr(dk), L.ptr(dv)
    a.ld_dq, a.ld_dk, a.ld_dv = dq.stride(0), dk.stride(0), dv.stride(0)
    a.delta = L.ptr(_attn_bwd_delta(q, B, H, nq_main + nq_extra))
    if (q.dtype == torch.float32 and d == 64 and _split3_on() and q.shape[1] == H * d and k.shape[1] == H * d and v.shape[1] == H * d
            and dout.shape[1] == H * d and dq.dtype == torch.float32):
        # bf16x3 mode: the five products on split-bf16 operands (streamed operands arrive split; stationary rows, P and dS are split in the kernels)
        g3 = split3(dout, 1)
        full = _qkv_split3(q, k, v, H * d)
        if full is not None:       # q / k / v are the column thirds of one packed buffer: ONE split (kept from the forward) serves all three
            q3, k3, v3, lo = full[:, :H * d], full[:, H * d:], full[:, 2 * H * d:], 3 * H * d
        else:
            q3, k3, v3, lo = split3(q, 1), split3(k, 1), split3(v, 1), H * d
            assert q3.stride(0) == k3.stride(0) == v3.stride(0)
        L.check(lib.vfm_attn_bwd_x3(C.byref(a), L.ptr(q3), L.ptr(k3), L.ptr(v3), q3.stride(0), lo, L.ptr(g3), g3.stride(0), H * d, L.stream()),
                "vfm_attn_bwd_x3")
        return
    fin = None
    if PROFILE is not None and is_half(q.dtype):   # algorithmic: S, dP, dV, dK, dQ = five products = 10 B H Nq Nk d
        fin = PROFILE("attn_bwd", 10.0 * B * H * (nq_main + nq_extra) * (nk_main + nk_extra) * d, REGION[-1])
    L.check(lib.vfm_attn_bwd(C.byref(a), L.stream()), "vfm_attn_bwd")
    if fin is not None:
        fin()


def sam_relpos_table(rel_pos, S, out):
    lib = L.load()
    L.check(lib.vfm_sam_relpos_table(L.ptr(rel_pos), rel_pos.shape[0], rel_pos.shape[1], S, L.ptr(out), L.stream()),
            "vfm_sam_relpos_table")
    return out


def sam_attn_flash_fwd(qkv, bias, tbl_h, tbl_w, out, nimg, G, S, H, d, scale):
    """One-launch SAM attention forward (inference): qkv [M, 3*H*d] bf16 -> out [M, H*d] bf16."""
    lib = L.load()
    assert is_half(qkv.dtype) and is_half(out.dtype) and is_half(tbl_h.dtype) and tbl_h.is_contiguous()
    fin = None
    if PROFILE is not None:   # algorithmic FLOPs: q.k and p.v over the S*S keys of every window token (bias products not counted)
        nws = 1 if S == 32 else (G + S - 1) // S
        fin = PROFILE("attn_fwd", 4.0 * nimg * nws * nws * H * (S * S) ** 2 * d, REGION[-1])
    L.check(lib.vfm_sam_attn_flash_fwd(L.ptr(qkv), _ld(qkv), L.ptr(bias), L.ptr(tbl_h), L.ptr(tbl_w), L.ptr(out), _ld(out), nimg, G, S, H, d,
                                       float(scale), L.stream()), "vfm_sam_attn_flash_fwd")
    if fin is not None:
        fin()
    return out


def sam_attn_flash_stats(nimg, G, S, H, dev):
    """Buffers the training forward leaves for the backward: lse [rows] fp32, qext [rows, 2*SP] bf16 (rows = window-heads x 256 / 1024)."""
    nws = 1 if S == 32 else (G + S - 1) // S
    rows = nimg * nws * nws * H * (256 if S == 14 else 1024)
    return (torch.empty(rows, dtype=torch.float32, device=dev), torch.empty(rows, 2 * (16 if S == 14 else 32), dtype=L.half_dtype(), device=dev))


def sam_attn_flash_fwd_train(qkv, bias, tbl_h, tbl_w, out, lse, qext, nimg, G, S, H, d, scale):
    """sam_attn_flash_fwd that also writes the per-query log2-sum-exp and the bias columns of the query operand."""
    lib = L.load()
    assert is_half(qkv.dtype) and is_half(out.dtype) and is_half(tbl_h.dtype) and tbl_h.is_contiguous()
    assert lse.dtype == torch.float32 and is_half(qext.dtype) and qext.is_contiguous()
    L.check(lib.vfm_sam_attn_flash_fwd_train(L.ptr(qkv), _ld(qkv), L.ptr(bias), L.ptr(tbl_h), L.ptr(tbl_w), L.ptr(out), _ld(out), nimg, G, S, H, d,
                                             float(scale), L.ptr(lse), L.ptr(qext), L.stream()), "vfm_sam_attn_flash_fwd_train")
    return out


def sam_attn_flash_bwd(qkv, bias, tbl_h, tbl_w, out, dout, lse, qext, dqkv, nimg, G, S, H, d, scale):
    """d(out) [M, H*d] -> dqkv [M, 3*H*d] (bf16, every element written) in two launches (csrc/sam_flash_bwd.hip)."""
    lib = L.load()
    assert all(is_half(t.dtype) for t in (qkv, out, dout, dqkv, qext)) and _ld(out) == _ld(dout)
    dsum = torch.empty_like(lse)
    L.check(lib.vfm_sam_attn_flash_bwd(L.ptr(qkv), _ld(qkv), L.ptr(bias), L.ptr(tbl_h), L.ptr(tbl_w), L.ptr(out), L.ptr(dout), _ld(out), L.ptr(lse),
                                       L.ptr(qext), L.ptr(dsum), L.ptr(dqkv), _ld(dqkv), nimg, G, S, H, d, float(scale), L.stream()),
            "vfm_sam_attn_flash_bwd")
    return dqkv


def sam_attn_prep(qkv, bias, rh, rw, q_aug, k_aug, v_win, nimg, G, S, H, d, scale):
    lib = L.load()
    L.check(lib.vfm_sam_attn_prep(L.ptr(qkv), L.dt_of(qkv), _ld(qkv), L.ptr(bias), L.ptr(rh), L.ptr(rw), L.ptr(q_aug),
                                  L.ptr(k_aug), L.ptr(v_win), nimg, G, S, H, d, q_aug.shape[-1], v_win.shape[1], q_aug.shape[1],
                                  float(scale), L.stream()), "vfm_sam_attn_prep")


def sam_attn_bwd_prep(dao, qkv, bias, dow, dowT, vp, qsT, nimg, G, S, H, d, scale):
    """dO / V window-partitioned with the head dim zero-padded, dO^T and (scale q)^T: the small operands of the backward GEMMs."""
    lib = L.load()
    L.check(lib.vfm_sam_attn_bwd_prep(L.ptr(dao), _ld(dao), L.ptr(qkv), _ld(qkv), L.ptr(bias), L.dt_of(qkv), L.ptr(dow), L.ptr(dowT),
                                      L.ptr(vp), L.ptr(qsT), nimg, G, S, H, d, dow.shape[-1], dow.shape[1], float(scale), L.stream()),
            "vfm_sam_attn_bwd_prep")


def softmax_rows_batched(scores, out, n, rows_per_batch, valid_rows):
    lib = L.load()
    L.check(lib.vfm_softmax_rows_batched(L.ptr(scores), _ld(scores), L.ptr(out), L.dt_of(out), _ld(out), scores.shape[0], n, out.shape[1],
                                         rows_per_batch, valid_rows, L.stream()), "vfm_softmax_rows_batched")
    return out


def softmax_rows_bwd(p, dp, ds, n, rows_per_batch, valid_rows):
    lib = L.load()
    L.check(lib.vfm_softmax_rows_bwd(L.ptr(p), L.ptr(dp), _ld(dp), L.ptr(ds), L.dt_of(p), _ld(p), p.shape[0], n, p.shape[1], rows_per_batch,
                                     valid_rows, L.stream()), "vfm_softmax_rows_bwd")
    return ds


def sam_attn_bwd_merge(dqa, dkT, dvT, rh, rw, dqkv, nimg, G, S, H, d, scale):
    lib = L.load()
    L.check(lib.vfm_sam_attn_bwd_merge(L.ptr(dqa), L.ptr(dkT), L.ptr(dvT), L.dt_of(dqa), L.ptr(rh), L.ptr(rw), L.ptr(dqkv), _ld(dqkv), nimg,
                                       G, S, H, d, dkT.shape[1], dkT.shape[2], dqa.shape[-1], float(scale), L.stream()),
            "vfm_sam_attn_bwd_merge")
    return dqkv


def softmax_rows(scores2d, out2d, n):
    lib = L.load()
    L.check(lib.vfm_softmax_rows(L.ptr(scores2d), _ld(scores2d), L.ptr(out2d), L.dt_of(out2d), _ld(out2d), scores2d.shape[0], n,
                                 out2d.shape[1], L.stream()), "vfm_softmax_rows")
    return out2d


def sam_attn_merge(o_win, out, nimg, G, S, H, d):
    lib = L.load()
    L.check(lib.vfm_sam_attn_merge(L.ptr(o_win), L.dt_of(o_win), L.ptr(out), _ld(out), nimg, G, S, H, d, o_win.shape[1],
                                   L.stream()), "vfm_sam_attn_merge")
    return out


def patchify(img, out, box=None, patch=16):
    """img fp32 [B,3,H,W] (any strides with unit x-stride); box=(y0,y1,x0,x1) crop; out [B*nh*nw, >=3*P*P]."""
    lib = L.load()
    B, _, H, W = img.shape
    y0, y1, x0, x1 = box if box is not None else (0, H, 0, W)
    assert img.stride(3) == 1 and img.dtype == torch.float32
    L.check(lib.vfm_patchify(L.ptr(img), img.stride(0), img.stride(1), img.stride(2), y0, x0, y1 - y0, x1 - x0, patch,
                             L.ptr(out), L.dt_of(out), _ld(out), B, L.stream()), "vfm_patchify")
    return out


def assemble_tokens(patch_tok, cls, pos, x, B, np_, C_):
    lib = L.load()
    L.check(lib.vfm_assemble_tokens(L.ptr(patch_tok), L.ptr(cls), L.ptr(pos), L.ptr(x), B, np_, C_, L.stream()),
            "vfm_assemble_tokens")
    return x


def resize_bilinear(inp, in_nchw, B, Hi, Wi, Cc, out, out_mode, virt, window=None, in_ld=None, out_ld=None):
    """virt=(Hv,Wv) virtual output size; window=(y0,x0,hc,wc) (default: whole). out_mode 0 NHWC, 1 NCHW, 2 blocked."""
    lib = L.load()
    Hv, Wv = virt
    y0, x0, hc, wc = window if window is not None else (0, 0, Hv, Wv)
    in_ld = Cc if in_ld is None else in_ld
    out_ld = Cc if out_ld is None else out_ld
    L.check(lib.vfm_resize_bilinear(L.ptr(inp), L.dt_of(inp), int(in_nchw), B, Hi, Wi, Cc, in_ld, L.ptr(out), L.dt_of(out),
                                    out_mode, out_ld, Hv, Wv, y0, x0, hc, wc, L.stream()), "vfm_resize_bilinear")
    return out


def resize_bicubic(inp, Hi, Wi, Cc, out, Ho, Wo, scale_y, scale_x):
    lib = L.load()
    L.check(lib.vfm_resize_bicubic(L.ptr(inp), Hi, Wi, Cc, L.ptr(out), Ho, Wo, float(scale_y), float(scale_x), L.stream()),
            "vfm_resize_bicubic")
    return out


def label_resize(lab, out, virt, window=None):
    lib = L.load()
    B, Hi, Wi = lab.shape
    Hv, Wv = virt
    y0, x0, hc, wc = window if window is not None else (0, 0, Hv, Wv)
    assert lab.dtype == torch.int64 and lab.is_contiguous() and out.is_contiguous()
    L.check(lib.vfm_label_resize(L.ptr(lab), B, Hi, Wi, L.ptr(out), Hv, Wv, y0, x0, hc, wc, L.stream()), "vfm_label_resize")
    return out


def unblock(x, y, B, H, W, Cc, levels, inverse=False):
    lib = L.load()
    L.check(lib.vfm_unblock(L.ptr(x), L.ptr(y), B, H, W, Cc, levels, int(inverse), L.stream()), "vfm_unblock")
    return y


def upsample_ce(logits_low, label, ignore_index=255, need_grad=True):
    """logits_low fp32 [B,h,w,C] NHWC; label int64 [B,H,W]. Returns (loss[1], counts int32[2], dlogits or None)."""
    lib = L.load()
    B, h, w, Cc = logits_low.shape
    _, H, W = label.shape
    dev = logits_low.device
    parts = torch.empty(B * h * w, dtype=torch.float32, device=dev)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    dl = torch.empty_like(logits_low) if need_grad else None
    assert logits_low.is_contiguous() and label.is_contiguous() and label.dtype == torch.int64
    L.check(lib.vfm_upsample_ce(L.ptr(logits_low), L.ptr(label), B, h, w, Cc, H, W, ignore_index, L.ptr(parts), L.ptr(counts),
                                L.ptr(dl), L.stream()), "vfm_upsample_ce")
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    L.check(lib.vfm_reduce_sum(L.ptr(parts), parts.numel(), 1.0 / (B * H * W), L.ptr(loss), L.stream()), "vfm_reduce_sum")
    return loss, counts, dl


_ce_counts = {}


def upsample_ce_loss_acc(logits_low, label, ignore_index=255, need_grad=True):
    """The training form of upsample_ce: returns (loss[1], acc[1], dlogits) with the loss / accuracy finished by ONE launch and
    a persistent, self-resetting counter pair per device (zeroed once)."""
    lib = L.load()
    B, h, w, Cc = logits_low.shape
    _, H, W = label.shape
    dev = logits_low.device
    counts = _ce_counts.get(dev)
    if counts is None:
        counts = _ce_counts[dev] = torch.zeros(2, dtype=torch.int32, device=dev)
    parts = torch.empty(B * h * w, dtype=torch.float32, device=dev)
    dl = torch.empty_like(logits_low) if need_grad else None
    assert logits_low.is_contiguous() and label.is_contiguous() and label.dtype == torch.int64
    L.check(lib.vfm_upsample_ce(L.ptr(logits_low), L.ptr(label), B, h, w, Cc, H, W, ignore_index, L.ptr(parts), L.ptr(counts),
                                L.ptr(dl), L.stream()), "vfm_upsample_ce")
    out = torch.empty(2, dtype=torch.float32, device=dev)
    eps = float(torch.finfo(torch.float32).eps)
    L.check(lib.vfm_ce_finish(L.ptr(parts), parts.numel(), 1.0 / (B * H * W), L.ptr(counts), eps, L.ptr(out), L.ptr(out[1:]), L.stream()),
            "vfm_ce_finish")
    return out[0:1], out[1:2], dl


def preprocess_u8(img_u8, out, mean, std, bgr_to_rgb, pad_val=0.0):
    """img_u8 uint8 [3,H,W] (cuda) -> out fp32 [3,Hp,Wp]: channel swap, normalise, pad (mmseg SegDataPreProcessor)."""
    lib = L.load()
    assert img_u8.dtype == torch.uint8 and img_u8.is_contiguous() and out.is_contiguous() and out.dtype == torch.float32
    _, H, W = img_u8.shape
    _, Hp, Wp = out.shape
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    L.check(lib.vfm_preprocess_u8(L.ptr(img_u8), H, W, L.ptr(out), Hp, Wp, m3, s3, int(bool(bgr_to_rgb)), float(pad_val), L.stream()),
            "vfm_preprocess_u8")
    return out


def conf_gate_count(logits_nchw, window, thr, count):
    lib = L.load()
    B, Cc, H, W = logits_nchw.shape
    y0, x0, hc, wc = window
    L.check(lib.vfm_conf_gate(L.ptr(logits_nchw), B, Cc, H, W, y0, x0, hc, wc, float(thr), L.ptr(count), L.stream()),
            "vfm_conf_gate")
    return count


def slide_accumulate(crop, crop_nchw, B, h, w, Cc, preds, count, window):
    lib = L.load()
    _, _, H, W = preds.shape
    y0, x0, hc, wc = window
    L.check(lib.vfm_slide_accumulate(L.ptr(crop), int(crop_nchw), B, h, w, Cc, L.ptr(preds), L.ptr(count), H, W, y0, x0, hc, wc,
                                     L.stream()), "vfm_slide_accumulate")


def slide_gather(wins, preds):
    """wins: list of (logits, nchw, window=(y0, x0, hc, wc)) in accumulation order; logits NHWC [B,h,w,C] or NCHW [B,C,h,w] fp32 contiguous.
    preds [B,C,H,W] = mean over the covering windows of their bilinear samples (vfm_slide_gather); returns False when the table does not fit
    (more than 16 windows / 32 channels): the caller then takes the per-window accumulate path."""
    B, Cc, H, W = preds.shape
    if len(wins) > 16 or Cc > 32 or len(wins) == 0 or os.environ.get("VFMSEG_SLIDE_GATHER", "1") == "0":   # (env: A/B of the two paths)
        return False
    arr = (L.SlideWin * len(wins))()
    for j, (t, nchw, (y0, x0, hc, wc)) in enumerate(wins):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] == B
        h, w = (t.shape[2], t.shape[3]) if nchw else (t.shape[1], t.shape[2])
        arr[j] = L.SlideWin(L.ptr(t), int(bool(nchw)), h, w, y0, x0, hc, wc)
    L.check(L.load().vfm_slide_gather(arr, len(wins), B, Cc, L.ptr(preds), H, W, L.stream()), "vfm_slide_gather")
    return True


def conf_gate_windows(logits_nchw, windows, thr, counts):
    """counts[j] += number of pixels of window j = (y0, x0, hc, wc) whose max softmax exceeds thr - all windows in one pass (<= 16)."""
    B, Cc, H, W = logits_nchw.shape
    flat = (L.ci * (4 * len(windows)))(*[int(v) for wdw in windows for v in wdw])
    L.check(L.load().vfm_conf_gate_windows(L.ptr(logits_nchw), B, Cc, H, W, flat, len(windows), float(thr), L.ptr(counts), L.stream()),
            "vfm_conf_gate_windows")
    return counts


def slide_finalize(preds, count, argmax=None):
    lib = L.load()
    B, Cc, H, W = preds.shape
    L.check(lib.vfm_slide_finalize(L.ptr(preds), L.ptr(count), L.ptr(argmax), B, Cc, H, W, L.stream()), "vfm_slide_finalize")
    return preds


def confusion_hist(pred_u8, label, hist, num_classes, ignore_index=255):
    """hist int64 [(nc+1)*nc] += confusion counts of (label, pred) over the pixels with label != ignore_index."""
    lib = L.load()
    assert pred_u8.dtype == torch.uint8 and pred_u8.is_contiguous() and label.is_contiguous() and pred_u8.numel() == label.numel()
    assert hist.dtype == torch.int64 and hist.numel() == (num_classes + 1) * num_classes
    L.check(lib.vfm_confusion_hist(L.ptr(pred_u8), L.ptr(label), L.dt_of(label), pred_u8.numel(), int(num_classes), int(ignore_index),
                                   L.ptr(hist), L.stream()), "vfm_confusion_hist")
    return hist


def adamw(p, g, m, v, seg_start, seg_lr_mult, seg_wd, lr, betas, eps, step, grad_scale=1.0, zero_grad=False, vec4=False, skip=None, amp_state=None):
    """skip: int32 device tensor [1]; non-zero = leave parameters and moments alone (loss-scaled training: the step's gradients overflowed).
    amp_state: fp32 device tensor [4] {loss scale, growth tracker, steps taken, steps skipped}: un-scale and bias corrections from it."""
    lib = L.load()
    assert skip is None or (skip.dtype == torch.int32 and skip.numel() == 1)
    assert amp_state is None or (amp_state.dtype == torch.float32 and amp_state.numel() == 4 and amp_state.is_contiguous())
    L.check(lib.vfm_adamw_guarded(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), L.ptr(seg_start), L.ptr(seg_lr_mult),
                                  L.ptr(seg_wd), seg_start.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps), max(int(step), 1),
                                  float(grad_scale), int(bool(zero_grad)), int(bool(vec4)), L.ptr(skip), L.ptr(amp_state), L.stream()), "vfm_adamw")


def amp_update(flag, amp_state, growth, backoff, interval, dynamic):
    """GradScaler.update() + the optimiser's step count on the device (vfm_amp_update)."""
    L.check(L.load().vfm_amp_update(L.ptr(flag), L.ptr(amp_state), float(growth), float(backoff), int(interval), int(bool(dynamic)), L.stream()),
            "vfm_amp_update")
