"""Autograd ops of the decoder heads, each a thin torch.autograd.Function over the C-ABI kernels.

Conventions: feature maps are token-major matrices [rows, C]; GEMM operands use the compute dtype (bf16 | f32),
normalisation inputs / residual streams / losses are fp32.  Every Function returns gradients in the dtype of the
corresponding forward input, so autograd never inserts a cast of its own.
"""
import torch

from .precision import is_half
from . import ops
from .precision import compute_dtype

_seed_state = {"seed": 0x5EED, "offset": 0}


def manual_seed(seed):
    _seed_state["seed"], _seed_state["offset"] = int(seed), 0


def _next_rng(n):
    off = _seed_state["offset"]
    _seed_state["offset"] += int(n)
    return _seed_state["seed"], off


def draw_seed(seed, n):
    """(seed, base offset) for a backbone call's LoRA-dropout masks.  seed=None (the training path) reserves n counters of
    the advancing generator the heads use, so every call - every train step, every DP rank after manual_seed(rank-specific)
    - samples fresh masks like peft's nn.Dropout (lora_backbone.py:16-23); an explicit int pins the masks (tests)."""
    if seed is None:
        return _next_rng(n)
    return int(seed), 0


def _pad64(n):
    return (n + 63) // 64 * 64


# ------------------------------------------------------------------------------------------------ weight-gradient side stream
# The decoder heads' weight and bias gradients are ~30 small launch-latency-bound GEMMs (split-K over the tokens, 64-128 blocks, ~19 us
# each for a microsecond of arithmetic) plus their slab reduces and column sums: 0.9 ms of a 15.7-ms step, and nothing in the backward
# pass waits for them - they only feed the optimiser.  They are issued on a SIDE stream behind an event, next to the input-gradient
# chain that is the critical path (the chip is mostly idle under those small kernels); the main stream joins it before anything reads the
# flat gradient buffer (DP bucket all-reduce, AMP overflow check, AdamW): join_wgrad_stream().  Same kernels, same arguments: results
# are unchanged.  VFMSEG_WGRAD_STREAM=0 keeps everything on one stream.
# (Tried and dropped, round 4: the LR decode head's forward - hence, by autograd's stream rule, its backward - on a second stream beside the
# detail head's backward, which does not depend on it: neutral, 15.17 vs 15.15 ms/step, profiles/r04_ab_step_head_stream.log.)
_WG = {"stream": None, "used": False}


def wgrad_stream():
    import os
    if os.environ.get("VFMSEG_WGRAD_STREAM", "1") == "0":
        return None
    if _WG["stream"] is None:
        _WG["stream"] = torch.cuda.Stream()
    return _WG["stream"]


def join_wgrad_stream():
    """The current stream waits for the weight-gradient work issued so far (no-op when there is none)."""
    if _WG["used"]:
        torch.cuda.current_stream().wait_stream(_WG["stream"])
        _WG["used"] = False


# ------------------------------------------------------------------------------------------------ weight layouts
# Each trainable weight is packed into the GEMM's B operand [N, K] (K contiguous) and its gradient un-packed.
#   linear / conv1x1 : param [N, K(,1,1)]                      -> B[n, k]
#   conv2x2s2        : param [Cout, Cin, 2, 2]                  -> B[co, (ky*2+kx)*Cin + ci]   (im2col order of blocked maps)
#   convT2x2         : param [Cin, Cout, 2, 2]                  -> B[(ky*2+kx)*Cout + co, ci] (pixel-shuffle-free ConvTranspose)
def _layout_dims(w, layout):
    if layout in ("linear", "conv1x1"):
        return w.shape[0], w[0].numel()
    if layout == "conv2x2s2":
        return w.shape[0], 4 * w.shape[1]
    if layout == "convT2x2":
        return 4 * w.shape[1], w.shape[0]
    raise ValueError(layout)


def _pack_job(w, layout, dst):
    """(src, dst, shape, src strides, dst strides) of the strided copy that packs `w` into dst [N, >=K] (any dtype)."""
    ld = dst.stride(0)
    if layout in ("linear", "conv1x1"):
        n, k = _layout_dims(w, layout)
        return (w, dst, (n, k), (k, 1), (ld, 1))
    if layout == "conv2x2s2":
        co, ci = w.shape[0], w.shape[1]
        # index space (co, kk, ci): src w[co, ci, kk] ; dst [co, kk*ci_n + ci]
        return (w, dst, (co, 4, ci), (ci * 4, 1, 4), (ld, ci, 1))
    if layout == "convT2x2":
        ci, co = w.shape[0], w.shape[1]
        # index space (kk, co, ci): src w[ci, co, kk] ; dst [(kk*co_n + co), ci]
        return (w, dst, (4, co, ci), (1, 4, co * 4), (co * ld, ld, 1))
    raise ValueError(layout)


def _pack(w, layout, dst):
    """dst [N, >=K] (any dtype) <- param"""
    src, dst, shape, sstr, dstr = _pack_job(w, layout, dst)
    ops.strided_copy(src, dst, shape, sstr, dstr)
    return dst


class _PackCache:
    """Packed GEMM operands of the trainable decoder weights, rebuilt ONCE per parameter epoch by ONE batched launch.

    An entry is keyed by the weights' storage (data pointers + shapes), the layouts, the padded K and the compute dtype.  It is
    current while optim.PARAM_EPOCH (bumped by the fused AdamW step, which rewrites parameters behind torch's version
    counters) and every weight's `_version` (load_state_dict, in-place edits) are unchanged.  The first stale lookup after an
    optimiser step re-packs every live entry with vfm_strided_copy_batch; entries whose parameters died or moved are dropped."""

    def __init__(self):
        self.entries = {}
        self.table = None
        self.table_sig = None

    @staticmethod
    def _stamp(weights):
        from .optim import PARAM_EPOCH
        return (PARAM_EPOCH[0],) + tuple(w._version for w in weights)

    def get(self, weights, layouts, dims, Kp, K, cd):
        import weakref
        key = (tuple((w.data_ptr(), tuple(w.shape)) for w in weights), tuple(layouts), Kp, cd)
        e = self.entries.get(key)
        if e is not None and any(r() is None for r in e["refs"]):
            e = None   # the storage address was recycled by other tensors
        stamp = self._stamp(weights)
        if e is None:
            N = sum(d[0] for d in dims)
            dev = weights[0].device
            wp = torch.zeros(N, Kp, dtype=cd, device=dev) if Kp > K else torch.empty(N, Kp, dtype=cd, device=dev)
            jobs, r0 = [], 0
            for w, l, d in zip(weights, layouts, dims):
                jobs.append(_pack_job(w.detach(), l, wp[r0:r0 + d[0]]))
                r0 += d[0]
            for j in jobs:
                ops.strided_copy(*j)
            self.entries[key] = dict(wp=wp, refs=[weakref.ref(w) for w in weights], layouts=list(layouts), dims=list(dims), stamp=stamp)
            self.table = None
            return wp
        if e["stamp"] != stamp:
            self.refresh_all()
            e["stamp"] = self._stamp(weights)
        return e["wp"]

    def refresh_all(self):
        live = {}
        for key, e in self.entries.items():
            ws = [r() for r in e["refs"]]
            if any(w is None for w in ws) or tuple((w.data_ptr(), tuple(w.shape)) for w in ws) != key[0]:
                continue
            live[key] = e
        if len(live) != len(self.entries):
            self.entries, self.table = live, None
        sig = tuple(self.entries.keys())
        if self.table is None or self.table_sig != sig:
            jobs = []
            for key, e in self.entries.items():
                r0 = 0
                for r, l, d in zip(e["refs"], e["layouts"], e["dims"]):
                    w = r().detach()
                    src = w if w.dtype == torch.float32 and w.is_contiguous() else None
                    if src is None:
                        jobs = None
                        break
                    jobs.append(_pack_job(src, l, e["wp"][r0:r0 + d[0]]))
                    r0 += d[0]
                if jobs is None:
                    break
            self.table = ops.CopyBatch(jobs) if jobs else False
            self.table_sig = sig
        if self.table:
            self.table.run()
        else:   # a non-fp32 / non-contiguous parameter: pack one by one
            for key, e in self.entries.items():
                r0 = 0
                for r, l, d in zip(e["refs"], e["layouts"], e["dims"]):
                    _pack(r().detach(), l, e["wp"][r0:r0 + d[0]])
                    r0 += d[0]
        for key, e in self.entries.items():
            e["stamp"] = self._stamp([r() for r in e["refs"]])
            # the re-pack rewrote wp behind torch's version counter: a split-bf16 image cached on it (ops.split3) is stale
            if hasattr(e["wp"], "_vfm_split3"):
                del e["wp"]._vfm_split3


class _PackCaches:
    """One cache per library (bf16 / fp16 twin): a refresh re-packs every live entry of ITS cache with the active library's kernels,
    and entries of the other 16-bit type (models of the other precision mode that are still alive) must not be touched by it."""

    def __init__(self):
        self.by_lib = {False: _PackCache(), True: _PackCache()}

    def get(self, weights, layouts, dims, Kp, K, cd):
        return self.by_lib[cd == torch.float16].get(weights, layouts, dims, Kp, K, cd)


PACKS = _PackCaches()


def direct_grad_target(p):
    """The optimiser (optim.FusedAdamW) pre-points every trainable parameter's .grad into one flat fp32 buffer that
    is zeroed each step.  Backward kernels then accumulate straight into it (and hand autograd `None`), which removes
    one allocation and one add-kernel per parameter per step."""
    g = p.grad
    if g is not None and g.dtype == torch.float32 and g.is_contiguous() and getattr(p, "_vfm_direct_grad", False):
        return g
    return None


def _unpack_grad(g2d, layout, w):
    """fp32 grad in param layout from the [N, >=K] GEMM-layout gradient (accumulated in place when the optimiser
    exposes a flat gradient buffer; returns None in that case)."""
    tgt = direct_grad_target(w)
    acc = tgt is not None
    out = tgt if acc else torch.empty_like(w, dtype=torch.float32)
    ld = g2d.stride(0)
    if layout in ("linear", "conv1x1"):
        n, k = _layout_dims(w, layout)
        ops.strided_copy(g2d, out, (n, k), (ld, 1), (k, 1), accumulate=acc)
    elif layout == "conv2x2s2":
        co, ci = w.shape[0], w.shape[1]
        ops.strided_copy(g2d, out, (co, 4, ci), (ld, ci, 1), (ci * 4, 1, 4), accumulate=acc)
    elif layout == "convT2x2":
        ci, co = w.shape[0], w.shape[1]
        ops.strided_copy(g2d, out, (4, co, ci), (co * ld, ld, 1), (1, 4, co * 4), accumulate=acc)
    return None if acc else out


def _splitk_wgrad(dyt, x, out, kchunks, m_rows):
    """out[N,Kp] = dyt[N,Mp] @ x[M,Kp] (x consumed in place as the transposed-B operand) with the token reduction
    split into `kchunks` batched slices that are then summed in a fixed order (deterministic split-K)."""
    n, mp = dyt.shape
    kp = x.shape[1]
    if kchunks <= 1:
        return ops.gemm(dyt, x, out, trans_b=True, kb_rows=m_rows)
    slabs = torch.empty(kchunks, n, kp, dtype=torch.float32, device=out.device)
    ops.gemm_splitk_bt(dyt, x, slabs, kchunks)
    ops.colsum(slabs.view(kchunks, n * kp), out.view(n * kp))
    return out


class LinearFn(torch.autograd.Function):
    """y = [dropout](act(x @ W^T + b)) [+ residual]   with W = concat_N(pack(w_i)).

    x: [M, Kp] compute dtype (columns beyond the weights' K must be zero).  out_dtype: torch dtype of y.
    act: None | 'gelu' | 'relu'.  residual: fp32/any [M, N] added after the (dropped-out) linear part.
    """

    @staticmethod
    def forward(ctx, x, residual, bias, opts, *weights):
        layouts, act, out_dtype, drop_p, bias_tile = opts["layouts"], opts.get("act"), opts["out_dtype"], opts.get("drop_p", 0.0), opts.get("bias_tile", 1)
        cd = x.dtype
        M, Kp = x.shape
        dims = [_layout_dims(w, l) for w, l in zip(weights, layouts)]
        N = sum(d[0] for d in dims)
        K = dims[0][1]
        assert all(d[1] == K for d in dims) and K <= Kp
        wp = PACKS.get(weights, layouts, dims, Kp, K, cd)   # packed once per parameter epoch, all decoder weights in one launch
        y = torch.empty(M, N, dtype=out_dtype, device=x.device)
        need_grad = any(ctx.needs_input_grad)
        pre = None
        ep, aux = ops.EP_NONE, None
        if act == "gelu":
            ep = ops.EP_GELU
            pre = torch.empty(M, N, dtype=cd, device=x.device) if need_grad else None
        elif act == "relu":
            ep = ops.EP_RELU
        mask = None
        if drop_p > 0:
            assert act is None
            mask = torch.empty(M, N, dtype=cd, device=x.device)
            seed, off = _next_rng(M * N)
            ops.dropout_mask(mask, drop_p, seed, off)
            ep, aux = ops.EP_MUL, mask
        bias_full = None
        if bias is not None:
            bias_full = bias.detach()
        ops.gemm(x, wp, y, bias=bias_full, bias_mod=(N // bias_tile if bias_tile > 1 else 0), residual=residual, ep_mode=ep,
                 aux=aux, c2=pre)
        ctx.save_for_backward(x, wp, pre if pre is not None else (y if act == "relu" else None), mask, *weights)
        ctx.opts, ctx.dims, ctx.has_res, ctx.has_bias = opts, dims, residual is not None, bias is not None
        ctx.bias_param = bias
        ctx.res_dtype = residual.dtype if residual is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wp, pre, mask, *weights = ctx.saved_tensors
        opts, dims = ctx.opts, ctx.dims
        layouts, act, bias_tile = opts["layouts"], opts.get("act"), opts.get("bias_tile", 1)
        cd = x.dtype
        M, Kp = x.shape
        N = wp.shape[0]
        dy = dy.contiguous()
        d_res = None
        if ctx.has_res and ctx.needs_input_grad[1]:
            d_res = dy if dy.dtype == ctx.res_dtype else _cast_new(dy, ctx.res_dtype)
        # effective gradient of the linear part, in the compute dtype, zero-padded to a multiple of 64 columns (bf16)
        npad = _pad64(N) if is_half(cd) else N
        if mask is None and act is None and dy.dtype == cd and npad == N and dy.data_ptr() % 16 == 0:
            g = gv = dy          # already the GEMM operand: no copy
        else:
            g = torch.zeros(M, npad, dtype=cd, device=dy.device) if npad > N else torch.empty(M, N, dtype=cd, device=dy.device)
            gv = g[:, :N]
        if g is dy:
            pass
        elif mask is not None:
            ops.mul_mask(dy, mask, gv)
        elif act == "gelu":
            _mul_act_grad(dy, pre, gv, "gelu")
        elif act == "relu":
            _mul_act_grad(dy, pre, gv, "relu")
        else:
            ops.cast(dy, gv)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, Kp, dtype=cd, device=dy.device)
            if is_half(cd):
                ops.gemm(g, wp, dx, trans_b=True, kb_rows=N)   # dX = g @ W: the packed weight [N, Kp] is the [K, N'] operand
            else:
                ops.gemm(gv, wp, dx, trans_b=True)
        # bias / weight gradients: off the critical path, on the side stream when every result lands in the flat gradient buffer
        side = None
        if is_half(cd) and dy.is_cuda and all(direct_grad_target(p_) is not None for p_ in
                                              ([ctx.bias_param] if (ctx.has_bias and ctx.needs_input_grad[2]) else []) +
                                              [w for i, w in enumerate(weights) if ctx.needs_input_grad[4 + i]]):
            side = wgrad_stream()
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())      # g (and x) are complete
            g.record_stream(side), x.record_stream(side)
            if not _WG["used"]:   # whoever runs this backward pass (a test, a tool) reads gradients behind it: join at its end
                torch.autograd.Variable._execution_engine.queue_callback(join_wgrad_stream)
            _WG["used"] = True
            with torch.cuda.stream(side):     # (ops.workspace hands every stream scratch of its own)
                dbias, dws = LinearFn._param_grads(ctx, g, gv, x, weights, dims, layouts, bias_tile, cd, M, N, Kp, npad)
        else:
            dbias, dws = LinearFn._param_grads(ctx, g, gv, x, weights, dims, layouts, bias_tile, cd, M, N, Kp, npad)
        return (dx, d_res, dbias, None) + tuple(dws)


def _param_grads_impl(ctx, g, gv, x, weights, dims, layouts, bias_tile, cd, M, N, Kp, npad):
        """(dbias, [dW...]) of one linear: column sums of g and g^T x; None where the result was accumulated into the flat buffer."""
        dy = g
        dbias = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            btgt = direct_grad_target(ctx.bias_param) if ctx.bias_param is not None else None
            if bias_tile > 1:
                full = torch.empty(N, dtype=torch.float32, device=dy.device)
                ops.colsum(gv, full)
                if btgt is not None:
                    ops.colsum(full.view(bias_tile, N // bias_tile), btgt.view(-1), accumulate=True)
                else:
                    dbias = torch.empty(N // bias_tile, dtype=torch.float32, device=dy.device)
                    ops.colsum(full.view(bias_tile, N // bias_tile), dbias)
            elif btgt is not None:
                ops.colsum(gv, btgt.view(-1), accumulate=True)
            else:
                dbias = torch.empty(N, dtype=torch.float32, device=dy.device)
                ops.colsum(gv, dbias)
        dws = [None] * len(weights)
        if any(ctx.needs_input_grad[4:]):
            gw = torch.empty(npad if is_half(cd) else N, Kp, dtype=torch.float32, device=dy.device)
            if is_half(cd):
                mp = _pad64(M)
                tiles = ((N + 127) // 128) * ((Kp + 127) // 128)
                kch = 1
                while tiles * kch < 256 and (mp // (kch * 2)) % 64 == 0 and mp // (kch * 2) >= 256:
                    kch *= 2
                # dW = g^T x with both operands token-major (TN form of vfm_gemm): no transposed copy of the gradient
                if kch > 1:
                    slabs = torch.empty(kch, npad, Kp, dtype=torch.float32, device=dy.device)
                    ops.gemm_splitk_tn(g, x, slabs, kch)
                    tgt = direct_grad_target(weights[0]) if len(weights) == 1 and ctx.needs_input_grad[4] else None
                    if tgt is not None and layouts[0] in ("linear", "conv1x1") and Kp == dims[0][1]:
                        # the split-K combine adds straight into the parameter's slot of the flat gradient buffer
                        ops.slab_reduce(slabs, N, tgt, Kp, 1, accumulate=True)
                        return dbias, dws
                    ops.colsum(slabs.view(kch, npad * Kp), gw.view(npad * Kp))
                else:
                    ops.gemm_splitk_tn(g, x, gw.view(1, npad, Kp), 1)
            else:
                ops.gemm(gv, x, gw, trans_a=True, trans_b=True)
            r0 = 0
            for i, (w, l, d) in enumerate(zip(weights, layouts, dims)):
                if ctx.needs_input_grad[4 + i]:
                    dws[i] = _unpack_grad(gw[r0:r0 + d[0]], l, w)
                r0 += d[0]
        return dbias, dws

LinearFn._param_grads = staticmethod(_param_grads_impl)


def _cast_new(t, dtype):
    out = torch.empty(t.shape, dtype=dtype, device=t.device)
    ops.cast(t.view(-1, t.shape[-1]), out.view(-1, t.shape[-1]))
    return out


def _mul_act_grad(dy, pre, out, kind):
    """out = dy * act'(pre)  (relu: `pre` may be the post-activation, the sign test is the same)."""
    M, N = dy.shape
    if kind == "gelu":
        # reuse the GEMM-free path: gelu'(pre) via the norm-free elementwise kernel family
        ops.act_grad_mul(dy, pre, out, ops.ACT_GELU)
    else:
        ops.act_grad_mul(dy, pre, out, ops.ACT_RELU)


def linear(x, weights, layouts, bias=None, residual=None, act=None, out_dtype=None, drop_p=0.0, bias_tile=1):
    if not isinstance(weights, (list, tuple)):
        weights, layouts = [weights], [layouts]
    opts = dict(layouts=list(layouts), act=act, out_dtype=out_dtype or x.dtype, drop_p=drop_p, bias_tile=bias_tile)
    return LinearFn.apply(x, residual, bias, opts, *weights)


# ------------------------------------------------------------------------------------------------ norms
def _norm_grad_slots(w, b, C, device):
    """(dw, db, returned dw, returned db) for a norm backward: the kernels ACCUMULATE into dw / db, so with the optimiser's flat
    gradient buffer they add straight into the parameters' slots (autograd gets None: no zero-fill, no add kernel per tensor)."""
    tw, tb = direct_grad_target(w), direct_grad_target(b)
    dw = tw.view(-1) if tw is not None else torch.zeros(C, dtype=torch.float32, device=device)
    db = tb.view(-1) if tb is not None else torch.zeros(C, dtype=torch.float32, device=device)
    return dw, db, (None if tw is not None else dw), (None if tb is not None else db)


class GroupNormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, B, P, groups, eps, act, out_dtype):
        x = x.contiguous()
        C = x.shape[1]
        y = torch.empty(B * P, C, dtype=out_dtype, device=x.device)
        stats = torch.empty(B, groups, 2, dtype=torch.float32, device=x.device)
        ops.groupnorm_fwd(x, w.detach(), b.detach(), eps, groups, act, y, stats, B, P)
        ctx.save_for_backward(x, w, b, stats)
        ctx.cfg = (B, P, groups, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, stats = ctx.saved_tensors
        B, P, groups, act = ctx.cfg
        C = x.shape[1]
        dx = torch.empty_like(x)
        dw, db, ret_w, ret_b = _norm_grad_slots(w, b, C, x.device)
        ops.groupnorm_bwd(dy.contiguous(), x, w.detach(), b.detach(), stats, groups, act, dx, dw, db, B, P)
        return dx, ret_w, ret_b, None, None, None, None, None, None


def group_norm_act(x, w, b, B, P, groups=32, eps=1e-5, act=ops.ACT_NONE, out_dtype=None):
    return GroupNormActFn.apply(x, w, b, B, P, groups, eps, act, out_dtype or compute_dtype())


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, out_dtype):
        x = x.contiguous()
        M, C = x.shape
        y = torch.empty(M, C, dtype=out_dtype, device=x.device)
        stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
        ops.layernorm_fwd(x, w.detach(), b.detach(), eps, y, stats)
        ctx.save_for_backward(x, w, b, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, stats = ctx.saved_tensors
        C = x.shape[1]
        dx = torch.empty_like(x)
        dw, db, ret_w, ret_b = _norm_grad_slots(w, b, C, x.device)
        ops.layernorm_bwd(dy.contiguous(), x, w.detach(), stats, dx, False, dw, db)
        return dx, ret_w, ret_b, None, None


def layer_norm(x, w, b, eps=1e-5, out_dtype=None):
    return LayerNormFn.apply(x, w, b, eps, out_dtype or compute_dtype())


class BatchNormActFn(torch.autograd.Function):
    """nn.SyncBatchNorm (train mode) + fused activation on [rows, C]; `sync` = callable all-reducing a fp32 tensor in
    place over the data-parallel group (None on a single GPU), `world_rows` = global row count."""

    @staticmethod
    def forward(ctx, x, w, b, running_mean, running_var, momentum, eps, act, out_dtype, sync, world):
        x = x.contiguous()
        rows, C = x.shape
        sums = torch.empty(2, C, dtype=torch.float32, device=x.device)
        ops.bn_moments(x, sums)
        total = rows * world
        if sync is not None:
            sync(sums)
        mv = torch.empty(2, C, dtype=torch.float32, device=x.device)
        ops.bn_finalize(sums, total, mv, running_mean, running_var, momentum)
        y = torch.empty(rows, C, dtype=out_dtype, device=x.device)
        ops.bn_apply(x, mv, w.detach(), b.detach(), eps, act, y)
        ctx.save_for_backward(x, w, b, mv)
        ctx.cfg = (eps, act, sync, total)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, mv = ctx.saved_tensors
        eps, act, sync, total = ctx.cfg
        rows, C = x.shape
        dy = dy.contiguous()
        sd = torch.empty(2, C, dtype=torch.float32, device=x.device)
        ops.bn_bwd_reduce(dy, x, mv, w.detach(), b.detach(), eps, act, sd)
        # local parameter gradients (DDP averages them later): straight into the flat gradient slots when there are any
        tw, tb = direct_grad_target(w), direct_grad_target(b)
        if tw is not None and tb is not None:
            ops.axpby(sd[1], 1.0, tw.view(-1), 1.0)
            ops.axpby(sd[0], 1.0, tb.view(-1), 1.0)
            dw = db = None
        else:
            db, dw = sd[0].clone(), sd[1].clone()
        if sync is not None:
            sync(sd)
        dx = torch.empty_like(x)
        ops.bn_bwd_apply(dy, x, mv, w.detach(), b.detach(), eps, act, sd, total, dx)
        return dx, dw, db, None, None, None, None, None, None, None, None


def batch_norm_act_train(x, w, b, running_mean, running_var, momentum=0.1, eps=1e-5, act=ops.ACT_NONE, out_dtype=None,
                         sync=None, world=1):
    return BatchNormActFn.apply(x, w, b, running_mean, running_var, momentum, eps, act, out_dtype or compute_dtype(), sync, world)


def batch_norm_act_eval(x, w, b, running_mean, running_var, eps=1e-5, act=ops.ACT_NONE, out_dtype=None):
    mv = torch.stack([running_mean.detach().float(), running_var.detach().float()]).contiguous()
    y = torch.empty(x.shape, dtype=out_dtype or compute_dtype(), device=x.device)
    ops.bn_apply(x.contiguous(), mv, w.detach(), b.detach(), eps, act, y)
    return y


# ------------------------------------------------------------------------------------------------ attention
class SelfAttnFn(torch.autograd.Function):
    """qkv [B*N, 3*H*d] packed (q | k | v) -> o [B*N, H*d]"""

    @staticmethod
    def forward(ctx, qkv, B, N, H, d):
        inner = H * d
        o = torch.empty(B * N, inner, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
        ops.attn_fwd(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], o, lse, B, H, d, N, 0, N, 0, d ** -0.5,
                     keep_split=ctx.needs_input_grad[0])
        ctx.save_for_backward(qkv, o, lse)
        ctx.cfg = (B, N, H, d)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        B, N, H, d = ctx.cfg
        inner = H * d
        dqkv = torch.empty_like(qkv)
        ops.attn_bwd(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], o, lse, do.contiguous(), dqkv[:, :inner],
                     dqkv[:, inner:2 * inner], dqkv[:, 2 * inner:], B, H, d, N, 0, N, 0, d ** -0.5)
        return dqkv, None, None, None, None


class CrossAttnFn(torch.autograd.Function):
    """q [B*Nq, H*d], kv [B*Nk, 2*H*d] packed (k | v) -> o [B*Nq, H*d]"""

    @staticmethod
    def forward(ctx, q, kv, B, Nq, Nk, H, d):
        inner = H * d
        o = torch.empty(B * Nq, inner, dtype=q.dtype, device=q.device)
        lse = torch.empty(B, H, Nq, dtype=torch.float32, device=q.device)
        ops.attn_fwd(q, kv[:, :inner], kv[:, inner:], o, lse, B, H, d, Nq, 0, Nk, 0, d ** -0.5)
        ctx.save_for_backward(q, kv, o, lse)
        ctx.cfg = (B, Nq, Nk, H, d)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, o, lse = ctx.saved_tensors
        B, Nq, Nk, H, d = ctx.cfg
        inner = H * d
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        ops.attn_bwd(q, kv[:, :inner], kv[:, inner:], o, lse, do.contiguous(), dq, dkv[:, :inner], dkv[:, inner:], B, H, d, Nq,
                     0, Nk, 0, d ** -0.5)
        return dq, dkv, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ elementwise
class GegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        M, c2 = h.shape
        out = torch.empty(M, c2 // 2, dtype=h.dtype, device=h.device)
        ops.geglu_fwd(h, out)
        ctx.save_for_backward(h)
        return out

    @staticmethod
    def backward(ctx, dout):
        (h,) = ctx.saved_tensors
        dh = torch.empty_like(h)
        ops.geglu_bwd(h, dout.contiguous(), dh)
        return dh


class DropoutFn(torch.autograd.Function):
    """y = x * mask[(row // rows_per_group), :]; mask multipliers drawn here (Dropout / Dropout2d)."""

    @staticmethod
    def forward(ctx, x, p, rows_per_group):
        M, C = x.shape
        g = (M + rows_per_group - 1) // rows_per_group
        mask = torch.empty(g, C, dtype=torch.float32, device=x.device)
        seed, off = _next_rng(g * C)
        ops.dropout_mask(mask, p, seed, off)
        y = torch.empty_like(x)
        ops.mul_mask(x, mask, y, rows_per_group)
        ctx.save_for_backward(mask)
        ctx.rpg = rows_per_group
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dx = torch.empty_like(dy)
        ops.mul_mask(dy.contiguous(), mask, dx, ctx.rpg)
        return dx, None, None


def dropout(x, p, training, rows_per_group=1):
    if not training or p <= 0:
        return x
    return DropoutFn.apply(x, p, rows_per_group)


class MaskTokenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keep_u8, token):
        out = torch.empty_like(x)
        ops.mask_token_fwd(x.contiguous(), keep_u8, token.detach().reshape(-1).contiguous(), out)
        ctx.save_for_backward(keep_u8)
        ctx.tshape = token.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        (keep,) = ctx.saved_tensors
        dx = torch.empty_like(dout)
        dt = torch.empty(dout.shape[1], dtype=torch.float32, device=dout.device)
        ops.mask_token_bwd(dout.contiguous(), keep, dx, dt)
        return dx, None, dt.view(ctx.tshape)


class UnblockFn(torch.autograd.Function):
    """[B, H*W (blocked), C] fp32 -> [B,H,W,C]"""

    @staticmethod
    def forward(ctx, x, B, H, W, levels):
        C = x.shape[-1]
        y = torch.empty(B, H, W, C, dtype=torch.float32, device=x.device)
        ops.unblock(x.contiguous(), y, B, H, W, C, levels, False)
        ctx.cfg = (B, H, W, C, levels, x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C, levels, shp = ctx.cfg
        dx = torch.empty(shp, dtype=torch.float32, device=dy.device)
        ops.unblock(dy.contiguous(), dx, B, H, W, C, levels, True)
        return dx, None, None, None, None


class UpsampleCEFn(torch.autograd.Function):
    """loss = mean over ALL pixels of CE(bilinear_up(logits_low), label) (ignore_index contributes 0) - mmseg
    CrossEntropyLoss(avg_non_ignore=False).  The full-resolution logits are never materialised."""

    @staticmethod
    def forward(ctx, logits_low, label, ignore_index, loss_weight):
        loss, acc, dl = ops.upsample_ce_loss_acc(logits_low.contiguous(), label, ignore_index, need_grad=True)
        ctx.save_for_backward(dl)
        ctx.lw = loss_weight
        ctx.mark_non_differentiable(acc)
        if loss_weight != 1.0:
            ops.axpby(loss, loss_weight, loss, 0.0)
        return loss.view(()), acc

    @staticmethod
    def backward(ctx, dloss, _):
        (dl,) = ctx.saved_tensors
        g = dl.clone() if False else dl
        ops.scale_by_device_scalar(g, dloss.contiguous().view(1))
        if ctx.lw != 1.0:
            ops.axpby(g, ctx.lw, g, 0.0)
        return g, None, None, None
