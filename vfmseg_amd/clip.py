"""CLIP ViT backbone on the HIP kernels (reference: rein/models/backbones/clip.py:174-348 CLIPVisionTransformer,
ResidualAttentionBlock :37-68, QuickGELU :18-20; config configs/_base_/models/lora_clip_ms_masked.py).

Differences from the DINOv2 engine: bias-less patch conv, class embedding added twice (clip.py:318-337), positional embedding
bilinearly re-interpolated to the token grid, ln_pre, nn.MultiheadAttention with a packed in_proj, no LayerScale, QuickGELU
MLP, LayerNorm eps 1e-5.  LoRA targets are `out_proj`, `mlp.c_fc`, `mlp.c_proj`: nn.MultiheadAttention consumes
out_proj.weight / .bias as tensors, so that adapter never enters the graph (confirmed by running the
reference: 48 gradient-less LoRA tensors); the two MLP adapters are fused into their GEMMs by K-concatenation
[LN2(x) | s*drop(LN2 x) A1^T] . [W_fc | B1]^T   and   [g | s*drop(g) A2^T] . [W_proj | B2]^T.
"""
import torch
import torch.nn as nn

from .precision import is_half
from . import ops
from .backbones import BACKWARD_EVENTS, R_PAD, LoraLinear, Packed, _BackboneFn, _Lin, _pack_at, _refresh_sites, _wgrad_small_t
from .precision import compute_dtype
from .registry import MODELS


class _MHA(nn.Module):
    """Parameter container with nn.MultiheadAttention's key layout."""

    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = _Lin(dim, dim, True)
        nn.init.xavier_uniform_(self.in_proj_weight)


class _ClipMlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.c_fc = _Lin(dim, 4 * dim, True)
        self.c_proj = _Lin(4 * dim, dim, True)


class _ResBlock(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.attn = _MHA(dim, heads)
        self.ln_1 = nn.LayerNorm(dim)
        self.mlp = _ClipMlp(dim)
        self.ln_2 = nn.LayerNorm(dim)


class _Transformer(nn.Module):
    def __init__(self, dim, layers, heads):
        super().__init__()
        self.resblocks = nn.ModuleList([_ResBlock(dim, heads) for _ in range(layers)])


class _Conv1(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(dim, 3, patch, patch))
        nn.init.trunc_normal_(self.weight, std=0.02)


@MODELS.register_module()
class CLIPVisionTransformer(nn.Module):
    """Same ctor kwargs / state_dict keys as clip.py:174-266 for get_embeddings=False.  The fpn1..4 modules the reference
    builds but never calls are kept as plain parameter containers so that reference checkpoints load key-for-key."""

    def __init__(self, input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512, drop_path_rate=0.0,
                 out_indices=(3, 5, 7, 11), pretrained=None, get_embeddings=False, **kw):
        super().__init__()
        if get_embeddings:
            raise NotImplementedError("get_embeddings=True (ln_post / proj) is not on the segmentation path")
        self.embed_dim = self.width = width
        self.patch_size, self.num_heads, self.input_resolution = patch_size, heads, input_resolution
        self.out_indices = list(out_indices)
        self.spatial_size = input_resolution // patch_size
        self.conv1 = _Conv1(width, patch_size)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.spatial_size ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _Transformer(width, layers, heads)
        if patch_size == 16:  # never called (clip.py:224-247); parameters only
            self.fpn1 = nn.Sequential(nn.GroupNorm(1, width), nn.ConvTranspose2d(width, width, 2, 2), nn.BatchNorm2d(width), nn.GELU(),
                                      nn.ConvTranspose2d(width, width, 2, 2))
            self.fpn2 = nn.Sequential(nn.GroupNorm(1, width), nn.ConvTranspose2d(width, width, 2, 2))
            self.fpn3 = nn.GroupNorm(1, width)
            self.fpn4 = nn.Sequential(nn.GroupNorm(1, width), nn.MaxPool2d(2, 2))
        self._engine = None
        self.register_load_state_dict_post_hook(lambda m, keys: m.engine().invalidate())

    @property
    def blocks(self):
        return self.transformer.resblocks

    def engine(self):
        if self._engine is None:
            self._engine = ClipEngine(self)
        return self._engine

    def forward_tokens(self, jobs, training=False, seed=None):
        return _BackboneFn.apply(self, jobs, training, seed, *self.engine().trainable())

    def forward(self, x):
        xcat, (hp, wp) = self.forward_tokens([(x, None)], training=False)
        b, d, nt = x.shape[0], self.embed_dim, len(self.out_indices)
        v = xcat.view(b, hp, wp, nt, d)
        return tuple(v[:, :, :, i].permute(0, 3, 1, 2) for i in range(nt))


_SITES = ("c_fc", "c_proj")


class ClipEngine:
    def __init__(self, vit):
        self.vit = vit
        self._packed = None
        self._pos_cache = {}

    def invalidate(self):
        self._packed = None
        self._pos_cache = {}

    def lora_on(self):
        m = self.vit.blocks[0].mlp
        return isinstance(m.c_fc, LoraLinear) and isinstance(m.c_proj, LoraLinear)

    def trainable(self):
        out = []
        if self.lora_on():
            for blk in self.vit.blocks:
                for nm in _SITES:
                    q = getattr(blk.mlp, nm)
                    out += [q.lora_A["default"].weight, q.lora_B["default"].weight]
        return out

    def inert_params(self):
        """The out_proj adapter exists (peft wraps it) but nn.MultiheadAttention never calls it."""
        out = []
        for blk in self.vit.blocks:
            m = blk.attn.out_proj
            if isinstance(m, LoraLinear):
                out += [m.lora_A["default"].weight, m.lora_B["default"].weight]
        return out

    @staticmethod
    def _base(m):
        return m.base_layer if isinstance(m, LoraLinear) else m

    def packed(self):
        cd = compute_dtype()
        dev = self.vit.positional_embedding.device
        if self._packed is not None and self._packed["cd"] == cd and self._packed["dev"] == dev:
            return self._packed
        v = self.vit
        D = v.embed_dim
        lora = self.lora_on()
        P = dict(cd=cd, dev=dev, layers=[])
        with torch.no_grad():
            P["pe"] = Packed(v.conv1.weight.detach().reshape(D, -1), cd)
            P["cls2"] = (2.0 * v.class_embedding.detach().float()).contiguous()      # added in the token AND in cls_pos
            P["lnp_w"], P["lnp_b"] = v.ln_pre.weight.detach().float().contiguous(), v.ln_pre.bias.detach().float().contiguous()
            for blk in v.blocks:
                a, m = blk.attn, blk.mlp
                op, fc, pr = self._base(a.out_proj), self._base(m.c_fc), self._base(m.c_proj)
                hid = fc.weight.shape[0]
                rp = R_PAD if lora else 0
                Lp = dict(
                    qkv=Packed(a.in_proj_weight.detach(), cd), qkv_b=a.in_proj_bias.detach().float().contiguous(),
                    out=Packed(op.weight.detach(), cd), out_b=op.bias.detach().float().contiguous(),
                    fc=Packed(fc.weight.detach(), cd, k_pad=D + rp), fc_b=fc.bias.detach().float().contiguous(),
                    pr=Packed(pr.weight.detach(), cd, k_pad=hid + rp), pr_b=pr.bias.detach().float().contiguous(), hid=hid,
                    n1w=blk.ln_1.weight.detach().float().contiguous(), n1b=blk.ln_1.bias.detach().float().contiguous(),
                    n2w=blk.ln_2.weight.detach().float().contiguous(), n2b=blk.ln_2.bias.detach().float().contiguous(),
                )
                if lora:
                    Lp["a1"] = torch.zeros(R_PAD, D, dtype=cd, device=dev)
                    Lp["at1"] = torch.zeros(D, R_PAD, dtype=cd, device=dev)
                    Lp["a2"] = ops.empty_ld(R_PAD, hid, cd, dev, zero=True)
                    Lp["at2"] = torch.zeros(hid, R_PAD, dtype=cd, device=dev)
                P["layers"].append(Lp)
        self._packed = P
        return P

    def refresh_lora(self, P):
        D = self.vit.embed_dim
        sites = []
        for blk, Lp in zip(self.vit.blocks, P["layers"]):
            for nm, ka, kat, kw, K in (("c_fc", "a1", "at1", "fc", D), ("c_proj", "a2", "at2", "pr", Lp["hid"])):
                q = getattr(blk.mlp, nm)
                A, Bm = q.lora_A["default"].weight.detach(), q.lora_B["default"].weight.detach()
                sites.append((A, Bm, Lp[ka], Lp[kat], Lp[kw].w, Lp[kw].wt, q.r, A.shape[1], Bm.shape[0], K))
        _refresh_sites(P, sites)

    def pos_tokens(self, hp, wp):
        """[1 + hp*wp, D]: row 0 = positional_embedding[0], rows 1.. = the spatial table bilinearly resized (clip.py:327-336)."""
        key = (hp, wp)
        if key in self._pos_cache:
            return self._pos_cache[key]
        v = self.vit
        D, ss = v.embed_dim, v.spatial_size
        pe = v.positional_embedding.detach().float().contiguous()
        out = torch.empty(1 + hp * wp, D, dtype=torch.float32, device=pe.device)
        ops.cast(pe[:1], out[:1])
        if (hp, wp) == (ss, ss):
            ops.cast(pe[1:], out[1:])
        else:
            ops.resize_bilinear(pe[1:], False, 1, ss, ss, D, out[1:], 0, (hp, wp))
        self._pos_cache[key] = out
        return out

    # ---- forward over a list of (image batch, crop box) jobs that share one token grid
    def forward(self, jobs, training, seed):
        v, P = self.vit, self.packed()
        cd, dev = P["cd"], P["dev"]
        D, H, ps = v.embed_dim, v.num_heads, v.patch_size
        hd = D // H
        lora = self.lora_on()
        if lora:
            self.refresh_lora(P)
        grids = set()
        for img, box in jobs:
            y0, y1, x0, x1 = box if box is not None else (0, img.shape[2], 0, img.shape[3])
            grids.add(((y1 - y0) // ps, (x1 - x0) // ps))
        assert len(grids) == 1, "all jobs of one backbone call must share the token grid"
        hp, wp = grids.pop()
        Np = hp * wp
        nimg = sum(j[0].shape[0] for j in jobs)
        Mp, M = nimg * Np, nimg * Np + nimg
        A0 = torch.empty(Mp, 3 * ps * ps, dtype=cd, device=dev)
        r0 = 0
        for img, box in jobs:
            b = img.shape[0]
            ops.patchify(img, A0[r0 * Np:(r0 + b) * Np], box=box, patch=ps)
            r0 += b
        ptok = torch.empty(Mp, D, dtype=torch.float32, device=dev)
        P["pe"].fwd(A0, ptok)
        x0_ = torch.empty(M, D, dtype=torch.float32, device=dev)
        ops.assemble_tokens(ptok, P["cls2"], self.pos_tokens(hp, wp), x0_, nimg, Np, D)
        x = torch.empty(M, D, dtype=torch.float32, device=dev)
        ops.layernorm_fwd(x0_, P["lnp_w"], P["lnp_b"], 1e-5, x, None)             # ln_pre (frozen: no backward needed)
        del ptok, A0, x0_
        nt = len(v.out_indices)
        xcat = torch.empty(Mp, nt * D, dtype=cd, device=dev)
        saved = []
        scale = hd ** -0.5
        from .functional import draw_seed
        seed, rng0 = draw_seed(seed, 2 * len(v.blocks) * M * P["layers"][0]["hid"]) if (lora and training) else (0, 0)
        for li, (blk, Lp) in enumerate(zip(v.blocks, P["layers"])):
            hid = Lp["hid"]
            S = {"x_in": x}
            a1 = torch.empty(M, D, dtype=cd, device=dev)
            st1 = torch.empty(M, 2, dtype=torch.float32, device=dev)
            ops.layernorm_fwd(x, Lp["n1w"], Lp["n1b"], 1e-5, a1, st1)
            qkv = torch.empty(M, 3 * D, dtype=cd, device=dev)
            Lp["qkv"].fwd(a1, qkv, bias=Lp["qkv_b"])
            ao = torch.empty(M, D, dtype=cd, device=dev)
            lse = torch.empty(nimg, H, Np + 1, dtype=torch.float32, device=dev)
            ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ao, lse, nimg, H, hd, Np, 1, Np, 1, scale)
            xm = torch.empty(M, D, dtype=torch.float32, device=dev)
            Lp["out"].fwd(ao, xm, bias=Lp["out_b"], residual=x)
            # ---- MLP: c_fc (+LoRA) -> QuickGELU -> c_proj (+LoRA)
            k1 = Lp["fc"].k
            a2 = torch.empty(M, k1, dtype=cd, device=dev)                          # [LN2(x) | T1]
            st2 = torch.empty(M, 2, dtype=torch.float32, device=dev)
            xd1 = mask1 = xd2 = mask2 = None
            q1, q2 = (blk.mlp.c_fc, blk.mlp.c_proj) if lora else (None, None)
            if lora and training and q1.p > 0 and is_half(cd) and D % 256 == 0:
                mask1 = torch.empty(M, D, dtype=cd, device=dev)
                xd1 = torch.empty(M, D, dtype=cd, device=dev)
                ops.layernorm_dropout_fwd(xm, Lp["n2w"], Lp["n2b"], 1e-5, a2[:, :D], st2, xd1, mask1, q1.p, seed, offset=rng0 + 2 * li * M * hid)
            else:
                ops.layernorm_fwd(xm, Lp["n2w"], Lp["n2b"], 1e-5, a2[:, :D], st2)
                if lora and training and q1.p > 0:
                    mask1 = torch.empty(M, D, dtype=cd, device=dev)
                    ops.dropout_mask(mask1, q1.p, seed, offset=rng0 + 2 * li * M * hid)
                    xd1 = torch.empty(M, D, dtype=cd, device=dev)
                    ops.mul_mask(a2[:, :D], mask1, xd1)
            if lora:
                ops.gemm(xd1 if xd1 is not None else a2[:, :D], Lp["a1"], a2[:, D:D + R_PAD], alpha=q1.scaling)
            k2 = Lp["pr"].k
            hpre = torch.empty(M, hid, dtype=cd, device=dev)
            g = torch.empty(M, k2, dtype=cd, device=dev) if k2 != hid else ops.empty_ld(M, hid, cd, dev)   # [g | T2]
            Lp["fc"].fwd(a2, g[:, :hid], bias=Lp["fc_b"], ep_mode=ops.EP_QGELU, c2=hpre)
            if lora:
                src = g[:, :hid]
                if training and q2.p > 0:
                    mask2 = torch.empty(M, hid, dtype=cd, device=dev)
                    ops.dropout_mask(mask2, q2.p, seed, offset=rng0 + (2 * li + 1) * M * hid)
                    xd2 = torch.empty(M, hid, dtype=cd, device=dev)
                    ops.mul_mask(src, mask2, xd2)
                    src = xd2
                ops.gemm(src, Lp["a2"], g[:, hid:hid + R_PAD], alpha=q2.scaling)
            xo = torch.empty(M, D, dtype=torch.float32, device=dev)
            Lp["pr"].fwd(g, xo, bias=Lp["pr_b"], residual=xm)
            S.update(a1=a1, st1=st1, qkv=qkv, ao=ao, lse=lse, x_mid=xm, a2=a2, st2=st2, hpre=hpre, g=g, xd1=xd1, mask1=mask1,
                     xd2=xd2, mask2=mask2)
            saved.append(S)
            x = xo
            for i, oi in enumerate(v.out_indices):   # (an index may be listed more than once: every copy is a tap of its own)
                if oi == li:
                    ops.cast(x[:Mp], xcat[:, i * D:(i + 1) * D])
        ctx = dict(saved=saved, nimg=nimg, Np=Np, M=M, Mp=Mp, P=P, training=training)
        return xcat, (hp, wp), ctx

    def _lora_grads(self, q, T, dy, dT, xd, out, li, j, grads):
        """dB^T = T^T dy, dA = s dT^T xd for one adapter; straight into the flat gradient buffer when there is one."""
        from .functional import direct_grad_target
        A, Bm, r = q.lora_A["default"].weight, q.lora_B["default"].weight, q.r
        dev = T.device
        tB, tA = direct_grad_target(Bm), direct_grad_target(A)
        gBt = torch.empty(R_PAD, Bm.shape[0], dtype=torch.float32, device=dev)
        gAp = torch.empty(R_PAD, A.shape[1], dtype=torch.float32, device=dev)
        doneB = _wgrad_small_t(T, dy, gBt, scatter=None if tB is None else (tB, r, 1, r))
        doneA = _wgrad_small_t(dT, xd, gAp, alpha=q.scaling, scatter=None if tA is None else (tA, r, A.shape[1], 1))
        if doneB is not True:
            if tB is not None:
                ops.strided_copy(gBt, tB, (Bm.shape[0], r), (1, gBt.stride(0)), (r, 1), accumulate=True)
            else:
                gB = torch.empty_like(Bm, dtype=torch.float32)
                ops.strided_copy(gBt, gB, (Bm.shape[0], r), (1, gBt.stride(0)), (r, 1))
                grads[4 * li + 2 * j + 1] = gB
        if doneA is not True:
            if tA is not None:
                ops.axpby(gAp[:r].reshape(-1), 1.0, tA.view(-1), 1.0)
            else:
                grads[4 * li + 2 * j] = gAp[:r]

    # ---- backward: d(xcat) -> LoRA grads [dA_fc0, dB_fc0, dA_proj0, dB_proj0, ...]
    def backward(self, ctx, dxcat):
        v, P = self.vit, ctx["P"]
        cd, dev = P["cd"], P["dev"]
        D, H = v.embed_dim, v.num_heads
        hd = D // H
        scale = hd ** -0.5
        M, Mp, nimg, Np = ctx["M"], ctx["Mp"], ctx["nimg"], ctx["Np"]
        dx = torch.zeros(M, D, dtype=torch.float32, device=dev)
        lora = self.lora_on()
        grads = [None] * (4 * len(v.blocks))
        fuse_t = is_half(cd) and D % 256 == 0  # LN backward emits the next dgrad operand bf16(dx) itself
        if fuse_t and "ones" not in P:
            P["ones"] = torch.ones(D, dtype=torch.float32, device=dev)

        def add_tap(li):
            for i, oi in enumerate(v.out_indices):
                if oi == li:
                    src = dxcat[:, i * D:(i + 1) * D]
                    ops.strided_copy(src, dx, (Mp, D), (src.stride(0), 1), (D, 1), accumulate=True)
        t = None
        for li in range(len(v.blocks) - 1, -1, -1):
            blk, Lp, S = v.blocks[li], P["layers"][li], ctx["saved"][li]
            hid = Lp["hid"]
            # ---- MLP branch: x_out = x_mid + c_proj([g | T2])
            if t is None:  # (otherwise the previous iteration's LN1 backward already produced t = bf16(dx))
                add_tap(li)
                t = torch.empty(M, D, dtype=cd, device=dev)
                ops.cast(dx, t)
            k2 = Lp["pr"].k
            dg = torch.empty(M, k2, dtype=cd, device=dev)                       # d[g | T2]
            Lp["pr"].dgrad(t, dg)
            if lora:
                q2 = blk.mlp.c_proj
                xd2 = S["xd2"] if S["xd2"] is not None else S["g"][:, :hid]
                self._lora_grads(q2, S["g"][:, hid:hid + R_PAD], t, dg[:, hid:hid + R_PAD], xd2, None, li, 1, grads)
                ep = dict(ep_mode=ops.EP_MUL, aux=S["mask2"]) if S["mask2"] is not None else {}
                ops.gemm(dg[:, hid:hid + R_PAD], Lp["at2"], dg[:, :hid], alpha=q2.scaling, residual=dg[:, :hid], **ep)
            dh = ops.empty_ld(M, hid, cd, dev)
            ops.act_grad_mul(dg[:, :hid], S["hpre"], dh, ops.ACT_QGELU)
            k1 = Lp["fc"].k
            dn = torch.empty(M, k1, dtype=cd, device=dev)                       # d[LN2(x) | T1]
            Lp["fc"].dgrad(dh, dn)
            if lora:
                q1 = blk.mlp.c_fc
                xd1 = S["xd1"] if S["xd1"] is not None else S["a2"][:, :D]
                self._lora_grads(q1, S["a2"][:, D:D + R_PAD], dh, dn[:, D:D + R_PAD], xd1, None, li, 0, grads)
                ep = dict(ep_mode=ops.EP_MUL, aux=S["mask1"]) if S["mask1"] is not None else {}
                ops.gemm(dn[:, D:D + R_PAD], Lp["at1"], dn[:, :D], alpha=q1.scaling, residual=dn[:, :D], **ep)
            if fuse_t:
                ops.layernorm_bwd_scaled(dn[:, :D], S["x_mid"], Lp["n2w"], S["st2"], dx, t, P["ones"], accumulate_dx=True)
            else:
                ops.layernorm_bwd(dn[:, :D], S["x_mid"], Lp["n2w"], S["st2"], dx, accumulate_dx=True)
                ops.cast(dx, t)
            del dg, dh, dn
            # ---- attention branch: x_mid = x_in + out_proj(attn(in_proj(LN1(x_in))))
            dao = torch.empty(M, D, dtype=cd, device=dev)
            Lp["out"].dgrad(t, dao)
            qkv = S["qkv"]
            dqkv = torch.empty(M, 3 * D, dtype=cd, device=dev)
            ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], S["ao"], S["lse"], dao, dqkv[:, :D], dqkv[:, D:2 * D],
                         dqkv[:, 2 * D:], nimg, H, hd, Np, 1, Np, 1, scale)
            dn1 = torch.empty(M, D, dtype=cd, device=dev)
            Lp["qkv"].dgrad(dqkv, dn1)
            if fuse_t and li > 0:  # dx becomes d(x_out) of block li-1: its tap gradient goes in first, then LN1 backward
                add_tap(li - 1)
                ops.layernorm_bwd_scaled(dn1, S["x_in"], Lp["n1w"], S["st1"], dx, t, P["ones"], accumulate_dx=True)
            else:
                ops.layernorm_bwd(dn1, S["x_in"], Lp["n1w"], S["st1"], dx, accumulate_dx=True)
                t = None
            ctx["saved"][li] = None
            if BACKWARD_EVENTS["block_done"] is not None:
                BACKWARD_EVENTS["block_done"](li)
        return grads
