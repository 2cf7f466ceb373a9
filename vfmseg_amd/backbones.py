"""ViT backbones + LoRA wrapper on the HIP kernels (reference: rein/models/backbones/{dino_v2,lora_backbone}.py).

The nn.Modules below are *parameter containers* that reproduce the reference's state_dict key scheme
 and registry names; the arithmetic lives in `DinoEngine`, which drives the C-ABI kernels:

  token layout  : rows [0, n*Np) = patch tokens (image-major), rows [n*Np, n*Np+n) = the [cls] tokens.  GEMM row
                  tiles therefore never straddle an odd 1025-token boundary and feature taps are plain row slices.
  residual      : fp32 [M, D]; GEMM operands in the compute dtype (bf16 | f32), fp32 accumulation.
  LoRA          : fused into the QKV projection by K-concatenation: [LN(x) | s*drop(LN(x)) A^T] @ [W | B]^T, so
                  the rank-32 path costs no extra pass over the activations (peft lora.Linear semantics,
                  lora_backbone.py:16-23; scaling alpha/r).
  backward      : hand-written (frozen base => only activation gradients + LoRA A/B weight gradients).
"""
import math

import os

import torch
import torch.nn as nn

from .precision import is_half
from . import ops
from .precision import compute_dtype
from .registry import MODELS

R_PAD = 64  # LoRA rank padded to one MFMA K-block

# Data-parallel overlap (vfmseg_amd.parallel): the backbone backward is the last and longest autograd node, so it tells
# the gradient synchroniser when buckets become final: "heads" when it starts (all decoder gradients are done) and
# "lora" (with the block index just finished) as it walks the blocks from the last to the first.
BACKWARD_EVENTS = {"heads_done": None, "block_done": None}


# ------------------------------------------------------------------------------------------ parameter containers
class _Lin(nn.Module):
    def __init__(self, i, o, bias=True):
        super().__init__()
        self.in_features, self.out_features = i, o
        self.weight = nn.Parameter(torch.empty(o, i))
        self.bias = nn.Parameter(torch.zeros(o)) if bias else None
        nn.init.trunc_normal_(self.weight, std=0.02)


class LoraLinear(nn.Module):
    """peft 0.10 lora.Linear key layout: base_layer.{weight,bias}, lora_A.default.weight, lora_B.default.weight."""

    def __init__(self, base, r, alpha, dropout):
        super().__init__()
        self.base_layer = base
        self.r, self.scaling, self.p = r, alpha / r, dropout
        a, b = _Lin(base.in_features, r, False), _Lin(r, base.out_features, False)
        nn.init.kaiming_uniform_(a.weight, a=math.sqrt(5))
        nn.init.zeros_(b.weight)
        self.lora_A = nn.ModuleDict({"default": a})
        self.lora_B = nn.ModuleDict({"default": b})


class _Attn(nn.Module):
    def __init__(self, dim, heads, qkv_bias, proj_bias):
        super().__init__()
        self.num_heads = heads
        self.qkv = _Lin(dim, dim * 3, qkv_bias)
        self.proj = _Lin(dim, dim, proj_bias)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden, bias):
        super().__init__()
        self.fc1 = _Lin(dim, hidden, bias)
        self.fc2 = _Lin(hidden, dim, bias)


class _LayerScale(nn.Module):
    def __init__(self, dim, init):
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))


class _Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, qkv_bias, proj_bias, ffn_bias, init_values):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attn(dim, heads, qkv_bias, proj_bias)
        self.ls1 = _LayerScale(dim, init_values if init_values else 1.0)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio), ffn_bias)
        self.ls2 = _LayerScale(dim, init_values if init_values else 1.0)


class _PatchEmbed(nn.Module):
    def __init__(self, patch, in_chans, dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, patch, patch)


@MODELS.register_module()
class DinoVisionTransformer(nn.Module):
    """Same ctor kwargs / state_dict keys as rein/models/backbones/dino_v2.py:55-182 (ffn_layer='mlp' only)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=True, ffn_bias=True, proj_bias=True, drop_path_rate=0.0, drop_path_uniform=False,
                 init_values=None, ffn_layer="mlp", block_chunks=1, out_indices=(7, 11, 15, 23), init_cfg=None,
                 resize_feat=False, **kw):
        super().__init__()
        if ffn_layer != "mlp":
            raise NotImplementedError("only ffn_layer='mlp' (the reference's LoRA configs) is on the HIP path")
        if block_chunks not in (0, None):
            raise NotImplementedError("block_chunks must be 0 (all reference LoRA configs, lora_dinov2_ms_masked.py:27)")
        self.embed_dim = self.num_features = embed_dim
        self.patch_size, self.num_heads, self.n_blocks = patch_size, num_heads, depth
        self.out_indices = list(out_indices)
        self.img_size = img_size if isinstance(img_size, int) else img_size[0]
        self.patch_embed = _PatchEmbed(patch_size, in_chans, embed_dim)
        n = (self.img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.blocks = nn.ModuleList(
            [_Block(embed_dim, num_heads, mlp_ratio, qkv_bias, proj_bias, ffn_bias, init_values) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)  # present in checkpoints, NOT applied to the taps (dino_v2.py:252-268)
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        self._engine = None
        self.register_load_state_dict_post_hook(lambda m, keys: m.engine().invalidate())

    def engine(self):
        if self._engine is None:
            self._engine = DinoEngine(self)
        return self._engine

    def forward_tokens(self, jobs, training=False, seed=None):
        """jobs: list of (img [B,3,H,W] fp32 cuda, box or None) that share one token grid. Returns Xcat
        [sum(B)*Np, 4*D] (compute dtype) - the four taps side by side, token-major - and (hp, wp)."""
        return _BackboneFn.apply(self, jobs, training, seed, *self.engine().trainable())

    def forward(self, x):
        xcat, (hp, wp) = self.forward_tokens([(x, None)], training=self.training and self.engine().lora_on())
        b, d = x.shape[0], self.embed_dim
        v = xcat.view(b, hp, wp, len(self.out_indices), d)
        return tuple(v[:, :, :, i].permute(0, 3, 1, 2) for i in range(len(self.out_indices)))  # NCHW-shaped views


class _Holder(nn.Module):
    def __init__(self, m):
        super().__init__()
        self.model = m


class _PeftModel(nn.Module):
    def __init__(self, m):
        super().__init__()
        self.base_model = _Holder(m)


@MODELS.register_module()
class LoRABackbone(nn.Module):
    """rein/models/backbones/lora_backbone.py:10-44.  `checkpoint=None` (random init) is accepted - a documented
    deviation: the reference torch.load()s unconditionally."""

    def __init__(self, backbone, checkpoint=None, Lora_config=None, init_cfg=None, **kw):
        super().__init__()
        vit = MODELS.build(backbone)
        cfg = Lora_config or {}
        self.lora_targets = list(cfg.get("target_modules", ["qkv"]))
        # peft: a module is wrapped when its dotted name equals a target or ends with '.' + target
        for name, mod in list(vit.named_modules()):
            if isinstance(mod, _Lin) and any(name == t or name.endswith("." + t) for t in self.lora_targets):
                parent = vit
                parts = name.split(".")
                for p_ in parts[:-1]:
                    parent = getattr(parent, p_)
                setattr(parent, parts[-1], LoraLinear(mod, cfg.get("r", 32), cfg.get("lora_alpha", 32), cfg.get("lora_dropout", 0.0)))
        if not vit.engine().lora_on():
            raise NotImplementedError(f"target_modules={self.lora_targets}: no adapter site the {type(vit).__name__} HIP engine fuses")
        self.model = _PeftModel(vit)
        self._lora_train = False
        if checkpoint is not None:
            self.load_pretrained_backbone(checkpoint, self.lora_targets)
        self.train(True)

    @property
    def vit(self):
        return self.model.base_model.model

    def load_pretrained_backbone(self, checkpoint, target_modules):
        sd = torch.load(checkpoint, map_location="cpu") if isinstance(checkpoint, str) else checkpoint
        new = {}
        for name, w in sd.items():
            for t in target_modules:  # lora_backbone.py:30-34 rename
                if t in name:
                    name = name.replace(t, t + ".base_layer")
            new[name] = w
        self.vit.load_state_dict(new, strict=False)
        self.vit.engine().invalidate()

    def train(self, mode=True):
        """lora_backbone.py:37-41 + utils.py:9-58: only parameters whose name contains 'lora' are trainable and the
        base model stays in eval (no drop-path / dropout); modules named lora* (lora_dropout) are in train mode."""
        super().train(False)
        self._lora_train = bool(mode)
        for n, p in self.named_parameters():
            p.requires_grad = ("lora" in n) if mode else p.requires_grad
        if mode:
            # adapters that never enter the graph (EVA02 q/k/v) would only ever see weight decay here;
            # torch's AdamW skips grad-less parameters, so they are simply kept frozen
            for p in getattr(self.vit.engine(), "inert_params", lambda: [])():
                p.requires_grad = False
        return self

    def forward_tokens(self, jobs, seed=None):
        return self.vit.forward_tokens(jobs, training=self._lora_train, seed=seed)

    def forward(self, x):
        xcat, (hp, wp) = self.forward_tokens([(x, None)])
        b, d, nt = x.shape[0], self.vit.embed_dim, len(self.vit.out_indices)
        v = xcat.view(b, hp, wp, nt, d)
        return tuple(v[:, :, :, i].permute(0, 3, 1, 2) for i in range(nt))


# ------------------------------------------------------------------------------------------ packed weights
class Packed:
    """A frozen [N,K] weight in the compute dtype, plus (bf16 mode) its transpose for the dgrad GEMM."""

    def __init__(self, w_f32, cd, k_pad=None):
        n, k = w_f32.shape
        k_pad = k_pad or k
        self.n, self.k = n, k_pad
        self.w = ops.empty_ld(n, k_pad, cd, w_f32.device, zero=True)
        ops.cast(w_f32, self.w[:, :k])
        self.wt = None
        if is_half(cd):
            self.wt = ops.empty_ld(k_pad, n, cd, w_f32.device, zero=True)
            ops.transpose(w_f32, self.wt[:k], pad_rows=n)

    def fwd(self, a, out, **epi):
        return ops.gemm(a, self.w, out, **epi)

    def dgrad(self, dy, out, **epi):
        if self.wt is not None:
            return ops.gemm(dy, self.wt, out, **epi)
        return ops.gemm(dy, self.w, out, trans_b=True, **epi)


def wgrad(dy, x, grad_out, alpha=1.0, n_rows=None, accumulate=True):
    """grad_out[N,K] (+)= alpha * dy[:, :N]^T @ x   (reduction over the M tokens).
    bf16: explicit transposes feed the NT MFMA GEMM (K = tokens, zero-padded to 64); f32: strided operands."""
    M = dy.shape[0]
    res = grad_out if accumulate else None
    if is_half(dy.dtype):
        mp = (M + 63) // 64 * 64
        dyt = torch.empty(dy.shape[1], mp, dtype=dy.dtype, device=dy.device)
        xt = torch.empty(x.shape[1], mp, dtype=x.dtype, device=x.device)
        ops.transpose(dy, dyt, pad_rows=mp)
        ops.transpose(x, xt, pad_rows=mp)
        ops.gemm(dyt, xt, grad_out, alpha=alpha, residual=res)
    else:
        ops.gemm(dy, x, grad_out, alpha=alpha, residual=res, trans_a=True, trans_b=True)
    return grad_out


# ------------------------------------------------------------------------------------------ launch plan of the train step
class _DinoTrainPlan:
    """The backbone's train-step launch sequence (LoRA on, 16-bit mode, fused LN + dropout) as launch plans (ops.Plan -> vfm_run_plan):
    9 launches per block forward, 9-11 backward, over PERSISTENT layer-batched activation buffers, replayed by one C call per pass
    instead of ~430 Python-issued launches (12.2 -> ~6 ms of host enqueue per step).  Same kernels, same arguments,
    same order as DinoEngine.forward / .backward issue one by one: results are bit-identical (tests/test_backbone_gpu.py).
    Buffers (3.9 GB at depth 24, bs 2: 1.4 % of the 288 GB): X[l] the fp32 residual stream entering block l, A1 = [LN1 x | T] the
    K-extended QKV operand, XD the dropped copy, MASK the dropout multiplier, QKV, AO, LSE, XMID, A2, HPRE = gelu'(fc1 pre-activation),
    and for backward DQKV / DA1 of every layer (the batched LoRA weight gradients read them) + DX, T, DH, DN, DAO."""

    def __init__(self, eng, P, nimg, Np):
        v = eng.vit
        cd, dev = P["cd"], P["dev"]
        D, H = v.embed_dim, v.num_heads
        nL = len(v.blocks)
        M, Mp = nimg * Np + nimg, nimg * Np
        Lp0 = P["layers"][0]
        kq, hid = Lp0["qkv"].k, Lp0["fc1"].n
        q0 = v.blocks[0].attn.qkv
        self.eng, self.P, self.nimg, self.Np, self.M, self.Mp, self.nL, self.D, self.H = eng, P, nimg, Np, M, Mp, nL, D, H
        self.busy = False
        e = lambda *sh, dt=cd: torch.empty(*sh, dtype=dt, device=dev)   # noqa: E731
        f32 = torch.float32
        self.X = e(nL + 1, M, D, dt=f32)
        self.A1, self.XD, self.MASK = e(nL, M, kq), e(nL, M, D), e(nL, M, D)
        self.ST1, self.ST2 = e(nL, M, 2, dt=f32), e(nL, M, 2, dt=f32)
        self.QKV, self.AO = e(nL, M, 3 * D), e(nL, M, D)
        self.LSE = e(nL, nimg, H, Np + 1, dt=f32)
        self.XMID, self.A2 = e(nL, M, D, dt=f32), e(nL, M, D)
        hld = hid + ops.ld_pad(hid, cd)
        self.HPRE = e(nL, M, hld)[:, :, :hid]
        self.G = e(M, hld)[:, :hid]
        self.hid, self.kq = hid, kq
        hd = D // H
        scale = hd ** -0.5
        # ---- forward
        f = self.fwd = ops.Plan("backbone")
        self.tap_ops = []     # (plan entry, column offset into xcat): their destination is this step's xcat
        self.nt = len(v.out_indices)
        xcat0 = e(Mp, self.nt * D)   # placeholder destination (patched every step)
        for li, (blk, Lp) in enumerate(zip(v.blocks, P["layers"])):
            q = blk.attn.qkv
            x, a1 = self.X[li], self.A1[li]
            f.layernorm_dropout_fwd(x, Lp["n1w"], Lp["n1b"], 1e-6, a1[:, :D], self.ST1[li], self.XD[li], self.MASK[li], q.p, li * M * D)
            f.gemm(self.XD[li], Lp["a"], a1[:, D:D + R_PAD], alpha=q.scaling)
            f.gemm(a1, Lp["qkv"].w, self.QKV[li], bias=Lp["qkv_b"])
            qkv = self.QKV[li]
            f.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], self.AO[li], self.LSE[li], nimg, H, hd, Np, 1, Np, 1, scale)
            f.gemm(self.AO[li], Lp["proj"].w, self.XMID[li], bias=Lp["proj_b"], colscale=Lp["g1"], residual=x)
            f.layernorm_fwd(self.XMID[li], Lp["n2w"], Lp["n2b"], 1e-6, self.A2[li], self.ST2[li])
            f.gemm(self.A2[li], Lp["fc1"].w, self.G, bias=Lp["fc1_b"], ep_mode=ops.EP_GELU_DGELU, c2=self.HPRE[li])
            f.gemm(self.G, Lp["fc2"].w, self.X[li + 1], bias=Lp["fc2_b"], colscale=Lp["g2"], residual=self.XMID[li])
            for i, oi in enumerate(v.out_indices):
                if oi == li:
                    self.tap_ops.append((f.cast(self.X[li + 1][:Mp], xcat0[:, i * D:(i + 1) * D]), i * D))
        self.bwd = None

    def eligible_backward(self):
        from .functional import direct_grad_target
        v = self.eng.vit
        return (os.environ.get("VFMSEG_LORA_WGRAD_BATCHED", "1") != "0" and
                all(direct_grad_target(b.attn.qkv.lora_A["default"].weight) is not None and
                    direct_grad_target(b.attn.qkv.lora_B["default"].weight) is not None for b in v.blocks))

    def run_forward(self, xcat, seed, rng0):
        esz = xcat.element_size()
        for idx, col in self.tap_ops:
            self.fwd.entry(idx).u.cast.dst = xcat.data_ptr() + col * esz
        self.fwd.run(seed, rng0)

    def _build_backward(self, dxcat):
        """Two plans: blocks L-1 .. L/2 and L/2-1 .. 0 (the batched LoRA weight gradients of a half, and the DP bucket of the first half,
        go out between them)."""
        v, P = self.eng.vit, self.P
        cd, dev = P["cd"], P["dev"]
        D, H, nL, M, Mp, nimg, Np = self.D, self.H, self.nL, self.M, self.Mp, self.nimg, self.Np
        hd = D // H
        scale = hd ** -0.5
        e = lambda *sh, dt=cd: torch.empty(*sh, dtype=dt, device=dev)   # noqa: E731
        self.DX = e(M, D, dt=torch.float32)
        self.T, self.DN, self.DAO = e(M, D), e(M, D), e(M, D)
        self.DH = e(M, self.hid + ops.ld_pad(self.hid, cd))[:, :self.hid]
        self.DQKV, self.DA1 = e(nL, M, 3 * D), e(nL, M, self.kq)
        half = nL // 2
        skip_l0 = os.environ.get("VFMSEG_PLAN_SKIP_L0", "1") != "0"
        self.half = half
        self.dx_sig = (tuple(dxcat.shape), tuple(dxcat.stride()), dxcat.dtype)
        plans, self.tap_src = [], []      # tap_src: (plan number, entry, column offset into dxcat)

        def add_tap(pl, pn, li):
            for i, oi in enumerate(v.out_indices):
                if oi == li:
                    src = dxcat[:, i * D:(i + 1) * D]
                    self.tap_src.append((pn, pl.strided_copy(src, self.DX, (Mp, D), (src.stride(0), 1), (D, 1), accumulate=True), i * D))

        for pn, (hi, lo) in enumerate(((nL, half), (half, 0)) if half > 0 else ((nL, 0),)):
            pl = ops.Plan("backbone")
            for li in range(hi - 1, lo - 1, -1):
                blk, Lp = v.blocks[li], P["layers"][li]
                q = blk.attn.qkv
                if li == nL - 1:
                    add_tap(pl, pn, li)
                    pl.cast(self.DX, self.T, Lp["g2"])
                pl.gemm(self.T, Lp["fc2"].wt, self.DH, ep_mode=ops.EP_MUL, aux=self.HPRE[li])
                pl.gemm(self.DH, Lp["fc1"].wt, self.DN)
                pl.layernorm_bwd_scaled(self.DN, self.XMID[li], Lp["n2w"], self.ST2[li], self.DX, self.T, Lp["g1"], accumulate_dx=True)
                pl.gemm(self.T, Lp["proj"].wt, self.DAO)
                qkv, dqkv, da1 = self.QKV[li], self.DQKV[li], self.DA1[li]
                pl.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], self.AO[li], self.LSE[li], self.DAO, dqkv[:, :D], dqkv[:, D:2 * D],
                            dqkv[:, 2 * D:], nimg, H, hd, Np, 1, Np, 1, scale)
                if li == 0 and skip_l0:
                    # block 0: nothing upstream of LN1 is trainable (patch embedding, position embedding and [cls] token are frozen,
                    # lora_backbone.py:36-44), so of d[LN1 x | T] only dT - the LoRA factors' gradient operand - is needed: 64 of the 1088
                    # columns of the qkv input gradient, no LoRA dgrad GEMM, no LN1 backward
                    pl.gemm(dqkv, Lp["qkv"].wt[D:D + R_PAD], da1[:, D:D + R_PAD])
                    continue
                pl.gemm(dqkv, Lp["qkv"].wt, da1)
                pl.gemm(da1[:, D:D + R_PAD], Lp["at"], da1[:, :D], alpha=q.scaling, residual=da1[:, :D], ep_mode=ops.EP_MUL, aux=self.MASK[li])
                if li > 0:
                    add_tap(pl, pn, li - 1)
                    pl.layernorm_bwd_scaled(da1[:, :D], self.X[li], Lp["n1w"], self.ST1[li], self.DX, self.T, P["layers"][li - 1]["g2"],
                                            accumulate_dx=True)
                else:   # (the one-by-one path's LN1 backward of block 0; its dx is not used by anything)
                    pl.layernorm_bwd_scaled(da1[:, :D], self.X[li], Lp["n1w"], self.ST1[li], self.DX, self.T, Lp["g2"], accumulate_dx=True)
            plans.append((pl, lo, hi))
        self.bwd = plans

    def run_backward(self, ctx, dxcat):
        sig = (tuple(dxcat.shape), tuple(dxcat.stride()), dxcat.dtype)
        if self.bwd is None or self.dx_sig != sig:
            self._build_backward(dxcat)
        esz = dxcat.element_size()
        for pn, idx, col in self.tap_src:
            self.bwd[pn][0].entry(idx).u.copy.src = dxcat.data_ptr() + col * esz
        self.DX.zero_()
        for pl, lo, hi in self.bwd:
            pl.run()
            self.eng._lora_wgrads_batched(self.P, ctx, self.DA1, self.DQKV, lo, hi, self.M, self.D)
            if BACKWARD_EVENTS["block_done"] is not None:
                for lj in range(hi - 1, lo - 1, -1):
                    BACKWARD_EVENTS["block_done"](lj)


# ------------------------------------------------------------------------------------------ engine
class DinoEngine:
    def __init__(self, vit):
        self.vit = vit
        self._packed = None
        self._pos_cache = {}

    def invalidate(self):
        self._packed = None
        self._pos_cache = {}

    def lora_on(self):
        return isinstance(self.vit.blocks[0].attn.qkv, LoraLinear)

    def trainable(self):
        out = []
        if self.lora_on():
            for blk in self.vit.blocks:
                out += [blk.attn.qkv.lora_A["default"].weight, blk.attn.qkv.lora_B["default"].weight]
        return out

    # ---- frozen weights, packed once per (dtype, device)
    def packed(self):
        cd = compute_dtype()
        dev = self.vit.pos_embed.device
        if self._packed is not None and self._packed["cd"] == cd and self._packed["dev"] == dev:
            return self._packed
        v = self.vit
        D = v.embed_dim
        P = dict(cd=cd, dev=dev, layers=[])
        with torch.no_grad():
            pw = v.patch_embed.proj.weight.detach().reshape(D, -1)
            P["pe"] = Packed(pw, cd)
            P["pe_b"] = v.patch_embed.proj.bias.detach().float().contiguous()
            P["cls"] = v.cls_token.detach().reshape(D).float().contiguous()
            for blk in v.blocks:
                qkv = blk.attn.qkv
                base = qkv.base_layer if isinstance(qkv, LoraLinear) else qkv
                kq = D + (R_PAD if isinstance(qkv, LoraLinear) else 0)
                Lp = dict(
                    qkv=Packed(base.weight.detach(), cd, k_pad=kq),
                    qkv_b=base.bias.detach().float().contiguous() if base.bias is not None else None,
                    proj=Packed(blk.attn.proj.weight.detach(), cd),
                    proj_b=blk.attn.proj.bias.detach().float().contiguous() if blk.attn.proj.bias is not None else None,
                    fc1=Packed(blk.mlp.fc1.weight.detach(), cd),
                    fc1_b=blk.mlp.fc1.bias.detach().float().contiguous() if blk.mlp.fc1.bias is not None else None,
                    fc2=Packed(blk.mlp.fc2.weight.detach(), cd),
                    fc2_b=blk.mlp.fc2.bias.detach().float().contiguous() if blk.mlp.fc2.bias is not None else None,
                    g1=blk.ls1.gamma.detach().float().contiguous(), g2=blk.ls2.gamma.detach().float().contiguous(),
                    n1w=blk.norm1.weight.detach().float().contiguous(), n1b=blk.norm1.bias.detach().float().contiguous(),
                    n2w=blk.norm2.weight.detach().float().contiguous(), n2b=blk.norm2.bias.detach().float().contiguous(),
                )
                if isinstance(qkv, LoraLinear):
                    Lp["a"] = torch.zeros(R_PAD, D, dtype=cd, device=dev)       # A padded to 64 rows   [r, in]
                    Lp["at"] = torch.zeros(D, R_PAD, dtype=cd, device=dev)      # A^T                   [in, r]
                P["layers"].append(Lp)
        self._packed = P
        return P

    def refresh_lora(self, P):
        """LoRA factors change every optimiser step: re-pack them into the concatenated QKV operand (one launch)."""
        D = self.vit.embed_dim
        sites = []
        for blk, Lp in zip(self.vit.blocks, P["layers"]):
            q = blk.attn.qkv
            if isinstance(q, LoraLinear):
                A, Bm = q.lora_A["default"].weight.detach(), q.lora_B["default"].weight.detach()
                sites.append((A, Bm, Lp["a"], Lp["at"], Lp["qkv"].w, Lp["qkv"].wt, q.r, A.shape[1], Bm.shape[0], D))
        _refresh_sites(P, sites)

    def merged_qkv(self, P):
        """Inference only (no LoRA dropout): per layer the bf16 copy of W + s * B A, so that the QKV projection is ONE K = D GEMM
        (no T = x A^T GEMM, no K extension).  Rebuilt when a LoRA factor changed (torch version counters, data pointers, and
        the optimiser's epoch for the fused AdamW kernel that updates parameters behind torch's back)."""
        from .optim import PARAM_EPOCH
        key = [PARAM_EPOCH[0]]
        for blk in self.vit.blocks:
            q = blk.attn.qkv
            A, Bm = q.lora_A["default"].weight, q.lora_B["default"].weight
            key += [A._version, Bm._version, A.data_ptr(), Bm.data_ptr()]
        key = tuple(key)
        if P.get("merged_key") == key:
            return P["merged"]
        D, cd, dev = self.vit.embed_dim, P["cd"], P["dev"]
        merged = []
        with torch.no_grad():
            for blk in self.vit.blocks:
                q = blk.attn.qkv
                A, Bm = q.lora_A["default"].weight.detach().float(), q.lora_B["default"].weight.detach().float()
                w = q.base_layer.weight.detach().float()
                weff = torch.empty_like(w)
                ops.gemm(Bm.contiguous(), A.contiguous(), weff, alpha=q.scaling, residual=w, trans_b=True)   # W + s * B A   (fp32 MFMA)
                wp = ops.empty_ld(w.shape[0], D, cd, dev)
                ops.cast(weff, wp)
                merged.append(wp)
        P["merged"], P["merged_key"] = merged, key
        return merged

    def pos_tokens(self, hp, wp):
        v = self.vit
        n = v.pos_embed.shape[1] - 1
        s = int(math.sqrt(n))
        if hp == s and wp == s:
            return v.pos_embed.detach().reshape(n + 1, -1).float().contiguous()
        key = (hp, wp)
        if key not in self._pos_cache:
            # dino_v2.py:184-215: bicubic with scale_factor = (h0 + 0.1) / sqrt(N); F.interpolate maps coordinates with
            # 1 / scale_factor.  The pos-embed is frozen, so the result is cached per token grid.
            D = v.embed_dim
            pe = v.pos_embed.detach().reshape(n + 1, D).float().contiguous()
            out = torch.empty(1 + hp * wp, D, dtype=torch.float32, device=pe.device)
            ops.cast(pe[:1], out[:1])
            sy, sx = s / (hp + 0.1), s / (wp + 0.1)
            assert int(s * ((hp + 0.1) / s)) == hp and int(s * ((wp + 0.1) / s)) == wp
            ops.resize_bicubic(pe[1:], s, s, D, out[1:], hp, wp, sy, sx)
            self._pos_cache[key] = out
        return self._pos_cache[key]

    # ---- forward
    def forward(self, jobs, training, seed, need_grad=False):
        v, P = self.vit, self.packed()
        cd = P["cd"]
        dev = P["dev"]
        D, H, ps = v.embed_dim, v.num_heads, v.patch_size
        lora = self.lora_on()
        merged = None
        from .precision import split3 as _x3_mode
        x3 = _x3_mode() and not is_half(cd)
        if (lora and not training and not torch.is_grad_enabled() and (is_half(cd) or x3)
                and os.environ.get("VFMSEG_MERGE_LORA_EVAL", "1") != "0"):
            merged = self.merged_qkv(P)
        elif lora:
            self.refresh_lora(P)
        sizes = []
        for img, box in jobs:
            y0, y1, x0, x1 = box if box is not None else (0, img.shape[2], 0, img.shape[3])
            sizes.append(((y1 - y0) // ps, (x1 - x0) // ps))
        hp, wp = sizes[0]
        assert all(s == (hp, wp) for s in sizes), "all jobs of one backbone call must share the token grid"
        Np = hp * wp
        nimg = sum(j[0].shape[0] for j in jobs)
        Mp, M = nimg * Np, nimg * Np + nimg
        kpe = 3 * ps * ps
        A0 = torch.empty(Mp, kpe, dtype=cd, device=dev)
        r0 = 0
        for img, box in jobs:
            b = img.shape[0]
            ops.patchify(img, A0[r0 * Np:(r0 + b) * Np], box=box, patch=ps)
            r0 += b
        ptok = torch.empty(Mp, D, dtype=torch.float32, device=dev)
        P["pe"].fwd(A0, ptok, bias=P["pe_b"])
        plan = self._train_plan(P, nimg, Np, training, merged) if need_grad else None
        x = plan.X[0] if plan is not None else torch.empty(M, D, dtype=torch.float32, device=dev)
        ops.assemble_tokens(ptok, P["cls"], self.pos_tokens(hp, wp), x, nimg, Np, D)
        del ptok, A0
        nt = len(v.out_indices)
        from .precision import eval_heads_fp32
        tap_dt = torch.float32 if (not training and not need_grad and eval_heads_fp32()) else cd
        xcat = torch.empty(Mp, nt * D, dtype=tap_dt, device=dev)
        saved = []
        if lora and training:
            from .functional import draw_seed
            seed, rng0 = draw_seed(seed, len(v.blocks) * M * D)
        else:
            seed, rng0 = 0, 0
        if plan is not None:
            plan.run_forward(xcat, seed, rng0)
            plan.busy = True
            ctx = dict(saved=None, plan=plan, nimg=nimg, Np=Np, M=M, Mp=Mp, P=P, training=training, A1all=plan.A1, XDall=plan.XD)
            return xcat, (hp, wp), ctx
        # training: the K-extended QKV operands [LN(x) | T] and the dropped LN copies of ALL layers live in two layer-batched
        # buffers, so that the LoRA weight gradients of many layers can run as ONE batched GEMM each (see backward)
        nL = len(v.blocks)
        batched = lora and training and is_half(cd) and merged is None
        A1all = torch.empty(nL, M, P["layers"][0]["qkv"].k, dtype=cd, device=dev) if batched else None
        XDall = None
        hd = D // H
        scale = hd ** -0.5
        for li, (blk, Lp) in enumerate(zip(v.blocks, P["layers"])):
            S = {"x_in": x}
            kq = D if merged is not None else Lp["qkv"].k
            # [LN(x) | T]: the T GEMM writes all R_PAD columns (A is zero-padded)
            a1 = A1all[li] if batched else torch.empty(M, kq, dtype=cd, device=dev)
            st1 = torch.empty(M, 2, dtype=torch.float32, device=dev)
            q = blk.attn.qkv if lora else None
            fused_drop = lora and training and q.p > 0 and is_half(cd) and D % 256 == 0
            if fused_drop:  # LN + dropout multiplier + dropped copy in one pass
                mask = torch.empty(M, D, dtype=cd, device=dev)
                if batched and XDall is None:
                    XDall = torch.empty(nL, M, D, dtype=cd, device=dev)
                xd = XDall[li] if batched else torch.empty(M, D, dtype=cd, device=dev)
                ops.layernorm_dropout_fwd(x, Lp["n1w"], Lp["n1b"], 1e-6, a1[:, :D], st1, xd, mask, q.p, seed, offset=rng0 + li * M * D)
            else:
                # bf16x3 predictions: outputs whose one consumer is a GEMM's A operand leave their producer as split-bf16 images
                so = "only" if (x3 and not training and not need_grad and kq == D) else None
                ops.layernorm_fwd(x, Lp["n1w"], Lp["n1b"], 1e-6, a1 if kq == D else a1[:, :D], st1, split_out=so)
            if lora and merged is None:
                if not fused_drop:
                    xd, mask = a1[:, :D], None
                    if training and q.p > 0:
                        mask = torch.empty(M, D, dtype=cd, device=dev)
                        ops.dropout_mask(mask, q.p, seed, offset=rng0 + li * M * D)
                        xd = torch.empty(M, D, dtype=cd, device=dev)
                        ops.mul_mask(a1[:, :D], mask, xd)
                ops.gemm(xd, Lp["a"], a1[:, D:D + R_PAD], alpha=q.scaling)  # T = s * drop(xn) A^T
                S.update(xd=xd if mask is not None else None, mask=mask)
            qkv = torch.empty(M, 3 * D, dtype=cd, device=dev)
            if merged is not None:
                ops.gemm(a1, merged[li], qkv, bias=Lp["qkv_b"])
            else:
                Lp["qkv"].fwd(a1, qkv, bias=Lp["qkv_b"])
            ao = torch.empty(M, D, dtype=cd, device=dev)
            lse = torch.empty(nimg, H, Np + 1, dtype=torch.float32, device=dev)
            so = "only" if (x3 and not training and not need_grad) else None
            ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ao, lse, nimg, H, hd, Np, 1, Np, 1, scale, keep_split=training, o_split=so)
            xm = torch.empty(M, D, dtype=torch.float32, device=dev)
            Lp["proj"].fwd(ao, xm, bias=Lp["proj_b"], colscale=Lp["g1"], residual=x)
            a2 = torch.empty(M, D, dtype=cd, device=dev)
            st2 = torch.empty(M, 2, dtype=torch.float32, device=dev)
            ops.layernorm_fwd(xm, Lp["n2w"], Lp["n2b"], 1e-6, a2, st2, split_out=so)
            hid = Lp["fc1"].n
            g = ops.empty_ld(M, hid, cd, dev)
            if training and lora:   # the forward also saves gelu'(pre-activation): what fc2's dgrad multiplies by (mlp.py:34-40)
                hpre = torch.empty(M, hid, dtype=cd, device=dev)
                Lp["fc1"].fwd(a2, g, bias=Lp["fc1_b"], ep_mode=ops.EP_GELU_DGELU, c2=hpre)
            else:
                hpre = None
                Lp["fc1"].fwd(a2, g, bias=Lp["fc1_b"], ep_mode=ops.EP_GELU, **({"c_split": so} if so else {}))
            xo = torch.empty(M, D, dtype=torch.float32, device=dev)
            Lp["fc2"].fwd(g, xo, bias=Lp["fc2_b"], colscale=Lp["g2"], residual=xm)
            S.update(a1=a1, st1=st1, qkv=qkv, ao=ao, lse=lse, x_mid=xm, a2=a2, st2=st2, hpre=hpre, g=g)
            saved.append(S)
            x = xo
            for i, oi in enumerate(v.out_indices):   # (an index may be listed more than once: every copy is a tap of its own)
                if oi == li:
                    ops.cast(x[:Mp], xcat[:, i * D:(i + 1) * D])
        ctx = dict(saved=saved, nimg=nimg, Np=Np, M=M, Mp=Mp, P=P, training=training, A1all=A1all, XDall=XDall)
        return xcat, (hp, wp), ctx

    def _train_plan(self, P, nimg, Np, training, merged):
        """The launch plan of this call's shape when the call is the hot one (training with gradients, LoRA with dropout on every
        block, 16-bit mode, weight gradients into the optimiser's flat buffer) and no earlier forward still owns the plan's buffers."""
        v = self.vit
        if (not training or merged is not None or not self.lora_on() or not is_half(P["cd"])
                or os.environ.get("VFMSEG_PLAN", "1") == "0" or v.embed_dim % 256 != 0):
            return None
        q0 = v.blocks[0].attn.qkv
        if not all(isinstance(b.attn.qkv, LoraLinear) and b.attn.qkv.p > 0 and b.attn.qkv.p == q0.p and b.attn.qkv.scaling == q0.scaling
                   for b in v.blocks):
            return None
        plans = P.setdefault("plans", {})
        pl = plans.get((nimg, Np))
        if pl is None:
            pl = plans[(nimg, Np)] = _DinoTrainPlan(self, P, nimg, Np)
        if pl.busy or not pl.eligible_backward():
            return None
        return pl

    # ---- backward: d(xcat) -> LoRA grads [dA0, dB0, dA1, dB1, ...]
    def backward(self, ctx, dxcat):
        if ctx.get("plan") is not None:
            plan = ctx["plan"]
            try:
                plan.run_backward(ctx, dxcat)
            finally:
                plan.busy = False
            return [None] * (2 * len(self.vit.blocks))
        v, P = self.vit, ctx["P"]
        cd, dev = P["cd"], P["dev"]
        D, H = v.embed_dim, v.num_heads
        hd = D // H
        scale = hd ** -0.5
        M, Mp, nimg, Np = ctx["M"], ctx["Mp"], ctx["nimg"], ctx["Np"]
        dx = torch.zeros(M, D, dtype=torch.float32, device=dev)
        grads = [None] * (2 * len(v.blocks))
        nL = len(v.blocks)
        # LoRA weight gradients, layer-batched: dB_l^T = T_l^T dqkv_l and dA_l = s dT_l^T drop(LN x)_l are 2 x 24 skinny GEMMs
        # (64 output rows, reduction over the 4100 tokens) that each needed split-K slabs + a combine launch (35 us per layer for
        # 0.3 % of the step's FLOPs).  Keeping dqkv / d[LN x | T] of the layers (0.8 GB of 288) turns them into TWO batched GEMMs
        # per half of the backbone - no split-K, no slabs - plus one batched scatter into the flat gradient buffer.  Two halves so
        # that the first LoRA gradient bucket can still leave (DP) while the second half of the backward runs.
        from .functional import direct_grad_target
        wg_batched = (ctx.get("A1all") is not None and self.lora_on() and os.environ.get("VFMSEG_LORA_WGRAD_BATCHED", "1") != "0" and
                      all(direct_grad_target(b.attn.qkv.lora_A["default"].weight) is not None and
                          direct_grad_target(b.attn.qkv.lora_B["default"].weight) is not None for b in v.blocks) and
                      (ctx.get("XDall") is not None or all(S_["mask"] is None for S_ in ctx["saved"])))
        if wg_batched:
            kq_all = ctx["A1all"].shape[2]
            DA1all = torch.empty(nL, M, kq_all, dtype=cd, device=dev)
            DQKVall = torch.empty(nL, M, 3 * D, dtype=cd, device=dev)
        half = nL // 2
        fuse_t = is_half(cd) and D % 256 == 0  # LN backward emits the next dgrad operand bf16(dx * gamma) itself
        def add_tap(li):
            for i, oi in enumerate(v.out_indices):
                if oi == li:
                    src = dxcat[:, i * D:(i + 1) * D]
                    ops.strided_copy(src, dx, (Mp, D), (src.stride(0), 1), (D, 1), accumulate=True)
        t = None
        for li in range(len(v.blocks) - 1, -1, -1):
            blk, Lp, S = v.blocks[li], P["layers"][li], ctx["saved"][li]
            q = blk.attn.qkv
            # ---- MLP branch: x_out = x_mid + g2 * fc2(gelu(fc1(LN2(x_mid))))
            if t is None:  # (the previous iteration's LN1 backward already produced t = bf16(dx * g2) otherwise)
                add_tap(li)
                t = torch.empty(M, D, dtype=cd, device=dev)
                ops.cast(dx, t, Lp["g2"])
            hid = Lp["fc1"].n
            dh = ops.empty_ld(M, hid, cd, dev)
            Lp["fc2"].dgrad(t, dh, ep_mode=ops.EP_MUL, aux=S["hpre"])
            dn = torch.empty(M, D, dtype=cd, device=dev)
            Lp["fc1"].dgrad(dh, dn)
            if fuse_t:
                ops.layernorm_bwd_scaled(dn, S["x_mid"], Lp["n2w"], S["st2"], dx, t, Lp["g1"], accumulate_dx=True)
            else:
                ops.layernorm_bwd(dn, S["x_mid"], Lp["n2w"], S["st2"], dx, accumulate_dx=True)
                ops.cast(dx, t, Lp["g1"])
            del dh
            # ---- attention branch: x_mid = x_in + g1 * proj(attn(qkv(LN1(x_in))))
            dao = torch.empty(M, D, dtype=cd, device=dev)
            Lp["proj"].dgrad(t, dao)
            dqkv = DQKVall[li] if wg_batched else torch.empty(M, 3 * D, dtype=cd, device=dev)
            qkv = S["qkv"]
            ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], S["ao"], S["lse"], dao, dqkv[:, :D], dqkv[:, D:2 * D],
                         dqkv[:, 2 * D:], nimg, H, hd, Np, 1, Np, 1, scale)
            kq = Lp["qkv"].k
            da1 = DA1all[li] if wg_batched else torch.empty(M, kq, dtype=cd, device=dev)
            Lp["qkv"].dgrad(dqkv, da1)
            if isinstance(q, LoraLinear) and wg_batched:
                # d LN1(x) = da1[:, :D] + mask * (s * dT @ A); the weight gradients of this layer wait for their batch
                ep = dict(ep_mode=ops.EP_MUL, aux=S["mask"]) if S["mask"] is not None else {}
                ops.gemm(da1[:, D:D + R_PAD], Lp["at"], da1[:, :D], alpha=q.scaling, residual=da1[:, :D], **ep)
            elif isinstance(q, LoraLinear):
                r = q.r
                A, Bm = q.lora_A["default"].weight, q.lora_B["default"].weight
                a1 = S["a1"]
                xd = S["xd"] if S["xd"] is not None else a1[:, :D]
                # dB^T[r, out] = T^T @ dqkv ;  dA[r, in] = s * dT^T @ drop(xn)   (reductions over the M tokens)
                from .functional import direct_grad_target
                tB, tA = direct_grad_target(Bm), direct_grad_target(A)
                gBt = torch.empty(R_PAD, Bm.shape[0], dtype=torch.float32, device=dev)
                gAp = torch.empty(R_PAD, A.shape[1], dtype=torch.float32, device=dev)
                # with a flat gradient buffer the split-K combine scatters straight into B.grad [out, r] / A.grad [r, in]
                doneB = _wgrad_small_t(a1[:, D:D + R_PAD], dqkv, gBt, scatter=None if tB is None else (tB, r, 1, r))
                doneA = _wgrad_small_t(da1[:, D:D + R_PAD], xd, gAp, alpha=q.scaling,
                                       scatter=None if tA is None else (tA, r, A.shape[1], 1))
                if doneB is True:
                    pass
                elif tB is not None:
                    ops.strided_copy(gBt, tB, (Bm.shape[0], r), (1, gBt.stride(0)), (r, 1), accumulate=True)
                else:
                    gB = torch.empty_like(Bm, dtype=torch.float32)
                    ops.strided_copy(gBt, gB, (Bm.shape[0], r), (1, gBt.stride(0)), (r, 1))
                    grads[2 * li + 1] = gB
                if doneA is True:
                    pass
                elif tA is not None:
                    ops.axpby(gAp[:r].reshape(-1), 1.0, tA.view(-1), 1.0)
                else:
                    grads[2 * li] = gAp[:r]
                # d LN1(x) = da1[:, :D] + mask * (s * dT @ A)
                ep = dict(ep_mode=ops.EP_MUL, aux=S["mask"]) if S["mask"] is not None else {}
                ops.gemm(da1[:, D:D + R_PAD], Lp["at"], da1[:, :D], alpha=q.scaling, residual=da1[:, :D], **ep)
            if fuse_t and li > 0:  # dx becomes d(x_out) of block li-1: its tap gradient goes in first, then LN1 backward
                add_tap(li - 1)     # accumulates and emits t = bf16(dx * g2[li-1]) for that block's fc2 dgrad
                ops.layernorm_bwd_scaled(da1[:, :D], S["x_in"], Lp["n1w"], S["st1"], dx, t, P["layers"][li - 1]["g2"],
                                         accumulate_dx=True)
            else:
                ops.layernorm_bwd(da1[:, :D], S["x_in"], Lp["n1w"], S["st1"], dx, accumulate_dx=True)
                t = None
            ctx["saved"][li] = None
            if wg_batched:
                if li == half or li == 0:
                    lo, hi = (half, nL) if li == half and half > 0 else (0, half if half > 0 else nL)
                    self._lora_wgrads_batched(P, ctx, DA1all, DQKVall, lo, hi, M, D)
                    if BACKWARD_EVENTS["block_done"] is not None:
                        for lj in range(hi - 1, lo - 1, -1):
                            BACKWARD_EVENTS["block_done"](lj)
            elif BACKWARD_EVENTS["block_done"] is not None:
                BACKWARD_EVENTS["block_done"](li)
        return grads

    def _lora_wgrads_batched(self, P, ctx, DA1all, DQKVall, lo, hi, M, D):
        """LoRA weight gradients of layers [lo, hi): two batched TN GEMMs + one batched scatter-accumulate into the flat buffer."""
        from .functional import direct_grad_target
        v = self.vit
        n = hi - lo
        dev = DA1all.device
        ws = P.setdefault("wg_ws", {})
        key = (lo, hi, M)
        w = ws.get(key)
        # dA has only D / 128 output tiles per layer (its 64 rows are one tile row): n * 8 = 96 blocks for 256 CUs.  The token
        # reduction is therefore cut into `ksp` equal row ranges that run as batch entries of their own (each layer's rows are
        # contiguous, so range c of layer l is entry ksp * l + c of ONE strided batch) and meet in the scatter-accumulate below.
        ksp = 4 if (M % 4 == 0 and n * (D // 128) * 4 <= 512) else (2 if (M % 2 == 0 and n * (D // 128) * 2 <= 512) else 1)
        if os.environ.get("VFMSEG_LORA_WGRAD_KSPLIT", "1") == "0":
            ksp = 1
        if w is None:
            w = ws[key] = dict(outB=torch.empty(n, R_PAD, 3 * D, dtype=torch.float32, device=dev),
                               outA=torch.empty(n * ksp, R_PAD, D, dtype=torch.float32, device=dev), table=None, sig=None)
        A1all, XDall = ctx["A1all"], ctx["XDall"]
        q0 = v.blocks[lo].attn.qkv
        ops.gemm_tn_batched(A1all[lo:hi, :, D:D + R_PAD], DQKVall[lo:hi], w["outB"], M)                       # dB^T = T^T dqkv
        Y = XDall[lo:hi] if XDall is not None else A1all[lo:hi, :, :D]
        dT = DA1all[lo:hi, :, D:D + R_PAD]
        if ksp > 1:
            mk = M // ksp
            dT = dT.as_strided((n * ksp, mk, R_PAD), (mk * dT.stride(1), dT.stride(1), 1), dT.storage_offset())
            Y = Y.as_strided((n * ksp, mk, D), (mk * Y.stride(1), Y.stride(1), 1), Y.storage_offset())
            ops.gemm_tn_batched(dT, Y, w["outA"], mk, alpha=q0.scaling)                                       # dA = s dT^T drop(LN x)
        else:
            ops.gemm_tn_batched(dT, Y, w["outA"], M, alpha=q0.scaling)
        tg = []
        for l in range(lo, hi):
            q = v.blocks[l].attn.qkv
            tg.append((direct_grad_target(q.lora_A["default"].weight), direct_grad_target(q.lora_B["default"].weight), q.r))
        sig = tuple((a.data_ptr(), b.data_ptr()) for a, b, _ in tg) + (ksp,)
        if w["table"] is None or w["sig"] != sig:
            jobs = []
            for i, (tA, tB, r) in enumerate(tg):
                jobs.append((w["outB"][i], tB, (3 * D, r), (1, 3 * D), (r, 1), True))     # B.grad[n, rr] += outB[rr, n]
                # A.grad[rr, k] += sum over the ksp token ranges of outA[rr, k] (ONE job: jobs of a table run concurrently)
                jobs.append((w["outA"][i * ksp], tA, (r, D), (D, 1), (D, 1), True, ksp, R_PAD * D))
            w["table"], w["sig"] = ops.CopyBatch(jobs), sig
        w["table"].run()


def _refresh_sites(P, sites):
    """sites: (A, B, a, at, w, wt, r, K, N, Kw) per adapter; fp32 contiguous parameters go through the batched pack kernel."""
    if not sites:
        return
    with torch.no_grad():
        if all(s_[0].dtype == torch.float32 and s_[0].is_contiguous() and s_[1].dtype == torch.float32 and s_[1].is_contiguous() for s_ in sites):
            tab = P.get("lora_table")
            if tab is None or not tab.valid_for(sites):
                tab = P["lora_table"] = ops.LoraPackTable(sites, sites[0][2])
            tab.run()
            return
        for (A, Bm, a, at, w, wt, r, K, N, Kw) in sites:
            ops.cast(A, a[:r])
            _pack_at(A, at, r)
            ops.cast(Bm, w[:, Kw:Kw + r])
            if wt is not None:
                ops.transpose(Bm, wt[Kw:Kw + r], pad_rows=Bm.shape[0])


def _pack_at(A, at, r):
    """at[:, :r] = A^T  (A [r, in])"""
    ops.strided_copy(A, at, (A.shape[1], r), (1, A.stride(0)), (at.stride(0), 1))


_KCH_MAX = int(os.environ.get("VFMSEG_WGRAD_KCH_MAX", "16"))


def _wgrad_small_t(xs, y, out, alpha=1.0, scatter=None):
    """out[P, Q] = alpha * xs^T @ y   with xs [M, P] small (P <= 64) and y [M, Q] large, consumed in place.
    bf16: only the small operand is transposed (zero-padded to a multiple of 64 tokens); the large one is the
    transposed-B operand of the MFMA GEMM.  f32: strided operands.
    scatter = (dst, rows_used, sp, sq): instead of writing `out`, accumulate dst[p*sp + q*sq] += result[p, q] for
    p < rows_used (the gradient's own layout inside the optimiser's flat buffer); returns True when that was done."""
    M = xs.shape[0]
    if is_half(xs.dtype):
        mp = (M + 63) // 64 * 64
        # few output tiles, long token reduction: split K into the largest divisor <= 16 of the 64-token steps
        steps = mp // 64
        kch = max(d for d in range(1, _KCH_MAX + 1) if steps % d == 0)
        tn = kch > 1 and xs.shape[1] % 8 == 0 and xs.data_ptr() % 16 == 0 and xs.stride(0) % 8 == 0
        if not tn:
            xt = torch.empty(xs.shape[1], mp, dtype=xs.dtype, device=xs.device)
            ops.transpose(xs, xt, pad_rows=mp)
        if kch > 1:
            P, Q = xs.shape[1], y.shape[1]
            slabs = torch.empty(kch, P, Q, dtype=torch.float32, device=xs.device)
            if tn:
                ops.gemm_splitk_tn(xs, y, slabs, kch)      # both operands token-major, consumed in place (no transposes)
            else:
                ops.gemm_splitk_bt(xt, y, slabs, kch)
            if scatter is not None:
                dst, rows_used, sp, sq = scatter
                ops.slab_reduce(slabs, rows_used, dst, sp, sq, alpha=alpha, accumulate=True)
                return True
            ops.colsum(slabs.view(kch, P * Q), out.view(P * Q))
            if alpha != 1.0:
                ops.axpby(out.view(-1), alpha, out.view(-1), 0.0)
        else:
            ops.gemm(xt, y, out, alpha=alpha, trans_b=True, kb_rows=M)
    else:
        ops.gemm(xs, y, out, alpha=alpha, trans_a=True, trans_b=True)
    return out


class _BackboneFn(torch.autograd.Function):
    """One autograd node for the whole ViT: forward saves activations, backward runs the hand-written block backward."""

    @staticmethod
    def forward(ctx, vit, jobs, training, seed, *lora_params):
        eng = vit.engine()
        need_grad = training and any(p.requires_grad for p in lora_params)
        with ops.region("backbone"):
            if isinstance(eng, DinoEngine):
                xcat, grid, c = eng.forward(jobs, training, seed, need_grad=need_grad)
            else:
                xcat, grid, c = eng.forward(jobs, training, seed)
        if need_grad:
            ctx.eng, ctx.c = eng, c
        else:
            c["saved"] = None
        ctx.nparams = len(lora_params)
        ctx.mark_non_differentiable()
        ctx.set_materialize_grads(False)
        ctx.grid = grid
        return xcat, _Grid(grid)

    @staticmethod
    def backward(ctx, dxcat, _):
        if dxcat is None:
            return (None, None, None, None) + (None,) * ctx.nparams
        if BACKWARD_EVENTS["heads_done"] is not None:
            from .functional import join_wgrad_stream
            join_wgrad_stream()     # (the heads' weight gradients run on a side stream; their DP buckets leave now)
            BACKWARD_EVENTS["heads_done"]()
        with ops.region("backbone"):
            grads = ctx.eng.backward(ctx.c, dxcat.contiguous())
        return (None, None, None, None) + tuple(grads)


class _Grid(tuple):
    """(hp, wp) carried through autograd.Function as a non-tensor output."""
    pass
