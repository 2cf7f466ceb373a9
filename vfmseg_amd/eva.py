"""EVA02 ViT backbone on the HIP kernels (reference: rein/models/backbones/eva_02.py:614-853; Attention :245-407,
SwiGLU :204-242, VisionRotaryEmbeddingFast :119-160, Block :410-493) - BASELINE config 4.

Differences from the DINOv2 engine: separate (bias-less) q/k/v projections + q_bias / v_bias packed into one QKV GEMM,
2-D RoPE applied in place to the patch tokens' q and k, no LayerScale, SwiGLU MLP with an inner LayerNorm over 2730
channels (laid out 2752-wide so every GEMM K is a multiple of 64), LayerNorm eps 1e-5 (the cfg norm_layer is ignored,
eva_02.py:719), fixed abs pos-embed (input must be img_size).  LoRA: the reference computes q/k/v with
F.linear(x, proj.weight, bias) (:337-339), which bypasses peft's wrapped forward, so only the `attn.proj` adapter is in
the graph; here it is fused into the proj GEMM by K-concatenation [attn_out | s*drop(attn_out) A^T].
"""
import math

import torch
import torch.nn as nn

from .precision import is_half
from . import ops
from .backbones import R_PAD, LoraLinear, Packed, _BackboneFn, _Lin, _PatchEmbed, _pack_at, _refresh_sites, _wgrad_small_t
from .precision import compute_dtype
from .registry import MODELS


def rope_tables(half_head_dim, pt_seq_len, ft_seq_len, theta=10000.0):
    """eva_02.py:119-157 (freqs_for='lang'): cos/sin [ft*ft, 2*half_head_dim]."""
    freqs = 1.0 / (theta ** (torch.arange(0, half_head_dim, 2)[: half_head_dim // 2].float() / half_head_dim))
    t = torch.arange(ft_seq_len) / ft_seq_len * pt_seq_len
    f = torch.einsum("i,f->if", t, freqs).repeat_interleave(2, dim=-1)
    fy = f[:, None, :].expand(ft_seq_len, ft_seq_len, -1)
    fx = f[None, :, :].expand(ft_seq_len, ft_seq_len, -1)
    fr = torch.cat((fy, fx), dim=-1).reshape(ft_seq_len * ft_seq_len, -1)
    return fr.cos().contiguous(), fr.sin().contiguous()


class _Rope(nn.Module):
    def __init__(self, cos, sin):
        super().__init__()
        self.register_buffer("freqs_cos", cos)
        self.register_buffer("freqs_sin", sin)


class _EvaAttn(nn.Module):
    def __init__(self, dim, heads, qkv_bias):
        super().__init__()
        self.num_heads = heads
        self.q_proj, self.k_proj, self.v_proj = _Lin(dim, dim, False), _Lin(dim, dim, False), _Lin(dim, dim, False)
        self.q_bias = nn.Parameter(torch.zeros(dim)) if qkv_bias else None
        self.v_bias = nn.Parameter(torch.zeros(dim)) if qkv_bias else None
        self.proj = _Lin(dim, dim, True)


class _SwiGLU(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.w1, self.w2 = _Lin(dim, hidden), _Lin(dim, hidden)
        self.ffn_ln = nn.LayerNorm(hidden)
        self.w3 = _Lin(hidden, dim)


class _EvaBlock(nn.Module):
    def __init__(self, dim, heads, hidden, qkv_bias):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _EvaAttn(dim, heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _SwiGLU(dim, hidden)


@MODELS.register_module()
class EVA2(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=80, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4 * 2 / 3, qkv_bias=False, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
                 hybrid_backbone=None, norm_layer=None, init_values=None, use_checkpoint=False, use_abs_pos_emb=True,
                 use_rel_pos_bias=False, use_shared_rel_pos_bias=False, out_indices=(3, 5, 7, 11), subln=True, xattn=True,
                 naiveswiglu=True, rope=True, pt_hw_seq_len=16, intp_freq=True, pretrained=None, **kw):
        super().__init__()
        if not (subln and naiveswiglu and rope and use_abs_pos_emb) or init_values is not None or use_rel_pos_bias or use_shared_rel_pos_bias:
            raise NotImplementedError("HIP path implements the reference's EVA02 config (subln, SwiGLU, rope, abs pos, no gamma)")
        self.embed_dim = self.num_features = embed_dim
        self.patch_size, self.num_heads, self.img_size = patch_size, num_heads, img_size
        self.out_indices = list(out_indices)
        self.patch_embed = _PatchEmbed(patch_size, in_chans, embed_dim)
        n = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        hw = img_size // patch_size
        cos, sin = rope_tables(embed_dim // num_heads // 2, pt_hw_seq_len, hw if intp_freq else pt_hw_seq_len)
        self.rope = _Rope(cos, sin)
        hidden = int(embed_dim * mlp_ratio)
        self.blocks = nn.ModuleList([_EvaBlock(embed_dim, num_heads, hidden, qkv_bias) for _ in range(depth)])
        self._engine = None
        self.register_load_state_dict_post_hook(lambda m, keys: m.engine().invalidate())

    def engine(self):
        if self._engine is None:
            self._engine = EvaEngine(self)
        return self._engine

    def forward_tokens(self, jobs, training=False, seed=None):
        return _BackboneFn.apply(self, jobs, training, seed, *self.engine().trainable())

    def forward(self, x):
        xcat, (hp, wp) = self.forward_tokens([(x, None)], training=False)
        b, d, nt = x.shape[0], self.embed_dim, len(self.out_indices)
        v = xcat.view(b, hp, wp, nt, d)
        return tuple(v[:, :, :, i].permute(0, 3, 1, 2) for i in range(nt))


def _pad64(n):
    return (n + 63) // 64 * 64


class EvaEngine:
    def __init__(self, vit):
        self.vit = vit
        self._packed = None

    def invalidate(self):
        self._packed = None

    def lora_on(self):
        return isinstance(self.vit.blocks[0].attn.proj, LoraLinear)

    def trainable(self):
        out = []
        if self.lora_on():
            for blk in self.vit.blocks:
                out += [blk.attn.proj.lora_A["default"].weight, blk.attn.proj.lora_B["default"].weight]
        return out

    def inert_params(self):
        """LoRA factors on q/k/v exist (peft wraps them) but never enter the graph."""
        out = []
        for blk in self.vit.blocks:
            for nm in ("q_proj", "k_proj", "v_proj"):
                m = getattr(blk.attn, nm)
                if isinstance(m, LoraLinear):
                    out += [m.lora_A["default"].weight, m.lora_B["default"].weight]
        return out

    @staticmethod
    def _base(m):
        return m.base_layer if isinstance(m, LoraLinear) else m

    def packed(self):
        cd = compute_dtype()
        dev = self.vit.pos_embed.device
        if self._packed is not None and self._packed["cd"] == cd and self._packed["dev"] == dev:
            return self._packed
        v = self.vit
        D = v.embed_dim
        P = dict(cd=cd, dev=dev, layers=[])
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.no_grad():
            P["pe"] = Packed(v.patch_embed.proj.weight.detach().reshape(D, -1), cd)
            P["pe_b"] = v.patch_embed.proj.bias.detach().float().contiguous()
            P["cls"] = v.cls_token.detach().reshape(D).float().contiguous()
            P["pos"] = v.pos_embed.detach().reshape(-1, D).float().contiguous()
            P["cos"], P["sin"] = v.rope.freqs_cos.float().contiguous(), v.rope.freqs_sin.float().contiguous()
            for blk in v.blocks:
                a, m = blk.attn, blk.mlp
                hid = m.w1.weight.shape[0]
                hp = _pad64(hid)
                wqkv = torch.empty(3 * D, D, **f32)
                for i, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                    ops.cast(self._base(getattr(a, nm)).weight.detach(), wqkv[i * D:(i + 1) * D])
                bqkv = torch.zeros(3 * D, **f32)
                if a.q_bias is not None:
                    ops.cast(a.q_bias.detach().view(1, -1), bqkv[:D].view(1, -1))
                    ops.cast(a.v_bias.detach().view(1, -1), bqkv[2 * D:].view(1, -1))
                proj = self._base(a.proj)
                w12 = torch.zeros(2 * hp, D, **f32)      # rows: [w1 | 0-pad | w2 | 0-pad]
                ops.cast(m.w1.weight.detach(), w12[:hid])
                ops.cast(m.w2.weight.detach(), w12[hp:hp + hid])
                b12 = torch.zeros(2 * hp, **f32)
                ops.cast(m.w1.bias.detach().view(1, -1), b12[:hid].view(1, -1))
                ops.cast(m.w2.bias.detach().view(1, -1), b12[hp:hp + hid].view(1, -1))
                kp = D + (R_PAD if isinstance(a.proj, LoraLinear) else 0)
                Lp = dict(
                    qkv=Packed(wqkv, cd), qkv_b=bqkv,
                    proj=Packed(proj.weight.detach(), cd, k_pad=kp), proj_b=proj.bias.detach().float().contiguous(),
                    w12=Packed(w12, cd), b12=b12, hid=hid, hp=hp,
                    w3=Packed(m.w3.weight.detach(), cd, k_pad=hp), b3=m.w3.bias.detach().float().contiguous(),
                    n1w=blk.norm1.weight.detach().float().contiguous(), n1b=blk.norm1.bias.detach().float().contiguous(),
                    n2w=blk.norm2.weight.detach().float().contiguous(), n2b=blk.norm2.bias.detach().float().contiguous(),
                    n3w=m.ffn_ln.weight.detach().float().contiguous(), n3b=m.ffn_ln.bias.detach().float().contiguous(),
                )
                if isinstance(a.proj, LoraLinear):
                    Lp["a"] = torch.zeros(R_PAD, D, dtype=cd, device=dev)
                    Lp["at"] = torch.zeros(D, R_PAD, dtype=cd, device=dev)
                P["layers"].append(Lp)
        self._packed = P
        return P

    def refresh_lora(self, P):
        D = self.vit.embed_dim
        sites = []
        for blk, Lp in zip(self.vit.blocks, P["layers"]):
            q = blk.attn.proj
            if isinstance(q, LoraLinear):
                A, Bm = q.lora_A["default"].weight.detach(), q.lora_B["default"].weight.detach()
                sites.append((A, Bm, Lp["a"], Lp["at"], Lp["proj"].w, Lp["proj"].wt, q.r, A.shape[1], Bm.shape[0], D))
        _refresh_sites(P, sites)

    def forward(self, jobs, training, seed):
        v, P = self.vit, self.packed()
        cd, dev = P["cd"], P["dev"]
        D, H, ps = v.embed_dim, v.num_heads, v.patch_size
        hd = D // H
        lora = self.lora_on()
        if lora:
            self.refresh_lora(P)
        hp = wp = v.img_size // ps
        for img, box in jobs:
            y0, y1, x0, x1 = box if box is not None else (0, img.shape[2], 0, img.shape[3])
            if (y1 - y0, x1 - x0) != (v.img_size, v.img_size):
                raise ValueError(f"EVA2 has a fixed pos-embed/rope grid: input must be {v.img_size}x{v.img_size} (eva_02.py:825-826)")
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
        P["pe"].fwd(A0, ptok, bias=P["pe_b"])
        x = torch.empty(M, D, dtype=torch.float32, device=dev)
        ops.assemble_tokens(ptok, P["cls"], P["pos"], x, nimg, Np, D)
        del ptok, A0
        nt = len(v.out_indices)
        xcat = torch.empty(Mp, nt * D, dtype=cd, device=dev)
        saved = []
        scale = hd ** -0.5
        from .functional import draw_seed
        seed, rng0 = draw_seed(seed, len(v.blocks) * M * D) if (lora and training) else (0, 0)
        for li, (blk, Lp) in enumerate(zip(v.blocks, P["layers"])):
            hid_p = Lp["hp"]
            S = {"x_in": x}
            a1 = torch.empty(M, D, dtype=cd, device=dev)
            st1 = torch.empty(M, 2, dtype=torch.float32, device=dev)
            ops.layernorm_fwd(x, Lp["n1w"], Lp["n1b"], 1e-5, a1, st1)
            qkv = torch.empty(M, 3 * D, dtype=cd, device=dev)
            Lp["qkv"].fwd(a1, qkv, bias=Lp["qkv_b"])
            ops.rope(qkv[:, :2 * D], Mp, Np, 2 * D, hd, P["cos"], P["sin"])          # patch tokens only; cls rows untouched
            kp = Lp["proj"].k
            ao = torch.empty(M, kp, dtype=cd, device=dev)                          # [attn_out | T]
            lse = torch.empty(nimg, H, Np + 1, dtype=torch.float32, device=dev)
            ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ao[:, :D], lse, nimg, H, hd, Np, 1, Np, 1, scale)
            if lora:
                q = blk.attn.proj
                xd, mask = ao[:, :D], None
                if training and q.p > 0:
                    mask = torch.empty(M, D, dtype=cd, device=dev)
                    ops.dropout_mask(mask, q.p, seed, offset=rng0 + li * M * D)
                    xd = torch.empty(M, D, dtype=cd, device=dev)
                    ops.mul_mask(ao[:, :D], mask, xd)
                ops.gemm(xd, Lp["a"], ao[:, D:D + R_PAD], alpha=q.scaling)
                S.update(xd=xd if mask is not None else None, mask=mask)
            xm = torch.empty(M, D, dtype=torch.float32, device=dev)
            Lp["proj"].fwd(ao, xm, bias=Lp["proj_b"], residual=x)
            a2 = torch.empty(M, D, dtype=cd, device=dev)
            st2 = torch.empty(M, 2, dtype=torch.float32, device=dev)
            ops.layernorm_fwd(xm, Lp["n2w"], Lp["n2b"], 1e-5, a2, st2)
            h12 = torch.empty(M, 2 * hid_p, dtype=cd, device=dev)
            Lp["w12"].fwd(a2, h12, bias=Lp["b12"])
            hidden = torch.empty(M, hid_p, dtype=torch.float32, device=dev)
            ops.swiglu_fwd(h12, hidden, hid_p)                                      # pad columns: silu(0)*0 = 0
            hn = torch.empty(M, hid_p, dtype=cd, device=dev)
            if hid_p > Lp["hid"]:
                hn[:, Lp["hid"]:].zero_()     # K padding of the next GEMM: only the pad columns need the fill (the LN writes the rest)
            st3 = torch.empty(M, 2, dtype=torch.float32, device=dev)
            ops.layernorm_fwd(hidden[:, :Lp["hid"]], Lp["n3w"], Lp["n3b"], 1e-5, hn[:, :Lp["hid"]], st3)
            xo = torch.empty(M, D, dtype=torch.float32, device=dev)
            Lp["w3"].fwd(hn, xo, bias=Lp["b3"], residual=xm)
            S.update(a1=a1, st1=st1, qkv=qkv, ao=ao, lse=lse, x_mid=xm, a2=a2, st2=st2, h12=h12, hidden=hidden, st3=st3, hn=hn)
            saved.append(S)
            x = xo
            for i, oi in enumerate(v.out_indices):   # (an index may be listed more than once: every copy is a tap of its own)
                if oi == li:
                    ops.cast(x[:Mp], xcat[:, i * D:(i + 1) * D])
        ctx = dict(saved=saved, nimg=nimg, Np=Np, M=M, Mp=Mp, P=P, training=training)
        return xcat, (hp, wp), ctx

    def backward(self, ctx, dxcat):
        from .functional import direct_grad_target
        v, P = self.vit, ctx["P"]
        cd, dev = P["cd"], P["dev"]
        D, H = v.embed_dim, v.num_heads
        hd = D // H
        scale = hd ** -0.5
        M, Mp, nimg, Np = ctx["M"], ctx["Mp"], ctx["nimg"], ctx["Np"]
        dx = torch.zeros(M, D, dtype=torch.float32, device=dev)
        grads = [None] * (2 * len(v.blocks))
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
            hid, hid_p = Lp["hid"], Lp["hp"]
            # ---- SwiGLU branch
            if t is None:  # (otherwise the previous iteration's LN1 backward already produced t = bf16(dx))
                add_tap(li)
                t = torch.empty(M, D, dtype=cd, device=dev)
                ops.cast(dx, t)
            dhn = torch.empty(M, hid_p, dtype=cd, device=dev)
            Lp["w3"].dgrad(t, dhn)
            dhid = torch.empty(M, hid_p, dtype=torch.float32, device=dev)
            if hid_p > hid:
                dhid[:, hid:].zero_()         # (pad columns only: the LN backward writes the rest)
            ops.layernorm_bwd(dhn[:, :hid], S["hidden"][:, :hid], Lp["n3w"], S["st3"], dhid[:, :hid], accumulate_dx=False)
            dh12 = torch.empty(M, 2 * hid_p, dtype=cd, device=dev)
            ops.swiglu_bwd(S["h12"], dhid, dh12, hid_p)
            dn = torch.empty(M, D, dtype=cd, device=dev)
            Lp["w12"].dgrad(dh12, dn)
            if fuse_t:
                ops.layernorm_bwd_scaled(dn, S["x_mid"], Lp["n2w"], S["st2"], dx, t, P["ones"], accumulate_dx=True)
            else:
                ops.layernorm_bwd(dn, S["x_mid"], Lp["n2w"], S["st2"], dx, accumulate_dx=True)
                ops.cast(dx, t)
            del dhn, dhid, dh12
            # ---- attention branch
            kp = Lp["proj"].k
            dao = torch.empty(M, kp, dtype=cd, device=dev)
            Lp["proj"].dgrad(t, dao)
            q = blk.attn.proj
            if isinstance(q, LoraLinear):
                r = q.r
                A, Bm = q.lora_A["default"].weight, q.lora_B["default"].weight
                ao = S["ao"]
                xd = S["xd"] if S["xd"] is not None else ao[:, :D]
                gBt = torch.empty(R_PAD, Bm.shape[0], dtype=torch.float32, device=dev)
                gAp = torch.empty(R_PAD, A.shape[1], dtype=torch.float32, device=dev)
                tB, tA = direct_grad_target(Bm), direct_grad_target(A)
                # with a flat gradient buffer the split-K combine scatters straight into B.grad [out, r] / A.grad [r, in]
                doneB = _wgrad_small_t(ao[:, D:D + R_PAD], t, gBt, scatter=None if tB is None else (tB, r, 1, r))   # dB^T = T^T @ d(proj out)
                doneA = _wgrad_small_t(dao[:, D:D + R_PAD], xd, gAp, alpha=q.scaling,
                                       scatter=None if tA is None else (tA, r, A.shape[1], 1))                    # dA = s * dT^T @ drop(attn_out)
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
                ep = dict(ep_mode=ops.EP_MUL, aux=S["mask"]) if S["mask"] is not None else {}
                ops.gemm(dao[:, D:D + R_PAD], Lp["at"], dao[:, :D], alpha=q.scaling, residual=dao[:, :D], **ep)
            qkv = S["qkv"]
            dqkv = torch.empty(M, 3 * D, dtype=cd, device=dev)
            ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], S["ao"][:, :D], S["lse"], dao[:, :D], dqkv[:, :D],
                         dqkv[:, D:2 * D], dqkv[:, 2 * D:], nimg, H, hd, Np, 1, Np, 1, scale)
            ops.rope(dqkv[:, :2 * D], Mp, Np, 2 * D, hd, P["cos"], P["sin"], inverse=True)
            dn1 = torch.empty(M, D, dtype=cd, device=dev)
            Lp["qkv"].dgrad(dqkv, dn1)
            if fuse_t and li > 0:  # dx becomes d(x_out) of block li-1: its tap gradient goes in first, then LN1 backward
                add_tap(li - 1)     # accumulates and emits t = bf16(dx) for that block's w3 dgrad
                ops.layernorm_bwd_scaled(dn1, S["x_in"], Lp["n1w"], S["st1"], dx, t, P["ones"], accumulate_dx=True)
            else:
                ops.layernorm_bwd(dn1, S["x_in"], Lp["n1w"], S["st1"], dx, accumulate_dx=True)
                t = None
            ctx["saved"][li] = None
            from .backbones import BACKWARD_EVENTS
            if BACKWARD_EVENTS["block_done"] is not None:
                BACKWARD_EVENTS["block_done"](li)
        return grads
