"""SAM ViT-H backbone on the HIP kernels, inference path (reference: rein/models/backbones/sam_vit.py:51-465) -
BASELINE config 5 (lora_sam_linear.py: EncoderDecoder + LoRA(qkv) SAM-H + LinearHead, 'slide' test mode).

NHWC tokens without a cls token: the residual stream is the matrix [n*G*G, 1280] (a multiple of 128 rows, no GEMM tail).
28 of the 32 blocks attend inside 14x14 windows of the zero-padded 42x42 grid (padded tokens are real keys whose
k/v equal the projection bias, as in the reference), 4 blocks attend globally; every block adds the decomposed
relative-position bias.  The bias is folded into augmented Q/K operands (see csrc/sam.hip), which turns the biased
attention into batched MFMA GEMMs + a row softmax; LoRA is merged into the QKV GEMM by K-concatenation as for DINOv2.
Training (lora_sam_ms_masked.py) differentiates the same form: dP = dO V^T, the row-softmax backward, dV^T = dO^T P,
dK^T = (scale q)^T dS and dQaug = dS Kaug are batched GEMMs whose large operand (P, dS, Kaug) is consumed in place as the
[K, N] operand; the rel-pos chain rule is folded into the final scatter back to token-major dqkv (csrc/sam.hip).
"""
import os

import torch
import torch.nn as nn

from .precision import is_half
from . import ops
from .backbones import BACKWARD_EVENTS, R_PAD, LoraLinear, Packed, _BackboneFn, _Lin, _PatchEmbed, _pack_at, _refresh_sites, _wgrad_small_t
from .precision import compute_dtype
from .registry import MODELS


class _SamAttn(nn.Module):
    def __init__(self, dim, heads, qkv_bias, use_rel_pos, L):
        super().__init__()
        self.num_heads = heads
        self.qkv = _Lin(dim, dim * 3, qkv_bias)
        self.proj = _Lin(dim, dim, True)
        self.use_rel_pos = use_rel_pos
        if use_rel_pos:
            self.rel_pos_h = nn.Parameter(torch.zeros(L, dim // heads))
            self.rel_pos_w = nn.Parameter(torch.zeros(L, dim // heads))


class _SamMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.lin1, self.lin2 = _Lin(dim, hidden), _Lin(hidden, dim)


class _SamBlock(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, qkv_bias, use_rel_pos, window, grid):
        super().__init__()
        self.window_size = window
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        L = (4 * grid - 1) if window == 0 else (2 * window - 1)   # sam_vit.py:255-267 (global tables are 4*size-1)
        self.attn = _SamAttn(dim, heads, qkv_bias, use_rel_pos, L)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _SamMlp(dim, int(dim * mlp_ratio))


@MODELS.register_module()
class SAMViT(nn.Module):
    def __init__(self, img_size=1024, out_indices=(3, 5, 7, 11), patch_size=16, in_chans=3, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=True, norm_layer=None, act_layer=None, use_abs_pos=True,
                 use_rel_pos=False, rel_pos_zero_init=True, window_size=0, global_attn_indexes=(), init_cfg=None, **kw):
        super().__init__()
        if not (use_abs_pos and use_rel_pos):
            raise NotImplementedError("HIP path implements the reference's SAM config (abs pos + decomposed rel pos)")
        self.img_size, self.patch_size, self.embed_dim, self.num_heads = img_size, patch_size, embed_dim, num_heads
        self.out_indices = list(out_indices)
        self.window_size, self.global_attn_indexes = window_size, list(global_attn_indexes)
        self.patch_embed = _PatchEmbed(patch_size, in_chans, embed_dim)
        g = img_size // patch_size
        self.pos_embed = nn.Parameter(torch.zeros(1, g, g, embed_dim))
        self.blocks = nn.ModuleList([
            _SamBlock(embed_dim, num_heads, mlp_ratio, qkv_bias, use_rel_pos, 0 if i in self.global_attn_indexes else window_size, g)
            for i in range(depth)])
        self._engine = None
        self.register_load_state_dict_post_hook(lambda m, keys: m.engine().invalidate())

    def engine(self):
        if self._engine is None:
            self._engine = SamEngine(self)
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


class SamEngine:
    def __init__(self, vit):
        self.vit = vit
        self._packed = None

    def invalidate(self):
        self._packed = None

    def lora_on(self):
        return isinstance(self.vit.blocks[0].attn.qkv, LoraLinear)

    def trainable(self):
        out = []
        if self.lora_on():
            for blk in self.vit.blocks:
                out += [blk.attn.qkv.lora_A["default"].weight, blk.attn.qkv.lora_B["default"].weight]
        return out

    def packed(self):
        cd = compute_dtype()
        dev = self.vit.pos_embed.device
        if self._packed is not None and self._packed["cd"] == cd and self._packed["dev"] == dev:
            return self._packed
        v = self.vit
        D, H = v.embed_dim, v.num_heads
        d = D // H
        G = v.img_size // v.patch_size
        P = dict(cd=cd, dev=dev, layers=[])
        with torch.no_grad():
            P["pe"] = Packed(v.patch_embed.proj.weight.detach().reshape(D, -1), cd)
            P["pe_b"] = v.patch_embed.proj.bias.detach().float().contiguous()
            P["pos"] = v.pos_embed.detach().reshape(G * G, D).float().contiguous()
            for blk in v.blocks:
                qkv = blk.attn.qkv
                base = qkv.base_layer if isinstance(qkv, LoraLinear) else qkv
                kq = D + (R_PAD if isinstance(qkv, LoraLinear) else 0)
                S = blk.window_size if blk.window_size > 0 else G
                rh = torch.empty(S, S, d, dtype=torch.float32, device=dev)
                rw = torch.empty(S, S, d, dtype=torch.float32, device=dev)
                ops.sam_relpos_table(blk.attn.rel_pos_h.detach().float().contiguous(), S, rh)   # frozen -> once
                ops.sam_relpos_table(blk.attn.rel_pos_w.detach().float().contiguous(), S, rw)
                # relative-index tables of the flash forward: tbl[j] = Rh[qh, kh] for qh - kh + S - 1 = j (rows >= 2S-1 zero)
                JP = 2 * (16 if S <= 16 else 32)
                idx_q = torch.tensor([max(j - (S - 1), 0) for j in range(2 * S - 1)], device=dev)
                idx_k = torch.tensor([max(S - 1 - j, 0) for j in range(2 * S - 1)], device=dev)
                tbl = []
                for r_ in (rh, rw):
                    t_ = torch.zeros(max(JP, 2 * S - 1), d, dtype=torch.float32, device=dev)
                    t_[:2 * S - 1] = r_[idx_q, idx_k]
                    tb = torch.zeros(t_.shape, dtype=cd, device=dev)
                    ops.cast(t_, tb)
                    tbl.append(tb)
                Lp = dict(
                    S=S, rh=rh, rw=rw, tbl_h=tbl[0], tbl_w=tbl[1],
                    qkv=Packed(base.weight.detach(), cd, k_pad=kq), qkv_b=base.bias.detach().float().contiguous(),
                    proj=Packed(blk.attn.proj.weight.detach(), cd), proj_b=blk.attn.proj.bias.detach().float().contiguous(),
                    fc1=Packed(blk.mlp.lin1.weight.detach(), cd), fc1_b=blk.mlp.lin1.bias.detach().float().contiguous(),
                    fc2=Packed(blk.mlp.lin2.weight.detach(), cd), fc2_b=blk.mlp.lin2.bias.detach().float().contiguous(),
                    n1w=blk.norm1.weight.detach().float().contiguous(), n1b=blk.norm1.bias.detach().float().contiguous(),
                    n2w=blk.norm2.weight.detach().float().contiguous(), n2b=blk.norm2.bias.detach().float().contiguous(),
                )
                if isinstance(qkv, LoraLinear):
                    Lp["a"] = torch.zeros(R_PAD, D, dtype=cd, device=dev)
                    Lp["at"] = torch.zeros(D, R_PAD, dtype=cd, device=dev)
                P["layers"].append(Lp)
        self._packed = P
        return P

    def refresh_lora(self, P):
        D = self.vit.embed_dim
        sites = []
        for blk, Lp in zip(self.vit.blocks, P["layers"]):
            q = blk.attn.qkv
            if isinstance(q, LoraLinear):
                A, Bm = q.lora_A["default"].weight.detach(), q.lora_B["default"].weight.detach()
                sites.append((A, Bm, Lp["a"], Lp["at"], Lp["qkv"].w, Lp["qkv"].wt, q.r, A.shape[1], Bm.shape[0], D))
        _refresh_sites(P, sites)

    def attention(self, qkv, Lp, nimg, G, H, d, cd, dev, keep=False):
        """-> token-major attention output [nimg*G*G, H*d] (+ the probabilities [nb, NP, NP] when keep=True: rows / columns
        beyond the S*S window tokens are zero, so P can later serve in place as a [K, N] GEMM operand)"""
        S = Lp["S"]
        if (is_half(cd) and d == 80 and (S == 14 or (S == 32 and G == 32))
                and os.environ.get("VFMSEG_SAM_FLASH", "1") != "0"):
            # one flash-style launch, no score matrix (csrc/sam_flash.hip); training keeps lse + the bias columns for sam_flash_bwd.hip
            ao = torch.empty(nimg * G * G, H * d, dtype=cd, device=dev)
            if keep and os.environ.get("VFMSEG_SAM_FLASH_BWD", "1") != "0":
                lse, qext = ops.sam_attn_flash_stats(nimg, G, S, H, dev)
                ops.sam_attn_flash_fwd_train(qkv, Lp["qkv_b"], Lp["tbl_h"], Lp["tbl_w"], ao, lse, qext, nimg, G, S, H, d, d ** -0.5)
                return ao, dict(ao=ao, lse=lse, qext=qext)
            if not keep:
                ops.sam_attn_flash_fwd(qkv, Lp["qkv_b"], Lp["tbl_h"], Lp["tbl_w"], ao, nimg, G, S, H, d, d ** -0.5)
                return ao
        nws = (G + S - 1) // S
        nb, Nw = nimg * nws * nws * H, S * S
        Dq, NP = _pad64(d + 2 * S), _pad64(Nw)
        qa = torch.empty(nb, Nw, Dq, dtype=cd, device=dev)
        ka = torch.empty(nb, Nw, Dq, dtype=cd, device=dev)
        vw = torch.zeros(nb, NP, d, dtype=cd, device=dev)
        ops.sam_attn_prep(qkv, Lp["qkv_b"], Lp["rh"], Lp["rw"], qa, ka, vw, nimg, G, S, H, d, d ** -0.5)
        if not keep:
            sc = torch.empty(nb, Nw, Nw, dtype=torch.float32, device=dev)
            ops.gemm(qa, ka, sc)                                    # batched: scores incl. the decomposed rel-pos bias
            pr = torch.empty(nb, Nw, NP, dtype=cd, device=dev)
            ops.softmax_rows(sc.view(nb * Nw, Nw), pr.view(nb * Nw, NP), Nw)
            pa = pr
        else:
            sc = torch.empty(nb, NP, NP, dtype=torch.float32, device=dev)
            ops.gemm(qa, ka, sc[:, :Nw, :Nw])
            pr = torch.empty(nb, NP, NP, dtype=cd, device=dev)
            ops.softmax_rows_batched(sc.view(nb * NP, NP), pr.view(nb * NP, NP), Nw, NP, Nw)
            pa = pr[:, :Nw]
        ow = torch.empty(nb, NP, d, dtype=cd, device=dev)
        ops.gemm(pa, vw, ow[:, :Nw], trans_b=True)                  # P @ V, V consumed in place as the [K, N] operand
        ao = torch.empty(nimg * G * G, H * d, dtype=cd, device=dev)
        ops.sam_attn_merge(ow, ao, nimg, G, S, H, d)
        return (ao, pr) if keep else ao

    def attention_bwd(self, dao, qkv, pr, Lp, nimg, G, H, d, cd, dev):
        """d(attention output) [M, H*d] -> dqkv [M, 3*H*d]; pr = the probabilities kept by attention(keep=True)."""
        S = Lp["S"]
        if isinstance(pr, dict):   # flash training forward
            dqkv = torch.empty(nimg * G * G, 3 * H * d, dtype=cd, device=dev)
            return ops.sam_attn_flash_bwd(qkv, Lp["qkv_b"], Lp["tbl_h"], Lp["tbl_w"], pr["ao"], dao, pr["lse"], pr["qext"], dqkv,
                                          nimg, G, S, H, d, d ** -0.5)
        nws = (G + S - 1) // S
        nb, Nw = nimg * nws * nws * H, S * S
        Dq, NP, dp = _pad64(d + 2 * S), _pad64(Nw), _pad64(d)
        scale = d ** -0.5
        dow = torch.empty(nb, NP, dp, dtype=cd, device=dev)
        vp = torch.empty(nb, NP, dp, dtype=cd, device=dev)
        dowT = torch.empty(nb, dp, NP, dtype=cd, device=dev)
        qsT = torch.empty(nb, dp, NP, dtype=cd, device=dev)
        ops.sam_attn_bwd_prep(dao, qkv, Lp["qkv_b"], dow, dowT, vp, qsT, nimg, G, S, H, d, scale)
        dP = torch.empty(nb, NP, NP, dtype=torch.float32, device=dev)
        ops.gemm(dow[:, :Nw], vp, dP[:, :Nw])                       # dP = dO V^T (rows of padded tokens are never read)
        dS = torch.empty(nb, NP, NP, dtype=cd, device=dev)
        ops.softmax_rows_bwd(pr.view(nb * NP, NP), dP.view(nb * NP, NP), dS.view(nb * NP, NP), Nw, NP, Nw)
        del dP
        dvT = torch.empty(nb, dp, NP, dtype=cd, device=dev)
        dkT = torch.empty(nb, dp, NP, dtype=cd, device=dev)
        ops.gemm(dowT, pr, dvT, trans_b=True)                       # dV^T = dO^T P
        ops.gemm(qsT, dS, dkT, trans_b=True)                        # dK^T = (scale q)^T dS
        qa = torch.empty(nb, NP, Dq, dtype=cd, device=dev)          # recomputed augmented operands, NP rows per batch
        ka = torch.zeros(nb, NP, Dq, dtype=cd, device=dev)
        vw = torch.empty(nb, NP, d, dtype=cd, device=dev)
        ops.sam_attn_prep(qkv, Lp["qkv_b"], Lp["rh"], Lp["rw"], qa, ka, vw, nimg, G, S, H, d, scale)
        dqa = torch.empty(nb, NP, Dq, dtype=cd, device=dev)
        ops.gemm(dS, ka, dqa, trans_b=True)                         # dQaug = dS Kaug
        dqkv = torch.empty(nimg * G * G, 3 * H * d, dtype=cd, device=dev)
        ops.sam_attn_bwd_merge(dqa, dkT, dvT, Lp["rh"], Lp["rw"], dqkv, nimg, G, S, H, d, scale)
        return dqkv

    def forward(self, jobs, training, seed):
        v, P = self.vit, self.packed()
        cd, dev = P["cd"], P["dev"]
        D, H, ps = v.embed_dim, v.num_heads, v.patch_size
        d = D // H
        G = v.img_size // ps
        lora = self.lora_on()
        if lora:
            self.refresh_lora(P)
        for img, box in jobs:
            y0, y1, x0, x1 = box if box is not None else (0, img.shape[2], 0, img.shape[3])
            if (y1 - y0, x1 - x0) != (v.img_size, v.img_size):
                raise ValueError(f"SAMViT has a fixed abs pos-embed: input must be {v.img_size}x{v.img_size} (sam_vit.py:131-132)")
        nimg = sum(j[0].shape[0] for j in jobs)
        M = nimg * G * G
        A0 = torch.empty(M, 3 * ps * ps, dtype=cd, device=dev)
        r0 = 0
        for img, box in jobs:
            b = img.shape[0]
            ops.patchify(img, A0[r0 * G * G:(r0 + b) * G * G], box=box, patch=ps)
            r0 += b
        x = torch.empty(M, D, dtype=torch.float32, device=dev)
        posb = P["pos"].repeat(nimg, 1) if nimg > 1 else P["pos"]      # pos-embed rows per image (plumbing copy)
        P["pe"].fwd(A0, x, bias=P["pe_b"], residual=posb)
        nt = len(v.out_indices)
        xcat = torch.empty(M, nt * D, dtype=cd, device=dev)
        keep = bool(training) and lora            # activations for the hand-written backward
        saved = []
        from .functional import draw_seed
        seed, rng0 = draw_seed(seed, len(v.blocks) * M * D) if keep else (0, 0)
        for li, (blk, Lp) in enumerate(zip(v.blocks, P["layers"])):
            kq = Lp["qkv"].k
            S_ = {"x_in": x}
            a1 = torch.empty(M, kq, dtype=cd, device=dev)
            st1 = torch.empty(M, 2, dtype=torch.float32, device=dev) if keep else None
            xd = mask = None
            q = blk.attn.qkv if lora else None
            if keep and q.p > 0 and is_half(cd) and D % 256 == 0:
                mask = torch.empty(M, D, dtype=cd, device=dev)
                xd = torch.empty(M, D, dtype=cd, device=dev)
                ops.layernorm_dropout_fwd(x, Lp["n1w"], Lp["n1b"], 1e-6, a1[:, :D], st1, xd, mask, q.p, seed, offset=rng0 + li * M * D)
            else:
                ops.layernorm_fwd(x, Lp["n1w"], Lp["n1b"], 1e-6, a1[:, :D], st1)
                if keep and q.p > 0:
                    mask = torch.empty(M, D, dtype=cd, device=dev)
                    ops.dropout_mask(mask, q.p, seed, offset=rng0 + li * M * D)
                    xd = torch.empty(M, D, dtype=cd, device=dev)
                    ops.mul_mask(a1[:, :D], mask, xd)
            if kq > D:
                ops.gemm(xd if xd is not None else a1[:, :D], Lp["a"], a1[:, D:D + R_PAD], alpha=blk.attn.qkv.scaling)
            qkv = torch.empty(M, 3 * D, dtype=cd, device=dev)
            Lp["qkv"].fwd(a1, qkv, bias=Lp["qkv_b"])
            if keep:
                ao, pr = self.attention(qkv, Lp, nimg, G, H, d, cd, dev, keep=True)
            else:
                ao, pr = self.attention(qkv, Lp, nimg, G, H, d, cd, dev), None
            xm = torch.empty(M, D, dtype=torch.float32, device=dev)
            Lp["proj"].fwd(ao, xm, bias=Lp["proj_b"], residual=x)
            a2 = torch.empty(M, D, dtype=cd, device=dev)
            st2 = torch.empty(M, 2, dtype=torch.float32, device=dev) if keep else None
            ops.layernorm_fwd(xm, Lp["n2w"], Lp["n2b"], 1e-6, a2, st2)
            hid = Lp["fc1"].n
            g = ops.empty_ld(M, hid, cd, dev)
            hpre = torch.empty(M, hid, dtype=cd, device=dev) if keep else None
            # training also saves gelu'(pre-activation): what fc2's dgrad multiplies by
            Lp["fc1"].fwd(a2, g, bias=Lp["fc1_b"], ep_mode=ops.EP_GELU_DGELU if keep else ops.EP_GELU, c2=hpre)
            xo = torch.empty(M, D, dtype=torch.float32, device=dev)
            Lp["fc2"].fwd(g, xo, bias=Lp["fc2_b"], residual=xm)
            if keep:
                S_.update(a1=a1, st1=st1, qkv=qkv, pr=pr, x_mid=xm, st2=st2, hpre=hpre, xd=xd, mask=mask)
                saved.append(S_)
            x = xo
            for i, oi in enumerate(v.out_indices):   # (an index may be listed more than once: every copy is a tap of its own)
                if oi == li:
                    ops.cast(x, xcat[:, i * D:(i + 1) * D])
        return xcat, (G, G), dict(saved=saved if keep else None, nimg=nimg, M=M, P=P, G=G)

    # ---- backward: d(xcat) -> LoRA grads [dA0, dB0, dA1, dB1, ...]
    def backward(self, ctx, dxcat):
        from .functional import direct_grad_target
        v, P = self.vit, ctx["P"]
        cd, dev = P["cd"], P["dev"]
        D, H = v.embed_dim, v.num_heads
        d = D // H
        M, nimg, G = ctx["M"], ctx["nimg"], ctx["G"]
        dx = torch.zeros(M, D, dtype=torch.float32, device=dev)
        grads = [None] * (2 * len(v.blocks))
        t = torch.empty(M, D, dtype=cd, device=dev)
        for li in range(len(v.blocks) - 1, -1, -1):
            blk, Lp, S_ = v.blocks[li], P["layers"][li], ctx["saved"][li]
            q = blk.attn.qkv
            for i, oi in enumerate(v.out_indices):
                if oi == li:
                    src = dxcat[:, i * D:(i + 1) * D]
                    ops.strided_copy(src, dx, (M, D), (src.stride(0), 1), (D, 1), accumulate=True)
            # ---- MLP branch: x_out = x_mid + lin2(gelu(lin1(LN2(x_mid))))
            ops.cast(dx, t)
            hid = Lp["fc1"].n
            dh = ops.empty_ld(M, hid, cd, dev)
            Lp["fc2"].dgrad(t, dh, ep_mode=ops.EP_MUL, aux=S_["hpre"])
            dn = torch.empty(M, D, dtype=cd, device=dev)
            Lp["fc1"].dgrad(dh, dn)
            ops.layernorm_bwd(dn, S_["x_mid"], Lp["n2w"], S_["st2"], dx, accumulate_dx=True)
            del dh, dn
            # ---- attention branch: x_mid = x_in + proj(attn(qkv(LN1(x_in))))
            ops.cast(dx, t)
            dao = torch.empty(M, D, dtype=cd, device=dev)
            Lp["proj"].dgrad(t, dao)
            dqkv = self.attention_bwd(dao, S_["qkv"], S_["pr"], Lp, nimg, G, H, d, cd, dev)
            kq = Lp["qkv"].k
            da1 = torch.empty(M, kq, dtype=cd, device=dev)
            Lp["qkv"].dgrad(dqkv, da1)
            r = q.r
            A, Bm = q.lora_A["default"].weight, q.lora_B["default"].weight
            a1 = S_["a1"]
            xd = S_["xd"] if S_["xd"] is not None else a1[:, :D]
            tB, tA = direct_grad_target(Bm), direct_grad_target(A)
            gBt = torch.empty(R_PAD, Bm.shape[0], dtype=torch.float32, device=dev)
            gAp = torch.empty(R_PAD, A.shape[1], dtype=torch.float32, device=dev)
            doneB = _wgrad_small_t(a1[:, D:D + R_PAD], dqkv, gBt, scatter=None if tB is None else (tB, r, 1, r))
            doneA = _wgrad_small_t(da1[:, D:D + R_PAD], xd, gAp, alpha=q.scaling, scatter=None if tA is None else (tA, r, A.shape[1], 1))
            if doneB is not True:
                if tB is not None:
                    ops.strided_copy(gBt, tB, (Bm.shape[0], r), (1, gBt.stride(0)), (r, 1), accumulate=True)
                else:
                    gB = torch.empty_like(Bm, dtype=torch.float32)
                    ops.strided_copy(gBt, gB, (Bm.shape[0], r), (1, gBt.stride(0)), (r, 1))
                    grads[2 * li + 1] = gB
            if doneA is not True:
                if tA is not None:
                    ops.axpby(gAp[:r].reshape(-1), 1.0, tA.view(-1), 1.0)
                else:
                    grads[2 * li] = gAp[:r]
            ep = dict(ep_mode=ops.EP_MUL, aux=S_["mask"]) if S_["mask"] is not None else {}
            ops.gemm(da1[:, D:D + R_PAD], Lp["at"], da1[:, :D], alpha=q.scaling, residual=da1[:, :D], **ep)
            ops.layernorm_bwd(da1[:, :D], S_["x_in"], Lp["n1w"], S_["st1"], dx, accumulate_dx=True)
            ctx["saved"][li] = None
            if BACKWARD_EVENTS["block_done"] is not None:
                BACKWARD_EVENTS["block_done"](li)
        return grads
