"""SAM ViT-H backbone on the HIP kernels, inference path (reference: rein/models/backbones/sam_vit.py:51-465) -
BASELINE config 5 (lora_sam_linear.py: EncoderDecoder + LoRA(qkv) SAM-H + LinearHead, 'slide' test mode).

NHWC tokens without a cls token: the residual stream is the matrix [n*G*G, 1280] (a multiple of 128 rows, no GEMM tail).
28 of the 32 blocks attend inside 14x14 windows of the zero-padded 42x42 grid (padded tokens are real keys whose
k/v equal the projection bias, as in the reference), 4 blocks attend globally; every block adds the decomposed
relative-position bias.  The bias is folded into augmented Q/K operands (see csrc/sam.hip), which turns the biased
attention into batched MFMA GEMMs + a row softmax; LoRA is merged into the QKV GEMM by K-concatenation as for DINOv2.
Training (backward through this attention form) is a later-round item: forward_tokens(training=True) raises.
"""
import torch
import torch.nn as nn

from . import ops
from .backbones import R_PAD, LoraLinear, Packed, _BackboneFn, _Lin, _PatchEmbed, _pack_at
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

    def forward_tokens(self, jobs, training=False, seed=0):
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
                Lp = dict(
                    S=S, rh=rh, rw=rw,
                    qkv=Packed(base.weight.detach(), cd, k_pad=kq), qkv_b=base.bias.detach().float().contiguous(),
                    proj=Packed(blk.attn.proj.weight.detach(), cd), proj_b=blk.attn.proj.bias.detach().float().contiguous(),
                    fc1=Packed(blk.mlp.lin1.weight.detach(), cd), fc1_b=blk.mlp.lin1.bias.detach().float().contiguous(),
                    fc2=Packed(blk.mlp.lin2.weight.detach(), cd), fc2_b=blk.mlp.lin2.bias.detach().float().contiguous(),
                    n1w=blk.norm1.weight.detach().float().contiguous(), n1b=blk.norm1.bias.detach().float().contiguous(),
                    n2w=blk.norm2.weight.detach().float().contiguous(), n2b=blk.norm2.bias.detach().float().contiguous(),
                )
                if isinstance(qkv, LoraLinear):
                    Lp["a"] = torch.zeros(R_PAD, D, dtype=cd, device=dev)
                P["layers"].append(Lp)
        self._packed = P
        return P

    def refresh_lora(self, P):
        D = self.vit.embed_dim
        with torch.no_grad():
            for blk, Lp in zip(self.vit.blocks, P["layers"]):
                q = blk.attn.qkv
                if isinstance(q, LoraLinear):
                    ops.cast(q.lora_A["default"].weight.detach(), Lp["a"][:q.r])
                    ops.cast(q.lora_B["default"].weight.detach(), Lp["qkv"].w[:, D:D + q.r])

    def attention(self, qkv, Lp, nimg, G, H, d, cd, dev):
        """-> token-major attention output [nimg*G*G, H*d]"""
        S = Lp["S"]
        nws = (G + S - 1) // S
        nb, Nw = nimg * nws * nws * H, S * S
        Dq, NP = _pad64(d + 2 * S), _pad64(Nw)
        qa = torch.empty(nb, Nw, Dq, dtype=cd, device=dev)
        ka = torch.empty(nb, Nw, Dq, dtype=cd, device=dev)
        vw = torch.zeros(nb, NP, d, dtype=cd, device=dev)
        ops.sam_attn_prep(qkv, Lp["qkv_b"], Lp["rh"], Lp["rw"], qa, ka, vw, nimg, G, S, H, d, d ** -0.5)
        sc = torch.empty(nb, Nw, Nw, dtype=torch.float32, device=dev)
        ops.gemm(qa, ka, sc)                                        # batched: scores incl. the decomposed rel-pos bias
        pr = torch.empty(nb, Nw, NP, dtype=cd, device=dev)
        ops.softmax_rows(sc.view(nb * Nw, Nw), pr.view(nb * Nw, NP), Nw)
        ow = torch.empty(nb, NP, d, dtype=cd, device=dev)
        ops.gemm(pr, vw, ow[:, :Nw], trans_b=True)                  # P @ V, V consumed in place as the [K, N] operand
        ao = torch.empty(nimg * G * G, H * d, dtype=cd, device=dev)
        ops.sam_attn_merge(ow, ao, nimg, G, S, H, d)
        return ao

    def forward(self, jobs, training, seed):
        if training:
            raise NotImplementedError("SAM backward is not on the HIP path yet (round 1 ships SAM inference, config 5)")
        v, P = self.vit, self.packed()
        cd, dev = P["cd"], P["dev"]
        D, H, ps = v.embed_dim, v.num_heads, v.patch_size
        d = D // H
        G = v.img_size // ps
        if self.lora_on():
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
        for li, (blk, Lp) in enumerate(zip(v.blocks, P["layers"])):
            kq = Lp["qkv"].k
            a1 = torch.empty(M, kq, dtype=cd, device=dev)
            ops.layernorm_fwd(x, Lp["n1w"], Lp["n1b"], 1e-6, a1[:, :D], None)
            if kq > D:
                ops.gemm(a1[:, :D], Lp["a"], a1[:, D:D + R_PAD], alpha=blk.attn.qkv.scaling)
            qkv = torch.empty(M, 3 * D, dtype=cd, device=dev)
            Lp["qkv"].fwd(a1, qkv, bias=Lp["qkv_b"])
            ao = self.attention(qkv, Lp, nimg, G, H, d, cd, dev)
            xm = torch.empty(M, D, dtype=torch.float32, device=dev)
            Lp["proj"].fwd(ao, xm, bias=Lp["proj_b"], residual=x)
            a2 = torch.empty(M, D, dtype=cd, device=dev)
            ops.layernorm_fwd(xm, Lp["n2w"], Lp["n2b"], 1e-6, a2, None)
            g = torch.empty(M, Lp["fc1"].n, dtype=cd, device=dev)
            Lp["fc1"].fwd(a2, g, bias=Lp["fc1_b"], ep_mode=ops.EP_GELU)
            xo = torch.empty(M, D, dtype=torch.float32, device=dev)
            Lp["fc2"].fwd(g, xo, bias=Lp["fc2_b"], residual=xm)
            x = xo
            if li in v.out_indices:
                i = v.out_indices.index(li)
                ops.cast(x, xcat[:, i * D:(i + 1) * D])
        return xcat, (G, G), dict(saved=None)

    def backward(self, ctx, dxcat):
        raise NotImplementedError("SAM backward is not on the HIP path yet")
