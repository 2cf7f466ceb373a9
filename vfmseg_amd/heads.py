"""Decode heads on the HIP kernels: LinearHead (rein/models/heads/linear_head.py:13-113), VFMHead
(rein/models/heads/VFMHead.py:12-133) and the (Mask)TransformerDecoder (rein/models/heads/Transformer.py:95-283).

Everything is a per-pixel GEMM on token-major maps.  ConvTranspose2d(k=2,s=2) is a GEMM to 4*Cout columns whose
output, read as [4*rows, Cout], is the up-sampled map in a 2x2-blocked pixel order; Conv2d(k=2,s=2) on a blocked map
is a GEMM on a free view.  Only the tiny 19-channel logits are ever re-ordered to raster order (vfm_unblock).
"""
import torch
import torch.nn as nn

from . import functional as Fh
from . import ops
from .precision import compute_dtype, is_half
from .registry import MODELS


class FeatPack:
    """The four backbone taps as one token-major matrix [B*hp*wp, 4*C] (compute dtype)."""

    def __init__(self, xcat, B, hp, wp):
        self.xcat, self.B, self.hp, self.wp = xcat, B, hp, wp

    def taps_nchw(self):
        n = 4
        c = self.xcat.shape[1] // n
        v = self.xcat.view(self.B, self.hp, self.wp, n, c)
        return tuple(v[:, :, :, i].permute(0, 3, 1, 2) for i in range(n))


def as_featpack(inputs, in_index=(0, 1, 2, 3)):
    """Accept the reference API (tuple of NCHW maps) as well as a FeatPack; the tuple path copies (no autograd)."""
    if isinstance(inputs, FeatPack):
        return inputs
    feats = [inputs[i] for i in in_index]
    B, C, h, w = feats[0].shape
    cd = compute_dtype()
    xcat = torch.empty(B * h * w, len(feats) * C, dtype=cd, device=feats[0].device)
    for i, f in enumerate(feats):
        f = f.detach()
        dst = xcat[:, i * C:(i + 1) * C]
        # index space (b, y, x, c): src strides of the NCHW(-shaped) tensor, dst token-major
        ops.strided_copy(f, dst, (B, h, w, C), (f.stride(0), f.stride(2), f.stride(3), f.stride(1)),
                         (h * w * dst.stride(0), w * dst.stride(0), dst.stride(0), 1))
    return FeatPack(xcat, B, h, w)


@MODELS.register_module()
class CrossEntropyLoss(nn.Module):
    """mmseg CrossEntropyLoss(use_sigmoid=False, avg_non_ignore=False): mean over ALL pixels, ignored ones add 0."""

    def __init__(self, use_sigmoid=False, loss_weight=1.0, loss_name="loss_ce", reduction="mean", class_weight=None,
                 avg_non_ignore=False, **kw):
        super().__init__()
        if use_sigmoid or class_weight is not None or avg_non_ignore or reduction != "mean":
            raise NotImplementedError("HIP path implements the reference configs' softmax CE (mean over all pixels)")
        self.loss_weight, self._loss_name = loss_weight, loss_name

    @property
    def loss_name(self):
        return self._loss_name

    def forward(self, cls_score, label, weight=None, ignore_index=255, **kw):
        """cls_score NCHW fp32 at label resolution (API parity); the heads use the fused low-res path instead."""
        lg = torch.empty(cls_score.shape[0], cls_score.shape[2], cls_score.shape[3], cls_score.shape[1],
                         dtype=torch.float32, device=cls_score.device)
        ops.permute_copy(cls_score.detach(), (0, 2, 3, 1), lg)
        loss, _ = Fh.UpsampleCEFn.apply(lg, label.contiguous(), ignore_index, self.loss_weight)
        return loss


class BaseDecodeHead(nn.Module):
    """The slice of mmseg 1.2.2 BaseDecodeHead the reference heads use."""

    def __init__(self, in_channels, channels, *, num_classes=None, out_channels=None, threshold=None, dropout_ratio=0.1,
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), in_index=-1, input_transform=None,
                 loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0), ignore_index=255,
                 sampler=None, align_corners=False, init_cfg=None):
        super().__init__()
        self.in_channels, self.channels, self.in_index = in_channels, channels, in_index
        self.input_transform = input_transform
        self.dropout_ratio, self.norm_cfg, self.act_cfg = dropout_ratio, norm_cfg, act_cfg
        self.ignore_index, self.align_corners = ignore_index, align_corners
        self.num_classes = num_classes
        self.out_channels = out_channels or num_classes
        if align_corners:
            raise NotImplementedError("align_corners=True is not on the HIP path (all reference configs use False)")
        self.loss_decode = MODELS.build(loss_decode)
        self.sampler = None
        self.conv_seg = nn.Conv2d(channels, self.out_channels, kernel_size=1)
        nn.init.normal_(self.conv_seg.weight, std=0.01)
        nn.init.zeros_(self.conv_seg.bias)

    def _stack_batch_gt(self, batch_data_samples):
        return torch.stack([d.gt_sem_seg.data for d in batch_data_samples], dim=0)

    # ---- shared loss path (linear_head.py:72-113 / VFMHead.py:91-133)
    def _loss_from_lowres(self, logits_nhwc, seg_label, return_logits):
        label = seg_label.squeeze(1).contiguous()
        # loss and accuracy (mmseg `accuracy`: 100 * correct / (valid + eps)) come out of the fused kernel pair
        loss, acc = Fh.UpsampleCEFn.apply(logits_nhwc, label, self.ignore_index, self.loss_decode.loss_weight)
        losses = {self.loss_decode.loss_name: loss, "acc_seg": acc}
        if return_logits:
            B, h, w, C = logits_nhwc.shape
            H, W = label.shape[1:]
            full = torch.empty(B, C, H, W, dtype=torch.float32, device=label.device)
            ops.resize_bilinear(logits_nhwc.detach(), False, B, h, w, C, full, 1, (H, W))
            return losses, full
        return losses

    def predict_by_feat(self, seg_logits_nhwc, batch_img_metas):
        m = batch_img_metas[0]
        if isinstance(m["img_shape"], torch.Size):
            size = tuple(m["img_shape"])
        elif "pad_shape" in m:
            size = tuple(m["pad_shape"][:2])
        else:
            size = tuple(m["img_shape"][:2])
        B, h, w, C = seg_logits_nhwc.shape
        out = torch.empty(B, C, size[0], size[1], dtype=torch.float32, device=seg_logits_nhwc.device)
        ops.resize_bilinear(seg_logits_nhwc, False, B, h, w, C, out, 1, size)
        return out


def _fp32_on_fp32_taps(fn):
    """A head handed fp32 feature taps while a 16-bit mode is active runs in fp32 (precision.eval_heads_fp32: the fp16 mode's predictions)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, fp, *a, **k):
        if fp.xcat.dtype == torch.float32 and is_half(compute_dtype()):
            from .precision import compute_as
            import os
            with compute_as(torch.float32, split3=os.environ.get("VFMSEG_FP16_EVAL_HEADS", "fp32") != "f32mfma"):
                return fn(self, fp, *a, **k)
        return fn(self, fp, *a, **k)
    return wrapped


@MODELS.register_module()
class LinearHead(BaseDecodeHead):
    def __init__(self, interpolate_mode="bilinear", **kwargs):
        super().__init__(input_transform="multiple_select", **kwargs)
        c = self.in_channels[0]
        n = len(self.in_channels)
        assert n == len(self.in_index) == 4 and c % 4 == 0
        self._channels = c

        class _ConvModule(nn.Module):  # mmcv ConvModule(norm=GN): conv (no bias) -> gn -> ReLU
            def __init__(s):
                super().__init__()
                s.conv = nn.Conv2d(c * n, c, 1, bias=False)
                s.gn = nn.GroupNorm(self.norm_cfg.get("num_groups", 32), c)

        self.fusion_conv = _ConvModule()
        self.output_upscaling = nn.Sequential(
            nn.ConvTranspose2d(c, c // 2, kernel_size=2, stride=2),
            nn.BatchNorm2d(c // 2),  # nn.SyncBatchNorm in the reference (linear_head.py:44): same parameters/buffers
            nn.GELU(),
            nn.ConvTranspose2d(c // 2, c // 4, kernel_size=2, stride=2),
            nn.GELU(),
        )
        self.bn_sync = None  # callable(tensor) all-reducing in place over the DP group (set by vfmseg_amd.parallel)
        self.bn_world = 1

    @_fp32_on_fp32_taps
    def forward_tokens(self, fp):
        cd = compute_dtype()
        B, P = fp.B, fp.hp * fp.wp
        fc, up = self.fusion_conv, self.output_upscaling
        y = Fh.linear(fp.xcat, fc.conv.weight, "conv1x1", out_dtype=torch.float32)
        y = Fh.group_norm_act(y, fc.gn.weight, fc.gn.bias, B, P, fc.gn.num_groups, fc.gn.eps, ops.ACT_RELU, cd)
        y = Fh.linear(y, up[0].weight, "convT2x2", bias=up[0].bias, bias_tile=4, out_dtype=torch.float32)
        y = y.view(4 * B * P, -1)
        bn = up[1]
        if self.training:
            y = Fh.batch_norm_act_train(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                        ops.ACT_GELU, cd, self.bn_sync, self.bn_world)
            with torch.no_grad():
                bn.num_batches_tracked += 1
        else:
            y = Fh.batch_norm_act_eval(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, ops.ACT_GELU, cd)
        y = Fh.linear(y, up[3].weight, "convT2x2", bias=up[3].bias, bias_tile=4, act="gelu", out_dtype=cd)
        y = y.view(16 * B * P, -1)
        y = Fh.dropout(y, self.dropout_ratio, self.training, rows_per_group=16 * P)  # Dropout2d: per (image, channel)
        lg = Fh.linear(y, self.conv_seg.weight, "conv1x1", bias=self.conv_seg.bias, out_dtype=torch.float32)
        return Fh.UnblockFn.apply(lg.view(B, 16 * P, -1), B, 4 * fp.hp, 4 * fp.wp, 2)  # [B, 4hp, 4wp, classes]

    def forward(self, inputs):
        return self.forward_tokens(as_featpack(inputs, self.in_index)).permute(0, 3, 1, 2)

    def loss(self, inputs, seg_label, return_logits=False):
        lg = self.forward_tokens(as_featpack(inputs, self.in_index))
        return self._loss_from_lowres(lg, seg_label, return_logits)

    def predict(self, inputs, batch_img_metas, test_cfg=None):
        return self.predict_by_feat(self.forward_tokens(as_featpack(inputs, self.in_index)), batch_img_metas)


# ------------------------------------------------------------------------------------------------ transformer decoder
class _CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = heads * dim_head
        context_dim = context_dim or query_dim
        self.heads, self.dim_head, self.p = heads, dim_head, dropout
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))


class _GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class _FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        self.p = dropout
        self.net = nn.Sequential(_GEGLU(dim, dim * mult), nn.Dropout(dropout), nn.Linear(dim * mult, dim))


class BasicTransformerBlock(nn.Module):
    """Transformer.py:158-177: x = attn1(LN1 x) + x ; x = attn2(LN2 x, ctx) + x ; x = ff(LN3 x) + x"""

    def __init__(self, query_dim, n_heads, d_head, dropout=0.0, context_dim=None):
        super().__init__()
        self.attn1 = _CrossAttention(query_dim, None, n_heads, d_head, dropout)
        self.ff = _FeedForward(query_dim, dropout=dropout)
        self.attn2 = _CrossAttention(query_dim, context_dim, n_heads, d_head, dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(query_dim), nn.LayerNorm(query_dim), nn.LayerNorm(query_dim)

    def run(self, x, ctx, B, N, Nk, training):
        cd = compute_dtype()
        a1, a2, ff = self.attn1, self.attn2, self.ff
        H, d = a1.heads, a1.dim_head
        lin = ["linear"]
        h = Fh.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, cd)
        qkv = Fh.linear(h, [a1.to_q.weight, a1.to_k.weight, a1.to_v.weight], lin * 3, out_dtype=cd)
        o = Fh.SelfAttnFn.apply(qkv, B, N, H, d)
        x = Fh.linear(o, a1.to_out[0].weight, "linear", bias=a1.to_out[0].bias, residual=x, out_dtype=torch.float32,
                      drop_p=a1.p if training else 0.0)
        h = Fh.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, cd)
        q = Fh.linear(h, a2.to_q.weight, "linear", out_dtype=cd)
        kv = Fh.linear(ctx, [a2.to_k.weight, a2.to_v.weight], lin * 2, out_dtype=cd)
        o = Fh.CrossAttnFn.apply(q, kv, B, N, Nk, H, d)
        x = Fh.linear(o, a2.to_out[0].weight, "linear", bias=a2.to_out[0].bias, residual=x, out_dtype=torch.float32,
                      drop_p=a2.p if training else 0.0)
        h = Fh.layer_norm(x, self.norm3.weight, self.norm3.bias, self.norm3.eps, cd)
        g = Fh.GegluFn.apply(Fh.linear(h, ff.net[0].proj.weight, "linear", bias=ff.net[0].proj.bias, out_dtype=cd))
        g = Fh.dropout(g, ff.p, training)
        return Fh.linear(g, ff.net[2].weight, "linear", bias=ff.net[2].bias, residual=x, out_dtype=torch.float32)


@MODELS.register_module()
class TransformerDecoder(nn.Module):
    """Transformer.py:228-252."""

    def __init__(self, query_dim, img_feat_dim, n_heads, d_head, depth=1, dropout=0.0):
        super().__init__()
        self.in_channels = query_dim
        self.norm = nn.GroupNorm(32, query_dim, eps=1e-6)  # Normalize() (:91-92)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(query_dim, n_heads, d_head, dropout, img_feat_dim) for _ in range(depth)])

    def run_tokens(self, query_f32, ctx_cd, B, P, Pk):
        x = Fh.group_norm_act(query_f32, self.norm.weight, self.norm.bias, B, P, 32, self.norm.eps, ops.ACT_NONE, torch.float32)
        for blk in self.transformer_blocks:
            x = blk.run(x, ctx_cd, B, P, Pk, self.training)
        return x


@MODELS.register_module()
class MaskTransformerDecoder(TransformerDecoder):
    """Transformer.py:254-283: a random `mask_ratio` of the query positions is replaced by a learned token whenever
    `mask_enable` (also in eval(); only ms_inference stage 1 turns it off)."""

    def __init__(self, mask_ratio, **kwargs):
        super().__init__(**kwargs)
        self.mask_ratio = mask_ratio
        self.mask_token = nn.Parameter(torch.randn(1, self.in_channels, 1, 1))
        self.mask_enable = True
        self.fixed_keep = None  # tests inject the reference's recorded mask here (bool [B,1,h,w])

    def draw_keep(self, B, P, device):
        if self.fixed_keep is not None:
            return self.fixed_keep.reshape(B * P).to(device=device, dtype=torch.uint8).contiguous()
        m = torch.empty(B * P, dtype=torch.float32, device=device)
        seed, off = Fh._next_rng(B * P)
        ops.dropout_mask(m, self.mask_ratio, seed, off)  # > 0 with probability 1 - mask_ratio
        return (m > 0).to(torch.uint8)

    def run_tokens(self, query_f32, ctx_cd, B, P, Pk):
        if self.mask_enable:
            query_f32 = Fh.MaskTokenFn.apply(query_f32, self.draw_keep(B, P, query_f32.device), self.mask_token)
        return super().run_tokens(query_f32, ctx_cd, B, P, Pk)


class CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        if x.dtype == dtype:
            return x
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
        ops.cast(x, out)
        return out

    @staticmethod
    def backward(ctx, dy):
        if dy.dtype == ctx.src:
            return dy, None
        out = torch.empty(dy.shape, dtype=ctx.src, device=dy.device)
        ops.cast(dy.contiguous(), out)
        return out, None


@MODELS.register_module()
class VFMHead(BaseDecodeHead):
    def __init__(self, transformer, interpolate_mode="bilinear", **kwargs):
        super().__init__(input_transform="multiple_select", **kwargs)
        n = len(self.in_channels)
        assert n == len(self.in_index) == 4
        transformer = dict(transformer)
        transformer["img_feat_dim"] = self.channels
        self.query_dim = transformer["query_dim"]
        ch = self.channels
        self.fuse_conv = nn.Sequential(nn.Conv2d(self.in_channels[0] * n, ch, 1), nn.GroupNorm(32, ch), nn.GELU())
        self.seg_logits_embed = nn.Sequential(
            nn.Conv2d(19, ch // 4, kernel_size=2, stride=2), nn.GroupNorm(32, ch // 4), nn.GELU(),  # 19 hard-coded (VFMHead.py:39)
            nn.Conv2d(ch // 4, ch // 2, kernel_size=2, stride=2), nn.GroupNorm(32, ch // 2), nn.GELU(),
            nn.Conv2d(ch // 2, ch, kernel_size=1), nn.GroupNorm(32, ch),
        )
        self.transformer_decoder = MODELS.build(transformer)

    @staticmethod
    def ctx_windows_ok(seg, boxes, hp, wp):
        """Can the context windows be sampled straight out of the full-size coarse map `seg` [Bimg,C,H,W] (forward_tokens ctx_windows=)?
        The window resize (Hc -> 4 hp) and the whole-map resize (H -> H / s) must share the integer scale s and the window origins must
        sit on the coarse grid: then both put the same taps and weights on the same pixels (align_corners=False: src = s (dst + 0.5) - 0.5)."""
        H, W = seg.shape[2:]
        ok = True
        for (y1, y2, x1, x2) in boxes:
            hc, wc = y2 - y1, x2 - x1
            ok = ok and hc % (4 * hp) == 0 and wc % (4 * wp) == 0 and hc // (4 * hp) == wc // (4 * wp)
            if ok:
                s_ = hc // (4 * hp)
                ok = y1 % s_ == 0 and x1 % s_ == 0 and H % s_ == 0 and W % s_ == 0
        return ok and len(boxes) > 0

    @_fp32_on_fp32_taps
    def forward_tokens(self, fp, ctx_nchw, ctx_windows=None):
        """fp: HR feature pack; ctx_nchw: fp32 [B,19,Hc,Wc] coarse logits (no gradient, as in the reference) - or ctx_windows = (seg, boxes):
        the context of pack image j * Bimg + b is window boxes[j] = (y1, y2, x1, x2) of the full-size map seg[b] (see ctx_windows_ok; saves
        the copy of every window into a batch tensor)."""
        cd = compute_dtype()
        B, hp, wp = fp.B, fp.hp, fp.wp
        P = hp * wp
        emb, fc = self.seg_logits_embed, self.fuse_conv
        if ctx_windows is not None:
            seg, boxes = ctx_windows
            seg = seg.detach()
            assert seg.is_contiguous() and seg.dtype == torch.float32 and len(boxes) * seg.shape[0] == B
            Bimg, Cc, H, W = seg.shape
            dev = seg.device
        else:
            ctx_nchw = ctx_nchw.detach().contiguous()
            Cc, Hc, Wc = ctx_nchw.shape[1:]
            dev = ctx_nchw.device
        kpad = 128 if 4 * Cc <= 128 else (4 * Cc + 63) // 64 * 64
        a1 = torch.zeros(B * 4 * P, kpad, dtype=cd, device=dev)
        # bilinear to 4x the feature grid (VFMHead.py:63-67), emitted in 2x2-blocked order: rows = stride-2 conv patches
        if ctx_windows is not None:
            for j, (y1, y2, x1, x2) in enumerate(boxes):
                s_ = (y2 - y1) // (4 * hp)
                ops.resize_bilinear(seg, True, Bimg, H, W, Cc, a1[j * Bimg * 4 * P:(j + 1) * Bimg * 4 * P], 2, (H // s_, W // s_),
                                    window=(y1 // s_, x1 // s_, 4 * hp, 4 * wp), out_ld=kpad)
        else:
            ops.resize_bilinear(ctx_nchw, True, B, Hc, Wc, Cc, a1, 2, (4 * hp, 4 * wp), out_ld=kpad)
        e = Fh.linear(a1, emb[0].weight, "conv2x2s2", bias=emb[0].bias, out_dtype=torch.float32)
        e = Fh.group_norm_act(e, emb[1].weight, emb[1].bias, B, 4 * P, 32, emb[1].eps, ops.ACT_GELU, cd)
        e = Fh.linear(e.view(B * P, -1), emb[3].weight, "conv2x2s2", bias=emb[3].bias, out_dtype=torch.float32)
        e = Fh.group_norm_act(e, emb[4].weight, emb[4].bias, B, P, 32, emb[4].eps, ops.ACT_GELU, cd)
        e = Fh.linear(e, emb[6].weight, "conv1x1", bias=emb[6].bias, out_dtype=torch.float32)
        e = Fh.group_norm_act(e, emb[7].weight, emb[7].bias, B, P, 32, emb[7].eps, ops.ACT_NONE, cd)
        f = Fh.linear(fp.xcat, fc[0].weight, "conv1x1", bias=fc[0].bias, out_dtype=torch.float32)
        f = Fh.group_norm_act(f, fc[1].weight, fc[1].bias, B, P, 32, fc[1].eps, ops.ACT_GELU, torch.float32)
        # NB (VFMHead.py:82 vs Transformer.py:270): query = fused image features, context = logits embedding
        x = self.transformer_decoder.run_tokens(f, e, B, P, P)
        x = CastFn.apply(x, cd)
        x = Fh.dropout(x, self.dropout_ratio, self.training, rows_per_group=P)
        lg = Fh.linear(x, self.conv_seg.weight, "conv1x1", bias=self.conv_seg.bias, out_dtype=torch.float32)
        return lg.view(B, hp, wp, -1)

    def forward(self, inputs, seg_logits, query=None):
        return self.forward_tokens(as_featpack(inputs, self.in_index), seg_logits).permute(0, 3, 1, 2)

    def loss(self, inputs, seg_logits_embed, seg_label, query=None, return_logits=False):
        lg = self.forward_tokens(as_featpack(inputs, self.in_index), seg_logits_embed)
        return self._loss_from_lowres(lg, seg_label, return_logits)
