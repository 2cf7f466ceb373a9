"""TEST INFRASTRUCTURE ONLY - CPU (torch fp32) restatement of the reference's hot path.

Functional style over a flat ``state_dict`` that uses the reference's key scheme
(SURVEY.md §8b), so the same dict feeds this oracle and the HIP product modules.
Pinned against outputs of the reference's own code by tests/test_oracle_golden.py
(fixtures from oracle/gen_golden.py).  Third-party semantics that are not under
/root/reference (mmseg BaseDecodeHead/CrossEntropyLoss/accuracy/EncoderDecoder, mmcv
ConvModule, peft LoRA, SyncBatchNorm) are restated from the pinned upstream versions
 - "pinned by restatement" at those boundaries.

All citations are relative to /root/reference/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BB = "backbone.model.base_model.model."  # LoRABackbone -> PeftModel -> base_model.model (lora_backbone.py:23)
IGNORE = 255


# =============================================================================== backbone
def lora_linear(sd, prefix, x, lora=True, drop_mask=None):
    """peft 0.10 lora.Linear.forward: base(x) + B(A(dropout(x))) * alpha/r (lora_backbone.py:16-23).
    `drop_mask` is the already-scaled dropout multiplier (1/(1-p) or 0) or None (= eval / p=0)."""
    if prefix + "base_layer.weight" in sd:
        y = F.linear(x, sd[prefix + "base_layer.weight"], sd.get(prefix + "base_layer.bias"))
        if lora:
            xa = x if drop_mask is None else x * drop_mask
            a = sd[prefix + "lora_A.default.weight"]
            b = sd[prefix + "lora_B.default.weight"]
            scaling = sd.get("__lora_scaling__", 1.0)  # alpha / r = 32/32 in every reference config
            y = y + F.linear(F.linear(xa, a), b) * scaling
        return y
    return F.linear(x, sd[prefix + "weight"], sd.get(prefix + "bias"))


def dinov2_pos_embed(sd, npatch, h, w, patch=16, p=BB):
    """dino_v2.py:184-215 interpolate_pos_encoding (bicubic, the +0.1 trick, non-square aware)."""
    pos = sd[p + "pos_embed"]
    n = pos.shape[1] - 1
    if npatch == n and w == h:
        return pos
    cls_pos, patch_pos = pos[:, 0], pos[:, 1:]
    dim = pos.shape[-1]
    # NB the reference calls this with (x, w, h) where its "w" is dim 2 of the image (dino_v2.py:218,227)
    w0, h0 = w // patch + 0.1, h // patch + 0.1
    s = int(math.sqrt(n))
    pp = F.interpolate(
        patch_pos.reshape(1, s, s, dim).permute(0, 3, 1, 2),
        scale_factor=(w0 / math.sqrt(n), h0 / math.sqrt(n)),
        mode="bicubic",
    )
    assert int(w0) == pp.shape[-2] and int(h0) == pp.shape[-1]
    pp = pp.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pos.unsqueeze(0), pp), dim=1)


def dinov2_tokens(sd, x, patch=16, p=BB):
    """dino_v2.py:217-228 prepare_tokens_with_masks + patch_embed.py:68-81."""
    b, _, d2, d3 = x.shape
    t = F.conv2d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=patch)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((sd[p + "cls_token"].expand(b, -1, -1), t), dim=1)
    return t + dinov2_pos_embed(sd, t.shape[1] - 1, d3, d2, patch, p)  # (w=d2, h=d3) naming as in the reference


def attention(q, k, v, scale):
    """attention.py:56-66 (plain softmax path) == xformers memory_efficient_attention. q,k,v: [B,H,N,d]."""
    a = (q * scale) @ k.transpose(-2, -1)
    return a.softmax(dim=-1) @ v


def dinov2_block(sd, x, i, heads, lora=True, drop_mask=None, eps=1e-6, p=BB):
    """block.py:89-114 (eval branch: base model is always in eval under LoRABackbone, utils.py:47-52)."""
    q = f"{p}blocks.{i}."
    b, n, c = x.shape
    h = F.layer_norm(x, (c,), sd[q + "norm1.weight"], sd[q + "norm1.bias"], eps)
    qkv = lora_linear(sd, q + "attn.qkv.", h, lora, drop_mask)
    qkv = qkv.reshape(b, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    o = attention(qkv[0], qkv[1], qkv[2], (c // heads) ** -0.5).transpose(1, 2).reshape(b, n, c)
    o = F.linear(o, sd[q + "attn.proj.weight"], sd[q + "attn.proj.bias"])
    x = x + o * sd[q + "ls1.gamma"]
    h = F.layer_norm(x, (c,), sd[q + "norm2.weight"], sd[q + "norm2.bias"], eps)
    h = F.gelu(F.linear(h, sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"]))
    h = F.linear(h, sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"])
    return x + h * sd[q + "ls2.gamma"]


def dinov2_forward(sd, x, depth=24, heads=16, out_indices=(7, 11, 15, 23), patch=16, lora=True, drop_masks=None, p=BB):
    """dino_v2.py:252-268 forward_features: taps are pre-final-norm, cls dropped, NCHW."""
    b, _, hh, ww = x.shape
    t = dinov2_tokens(sd, x, patch, p)
    outs = []
    for i in range(depth):
        t = dinov2_block(sd, t, i, heads, lora, None if drop_masks is None else drop_masks[i], p=p)
        if i in out_indices:
            outs.append(t[:, 1:].permute(0, 2, 1).reshape(b, -1, hh // patch, ww // patch).contiguous())
    return outs


# =============================================================================== EVA02 backbone
def eva_rope_tables(half_head_dim=32, pt_seq_len=16, ft_seq_len=32, theta=10000.0):
    """VisionRotaryEmbeddingFast.__init__ (eva_02.py:119-157, freqs_for='lang'): cos/sin tables [ft*ft, 2*half_head_dim];
    first half of the channels rotates with the row index, second half with the column index, pairs interleaved."""
    freqs = 1.0 / (theta ** (torch.arange(0, half_head_dim, 2)[: half_head_dim // 2].float() / half_head_dim))
    t = torch.arange(ft_seq_len) / ft_seq_len * pt_seq_len
    f = torch.einsum("i,f->if", t, freqs).repeat_interleave(2, dim=-1)           # [ft, half]
    fy = f[:, None, :].expand(ft_seq_len, ft_seq_len, -1)
    fx = f[None, :, :].expand(ft_seq_len, ft_seq_len, -1)
    fr = torch.cat((fy, fx), dim=-1).reshape(ft_seq_len * ft_seq_len, -1)
    return fr.cos(), fr.sin()


def rotate_half(x):
    """eva_02.py:54-58: (x0, x1, x2, x3, ...) -> (-x1, x0, -x3, x2, ...)"""
    x1, x2 = x[..., 0::2], x[..., 1::2]
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def eva02_block(sd, x, i, heads, cos, sin, lora=True, p=BB):
    """eva_02.py:479-489 (init_values=None) with Attention.forward :333-379 (subln, rope, xattn) and SwiGLU :235-242.
    q/k/v go through F.linear on the *base* weights, so LoRA on q_proj/k_proj/v_proj never enters the graph (Q1);
    LoRA on attn.proj is live.  LayerNorm eps is nn.LayerNorm's default 1e-5 (the cfg norm_layer is ignored, :719)."""
    q_ = f"{p}blocks.{i}."
    b, n, c = x.shape
    d = c // heads
    h = F.layer_norm(x, (c,), sd[q_ + "norm1.weight"], sd[q_ + "norm1.bias"], 1e-5)
    wkey = lambda nm: sd.get(q_ + f"attn.{nm}.base_layer.weight", sd.get(q_ + f"attn.{nm}.weight"))
    q = F.linear(h, wkey("q_proj"), sd[q_ + "attn.q_bias"])
    k = F.linear(h, wkey("k_proj"))
    v = F.linear(h, wkey("v_proj"), sd[q_ + "attn.v_bias"])
    sp = lambda t: t.reshape(b, n, heads, d).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    rope = lambda t: torch.cat((t[:, :, :1], t[:, :, 1:] * cos + rotate_half(t[:, :, 1:]) * sin), dim=2)
    q, k = rope(q), rope(k)
    o = attention(q, k, v, d ** -0.5).transpose(1, 2).reshape(b, n, c)
    o = lora_linear(sd, q_ + "attn.proj.", o, lora)
    x = x + o
    h = F.layer_norm(x, (c,), sd[q_ + "norm2.weight"], sd[q_ + "norm2.bias"], 1e-5)
    x1 = F.linear(h, sd[q_ + "mlp.w1.weight"], sd[q_ + "mlp.w1.bias"])
    x2 = F.linear(h, sd[q_ + "mlp.w2.weight"], sd[q_ + "mlp.w2.bias"])
    hid = F.silu(x1) * x2
    hid = F.layer_norm(hid, (hid.shape[-1],), sd[q_ + "mlp.ffn_ln.weight"], sd[q_ + "mlp.ffn_ln.bias"], 1e-5)
    return x + F.linear(hid, sd[q_ + "mlp.w3.weight"], sd[q_ + "mlp.w3.bias"])


def eva02_forward(sd, x, depth=24, heads=16, out_indices=(7, 11, 15, 23), patch=16, lora=True, p=BB, rope=None):
    """EVA2.forward_features (eva_02.py:816-849): fixed abs pos-embed (input must be img_size), no final norm."""
    b, _, hh, ww = x.shape
    t = F.conv2d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=patch).flatten(2).transpose(1, 2)
    t = torch.cat((sd[p + "cls_token"].expand(b, -1, -1), t), dim=1) + sd[p + "pos_embed"]
    cos, sin = rope if rope is not None else eva_rope_tables(t.shape[-1] // heads // 2, 16, hh // patch)
    outs = []
    for i in range(depth):
        t = eva02_block(sd, t, i, heads, cos, sin, lora, p)
        if i in out_indices:
            outs.append(t[:, 1:].permute(0, 2, 1).reshape(b, -1, hh // patch, ww // patch).contiguous())
    return outs


# =============================================================================== CLIP backbone
def clip_block(sd, x, i, heads, lora=True, p=BB):
    """ResidualAttentionBlock.forward (clip.py:65-68): x + MHA(ln_1(x)); x + c_proj(QuickGELU(c_fc(ln_2(x)))).
    nn.MultiheadAttention consumes in_proj_weight/bias and out_proj.weight/bias as tensors, so the LoRA adapter peft puts on
    `out_proj` never enters the graph; the adapters on mlp.c_fc / mlp.c_proj are live.  x: [B, N, C]."""
    q_ = f"{p}transformer.resblocks.{i}."
    b, n, c = x.shape
    d = c // heads
    h = F.layer_norm(x, (c,), sd[q_ + "ln_1.weight"], sd[q_ + "ln_1.bias"], 1e-5)
    qkv = F.linear(h, sd[q_ + "attn.in_proj_weight"], sd[q_ + "attn.in_proj_bias"]).reshape(b, n, 3, heads, d).permute(2, 0, 3, 1, 4)
    o = attention(qkv[0], qkv[1], qkv[2], d ** -0.5).transpose(1, 2).reshape(b, n, c)
    ow = q_ + "attn.out_proj." + ("base_layer." if (q_ + "attn.out_proj.base_layer.weight") in sd else "")
    x = x + F.linear(o, sd[ow + "weight"], sd[ow + "bias"])
    h = F.layer_norm(x, (c,), sd[q_ + "ln_2.weight"], sd[q_ + "ln_2.bias"], 1e-5)
    h = lora_linear(sd, q_ + "mlp.c_fc.", h, lora)
    h = h * torch.sigmoid(1.702 * h)
    return x + lora_linear(sd, q_ + "mlp.c_proj.", h, lora)


def clip_forward(sd, x, depth=24, heads=16, out_indices=(7, 11, 15, 23), patch=16, lora=True, p=BB):
    """CLIPVisionTransformer.forward (clip.py:315-348, get_embeddings=False): bias-less conv1, the class embedding is added
    twice (once in the token, once in cls_pos, :318-337), positional embedding bilinearly re-interpolated to the token grid
    each forward, ln_pre, taps without the cls token; the fpn layers built in __init__ are never called."""
    b = x.shape[0]
    t = F.conv2d(x, sd[p + "conv1.weight"], None, stride=patch)
    _, c, hh, ww = t.shape
    t = t.reshape(b, c, -1).permute(0, 2, 1)
    ce = sd[p + "class_embedding"]
    t = torch.cat([ce + torch.zeros(b, 1, c), t], dim=1)
    pos = sd[p + "positional_embedding"]
    ss = int(round((pos.shape[0] - 1) ** 0.5))
    cls_pos = pos[0] + ce
    sp = F.interpolate(pos[1:].reshape(1, ss, ss, c).permute(0, 3, 1, 2), size=(hh, ww), mode="bilinear")
    sp = sp.reshape(1, c, hh * ww).permute(0, 2, 1)
    t = t + torch.cat([cls_pos.reshape(1, 1, c), sp], dim=1)
    t = F.layer_norm(t, (c,), sd[p + "ln_pre.weight"], sd[p + "ln_pre.bias"], 1e-5)
    outs = []
    for i in range(depth):
        t = clip_block(sd, t, i, heads, lora, p)
        if i in out_indices:
            outs.append(t[:, 1:].permute(0, 2, 1).reshape(b, -1, hh, ww).contiguous())
    return outs


# =============================================================================== SAM backbone
def sam_rel_pos(q_size, k_size, rel_pos):
    """sam_vit.py:359-389 get_rel_pos: (optionally linearly re-interpolated) table gathered at q - k + (k_size-1)."""
    max_rel = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel, mode="linear")
        r = r.reshape(-1, max_rel).permute(1, 0)
    else:
        r = rel_pos
    qc = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    kc = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    return r[((qc - kc) + (k_size - 1) * max(q_size / k_size, 1.0)).long()]


def sam_attention(sd, q_, x, heads, lora=True):
    """sam_vit.py:273-298 + add_decomposed_rel_pos :392-430. x: [B', H, W, C] (a window batch or the whole map)."""
    b, hh, ww, c = x.shape
    d = c // heads
    qkv = lora_linear(sd, q_ + "attn.qkv.", x, lora).reshape(b, hh * ww, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, b * heads, hh * ww, d).unbind(0)
    attn = (q * d ** -0.5) @ k.transpose(-2, -1)
    rh = sam_rel_pos(hh, hh, sd[q_ + "attn.rel_pos_h"])
    rw = sam_rel_pos(ww, ww, sd[q_ + "attn.rel_pos_w"])
    rq = q.reshape(b * heads, hh, ww, d)
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, rw)
    attn = (attn.view(b * heads, hh, ww, hh, ww) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(b * heads, hh * ww, hh * ww)
    o = (attn.softmax(dim=-1) @ v).view(b, heads, hh, ww, d).permute(0, 2, 3, 1, 4).reshape(b, hh, ww, c)
    return F.linear(o, sd[q_ + "attn.proj.weight"], sd[q_ + "attn.proj.bias"])


def sam_block(sd, x, i, heads, window, lora=True, p=BB):
    """sam_vit.py:201-217 Block.forward with window_partition / unpartition :301-356 (zero padding AFTER norm1; padded
    tokens take part in the window's attention as keys)."""
    q_ = f"{p}blocks.{i}."
    b, hh, ww, c = x.shape
    h = F.layer_norm(x, (c,), sd[q_ + "norm1.weight"], sd[q_ + "norm1.bias"], 1e-6)
    if window > 0:
        ph, pw = (window - hh % window) % window, (window - ww % window) % window
        h = F.pad(h, (0, 0, 0, pw, 0, ph))
        hp, wp = hh + ph, ww + pw
        h = h.view(b, hp // window, window, wp // window, window, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, window, window, c)
    h = sam_attention(sd, q_, h, heads, lora)
    if window > 0:
        h = h.view(b, hp // window, wp // window, window, window, c).permute(0, 1, 3, 2, 4, 5).reshape(b, hp, wp, c)[:, :hh, :ww]
    x = x + h
    h = F.layer_norm(x, (c,), sd[q_ + "norm2.weight"], sd[q_ + "norm2.bias"], 1e-6)
    h = F.linear(F.gelu(F.linear(h, sd[q_ + "mlp.lin1.weight"], sd[q_ + "mlp.lin1.bias"])), sd[q_ + "mlp.lin2.weight"], sd[q_ + "mlp.lin2.bias"])
    return x + h


def sam_forward(sd, x, depth=32, heads=16, window=14, global_idx=(7, 15, 23, 31), out_indices=(7, 15, 23, 31), patch=16, lora=True, p=BB):
    """SAMViT.forward (sam_vit.py:127-148): NHWC tokens, no cls, fixed abs pos (input must be img_size), no neck."""
    t = F.conv2d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=patch).permute(0, 2, 3, 1)
    t = t + sd[p + "pos_embed"]
    outs = []
    for i in range(depth):
        t = sam_block(sd, t, i, heads, 0 if i in global_idx else window, lora, p)
        if i in out_indices:
            outs.append(t.permute(0, 3, 1, 2))
    return outs


# =============================================================================== heads
def batch_norm_train(x, w, b, running_mean, running_var, momentum=0.1, eps=1e-5, stats=None):
    """nn.SyncBatchNorm without a process group == BatchNorm2d (linear_head.py:44). Returns y and new running stats.
    `stats`=(mean, biased_var, count) overrides local batch stats (what a DP all-reduce of moments would give)."""
    in_dt = x.dtype
    x = x.float()   # fp16 inputs (oracle/amp_emul.py): batch-norm kernels accumulate and normalise in fp32 and round the result once
    if stats is None:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        cnt = x.numel() // x.shape[1]
    else:
        mean, var, cnt = stats
    y = (x - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + eps)
    y = (y * w[None, :, None, None] + b[None, :, None, None]).to(in_dt)
    new_rm = (1 - momentum) * running_mean + momentum * mean.detach()
    new_rv = (1 - momentum) * running_var + momentum * var.detach() * (cnt / max(cnt - 1, 1))
    return y, new_rm, new_rv


def linear_head_forward(sd, feats, training=False, drop2d=None, p="decode_head.", bn_out=None):
    """linear_head.py:50-70. feats: 4 x [B,C,h,w] -> logits [B,19,4h,4w].
    drop2d: Dropout2d multiplier [B,C/4,1,1] or None.  bn_out: dict that receives new running stats in training."""
    x = torch.cat(list(feats), dim=1)
    x = F.conv2d(x, sd[p + "fusion_conv.conv.weight"])  # ConvModule: bias off when a norm follows
    x = F.relu(F.group_norm(x, 32, sd[p + "fusion_conv.gn.weight"], sd[p + "fusion_conv.gn.bias"], 1e-5))
    x = F.conv_transpose2d(x, sd[p + "output_upscaling.0.weight"], sd[p + "output_upscaling.0.bias"], stride=2)
    if training:
        x, rm, rv = batch_norm_train(
            x, sd[p + "output_upscaling.1.weight"], sd[p + "output_upscaling.1.bias"],
            sd[p + "output_upscaling.1.running_mean"], sd[p + "output_upscaling.1.running_var"])
        if bn_out is not None:
            bn_out["running_mean"], bn_out["running_var"] = rm, rv
    else:
        x = F.batch_norm(x, sd[p + "output_upscaling.1.running_mean"], sd[p + "output_upscaling.1.running_var"],
                         sd[p + "output_upscaling.1.weight"], sd[p + "output_upscaling.1.bias"], False, 0.1, 1e-5)
    x = F.gelu(x)
    x = F.conv_transpose2d(x, sd[p + "output_upscaling.3.weight"], sd[p + "output_upscaling.3.bias"], stride=2)
    x = F.gelu(x)
    if drop2d is not None:
        x = x * drop2d
    return F.conv2d(x, sd[p + "conv_seg.weight"], sd[p + "conv_seg.bias"])


def _mha(sd, q_pre, x, ctx, heads=8):
    """Transformer.py:95-136 CrossAttention._forward (== the xformers branch :140-156)."""
    q = F.linear(x, sd[q_pre + "to_q.weight"])
    k = F.linear(ctx, sd[q_pre + "to_k.weight"])
    v = F.linear(ctx, sd[q_pre + "to_v.weight"])
    b, n, inner = q.shape
    d = inner // heads
    sp = lambda t: t.reshape(b, -1, heads, d).transpose(1, 2)
    o = attention(sp(q), sp(k), sp(v), d ** -0.5).transpose(1, 2).reshape(b, n, inner)
    return F.linear(o, sd[q_pre + "to_out.0.weight"], sd[q_pre + "to_out.0.bias"])


def mask_decoder_forward(sd, query, ctx, mask_keep=None, depth=3, heads=8, p="aux_decoder.transformer_decoder."):
    """Transformer.py:263-283 MaskTransformerDecoder.forward. query, ctx: [B,C,h,w].
    mask_keep: bool [B,1,h,w] (True keeps the feature, False -> mask_token) or None (mask disabled)."""
    b, c, h, w = ctx.shape
    if mask_keep is not None:
        query = torch.where(mask_keep, query, sd[p + "mask_token"].expand(b, -1, h, w))
    x = F.group_norm(query, 32, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)  # Normalize(): eps 1e-6 (:91-92)
    x = x.flatten(2).transpose(1, 2)
    cx = ctx.flatten(2).transpose(1, 2)
    for i in range(depth):
        q = f"{p}transformer_blocks.{i}."
        ln = lambda t, nm: F.layer_norm(t, (c,), sd[q + nm + ".weight"], sd[q + nm + ".bias"], 1e-5)
        x = _mha(sd, q + "attn1.", ln(x, "norm1"), ln(x, "norm1"), heads) + x  # self-attention (:174)
        x = _mha(sd, q + "attn2.", ln(x, "norm2"), cx, heads) + x  # cross-attention to the context (:175)
        hgl = F.linear(ln(x, "norm3"), sd[q + "ff.net.0.proj.weight"], sd[q + "ff.net.0.proj.bias"])
        a, gate = hgl.chunk(2, dim=-1)  # GEGLU (:52-59)
        x = F.linear(a * F.gelu(gate), sd[q + "ff.net.2.weight"], sd[q + "ff.net.2.bias"]) + x
    return x.transpose(1, 2).reshape(b, c, h, w)


def vfm_head_forward(sd, feats, ctx_logits, mask_keep=None, drop2d=None, p="aux_decoder.", depth=3):
    """VFMHead.py:61-89. feats 4x[B,C,h,w], ctx_logits [B,19,Hc,Wc] -> logits [B,19,h,w]."""
    h, w = feats[0].shape[2:]
    s = F.interpolate(ctx_logits, size=(h * 4, w * 4), mode="bilinear", align_corners=False)
    gelu = F.gelu
    f = F.conv2d(torch.cat(list(feats), dim=1), sd[p + "fuse_conv.0.weight"], sd[p + "fuse_conv.0.bias"])
    f = gelu(F.group_norm(f, 32, sd[p + "fuse_conv.1.weight"], sd[p + "fuse_conv.1.bias"], 1e-5))
    e = F.conv2d(s, sd[p + "seg_logits_embed.0.weight"], sd[p + "seg_logits_embed.0.bias"], stride=2)
    e = gelu(F.group_norm(e, 32, sd[p + "seg_logits_embed.1.weight"], sd[p + "seg_logits_embed.1.bias"], 1e-5))
    e = F.conv2d(e, sd[p + "seg_logits_embed.3.weight"], sd[p + "seg_logits_embed.3.bias"], stride=2)
    e = gelu(F.group_norm(e, 32, sd[p + "seg_logits_embed.4.weight"], sd[p + "seg_logits_embed.4.bias"], 1e-5))
    e = F.conv2d(e, sd[p + "seg_logits_embed.6.weight"], sd[p + "seg_logits_embed.6.bias"])
    e = F.group_norm(e, 32, sd[p + "seg_logits_embed.7.weight"], sd[p + "seg_logits_embed.7.bias"], 1e-5)
    # NB argument order (VFMHead.py:82 vs Transformer.py:270): query = fused image feats, context = logits embedding
    o = mask_decoder_forward(sd, f, e, mask_keep, depth=depth, p=p + "transformer_decoder.")
    if drop2d is not None:
        o = o * drop2d
    return F.conv2d(o, sd[p + "conv_seg.weight"], sd[p + "conv_seg.bias"])


def ce_loss_acc(seg_logits, label, ignore=IGNORE):
    """mmseg CrossEntropyLoss(avg_non_ignore=False) + accuracy (linear_head.py:93-108): the mean runs over ALL pixels."""
    loss = F.cross_entropy(seg_logits, label, reduction="none", ignore_index=ignore).mean()
    valid = label != ignore
    hit = (seg_logits.argmax(1) == label) & valid
    acc = hit.sum().float() * (100.0 / (valid.sum().float() + torch.finfo(torch.float32).eps))
    return loss, acc


def head_loss(logits_lowres, label):
    """`.loss()` of both heads: bilinear to the label size, CE, acc. label [B,1,H,W] int64."""
    up = F.interpolate(logits_lowres, size=label.shape[2:], mode="bilinear", align_corners=False)
    loss, acc = ce_loss_acc(up, label.squeeze(1))
    return loss, acc, up


# =============================================================================== segmentor: training
def get_crop_bbox(img_h, img_w, crop_size, divisible=1, rng=np.random):
    """Ms_VFM_encoder_decoder.py:34-46."""
    if img_h == crop_size[-2] and img_w == crop_size[-1]:
        return (0, img_h, 0, img_w)
    mh, mw = max(img_h - crop_size[-2], 0), max(img_w - crop_size[-1], 0)
    oh = rng.randint(0, (mh + 1) // divisible) * divisible
    ow = rng.randint(0, (mw + 1) // divisible) * divisible
    return oh, oh + crop_size[0], ow, ow + crop_size[1]


def forward_train(sd, img, label, hr_box, mask_keep, depth=24, heads=16, out_indices=(7, 11, 15, 23),
                  lora_drop_masks=(None, None), drop2d=(None, None), detail_loss=1.0, bn_out=None, dec_depth=3,
                  backbone="dinov2", backbone_kw=None):
    """MsVFMEncoderDecoder.forward_train (Ms_VFM_encoder_decoder.py:125-200) with the RNG consumers made explicit
    (SURVEY App. B): hr_box = (y1,y2,x1,x2), mask_keep = bool [B,1,32,32], optional dropout multipliers."""
    y1, y2, x1, x2 = hr_box
    lr_img = F.interpolate(img, scale_factor=0.5, mode="bilinear", align_corners=False)  # scales sorted [0.5, 1] (:87)
    hr_img = img[:, :, y1:y2, x1:x2]
    if backbone == "eva02":
        lr_feats = eva02_forward(sd, lr_img, depth, heads, out_indices)
        hr_feats = eva02_forward(sd, hr_img, depth, heads, out_indices)
    elif backbone == "clip":
        lr_feats = clip_forward(sd, lr_img, depth, heads, out_indices)
        hr_feats = clip_forward(sd, hr_img, depth, heads, out_indices)
    elif backbone == "sam":
        kw = dict(depth=depth, heads=heads, out_indices=out_indices, **(backbone_kw or {}))
        lr_feats = sam_forward(sd, lr_img, **kw)
        hr_feats = sam_forward(sd, hr_img, **kw)
    else:
        lr_feats = dinov2_forward(sd, lr_img, depth, heads, out_indices, drop_masks=lora_drop_masks[0])
        hr_feats = dinov2_forward(sd, hr_img, depth, heads, out_indices, drop_masks=lora_drop_masks[1])
    lr_gt = F.interpolate(label.float(), scale_factor=0.5, mode="nearest").long()  # get_lr_seg (:148-153)
    hr_gt = label[:, :, y1:y2, x1:x2]  # get_hr_seg (:155-158)
    lr_logits = linear_head_forward(sd, lr_feats, training=True, drop2d=drop2d[0], bn_out=bn_out)
    l_lr, a_lr, lr_up = head_loss(lr_logits, lr_gt)
    ctx = lr_up.detach()[:, :, y1 // 2:y2 // 2, x1 // 2:x2 // 2]  # get_seg_logits + resize_box(ratio=2) (:160-167)
    hr_logits = vfm_head_forward(sd, hr_feats, ctx, mask_keep, drop2d=drop2d[1], depth=dec_depth)
    l_hr, a_hr, _ = head_loss(hr_logits, hr_gt)
    return {
        "decode_lr.loss_ce": l_lr,
        "decode_lr.acc_seg": a_lr,
        "decode_hr.loss_ce": l_hr * detail_loss,
        "decode_hr.acc_seg": a_hr,
    }


def trainable_keys(sd):
    """LoRABackbone.train(): only names containing 'lora' train in the backbone (utils.py:9-23); heads train fully."""
    out = []
    for k, v in sd.items():
        if not torch.is_tensor(v) or not v.is_floating_point():
            continue
        if "running_" in k:
            continue
        if k.startswith("backbone."):
            if "lora_" in k:
                out.append(k)
        else:
            out.append(k)
    return out


def total_loss(losses):
    """mmengine BaseModel.parse_losses: sum of entries whose key contains 'loss'."""
    return sum(v for k, v in losses.items() if "loss" in k)


# =============================================================================== segmentor: inference
def grid_boxes(h_img, w_img, crop=(512, 512), stride=(320, 320)):
    """mmseg slide_inference window grid (also Ms_VFM_encoder_decoder.py:428-443)."""
    hc, wc = crop
    hs, ws = stride
    hg = max(h_img - hc + hs - 1, 0) // hs + 1
    wg = max(w_img - wc + ws - 1, 0) // ws + 1
    boxes = []
    for hi in range(hg):
        for wi in range(wg):
            y2 = min(hi * hs + hc, h_img)
            x2 = min(wi * ws + wc, w_img)
            boxes.append((max(y2 - hc, 0), y2, max(x2 - wc, 0), x2))
    return boxes


def whole_inference(sd, img, out_size, backbone="dinov2", **kw):
    """mmseg whole_inference -> encode_decode -> LinearHead.forward -> predict_by_feat (bilinear to img_shape)."""
    fwd = {"dinov2": dinov2_forward, "eva02": eva02_forward, "sam": sam_forward, "clip": clip_forward}[backbone]
    feats = fwd(sd, img, **kw)
    lg = linear_head_forward(sd, feats, training=False)
    return F.interpolate(lg, size=out_size, mode="bilinear", align_corners=False)


def slide_inference(sd, img, crop=(512, 512), stride=(320, 320), **kw):
    """mmseg EncoderDecoder.slide_inference with the LinearHead (cfg5-style 'slide' / 'hr_slide_inference')."""
    b, _, h, w = img.shape
    preds = img.new_zeros((b, 19, h, w))
    cnt = img.new_zeros((b, 1, h, w))
    for (y1, y2, x1, x2) in grid_boxes(h, w, crop, stride):
        lg = whole_inference(sd, img[:, :, y1:y2, x1:x2], (y2 - y1, x2 - x1), **kw)
        preds[:, :, y1:y2, x1:x2] += lg
        cnt[:, :, y1:y2, x1:x2] += 1
    return preds / cnt


def ms_inference(sd, img, thr=0.968, conf=0.8, crop=(512, 512), stride=(320, 320), trace=None, **kw):
    """Ms_VFM_encoder_decoder.py:400-466: coarse whole pass at a hard-coded (512,1024), then confidence-gated
    512^2 refinement with the VFMHead (mask disabled at test time, :422-423)."""
    b, _, h, w = img.shape
    small = F.interpolate(img, size=(512, 1024), mode="bilinear", align_corners=False)
    seg = whole_inference(sd, small, (h, w), **kw)  # predict_by_feat resizes straight to the image size
    preds = img.new_zeros((b, 19, h, w))
    cnt = img.new_zeros((b, 1, h, w))
    for (y1, y2, x1, x2) in grid_boxes(h, w, crop, stride):
        ctx = seg[:, :, y1:y2, x1:x2]
        frac = (ctx.softmax(dim=1).max(dim=1)[0] > thr).float().mean().item()
        if frac < conf:
            feats = dinov2_forward(sd, img[:, :, y1:y2, x1:x2], **kw)
            lg = vfm_head_forward(sd, feats, ctx, mask_keep=None)
            if trace is not None:
                trace.append((y1, y2, x1, x2))
        else:
            lg = ctx
        lg = F.interpolate(lg, size=(y2 - y1, x2 - x1), mode="bilinear", align_corners=False)
        preds[:, :, y1:y2, x1:x2] += lg
        cnt[:, :, y1:y2, x1:x2] += 1
    return preds / cnt


def lr_slide_inference(sd, img, crop=(512, 512), stride=(320, 320), **kw):
    """Ms_VFM_encoder_decoder.py:280-283."""
    lr = F.interpolate(img, scale_factor=0.5, mode="bilinear", align_corners=False)
    return F.interpolate(slide_inference(sd, lr, crop, stride, **kw), scale_factor=2, mode="bilinear", align_corners=False)


def msfull_slide_inference(sd, img, mask_keeps=None, crop=(512, 512), stride=(320, 320), **kw):
    """Ms_VFM_encoder_decoder.py:286-328: coarse sliding pass at (512, 1024), then every window through the VFMHead with the
    query mask ENABLED (enc_dec -> aux_decoder.forward; only ms_inference disables it).  mask_keeps: one bool [B,1,32,32] per
    window in grid order (the torch.rand consumer made explicit), or None for no masking."""
    b, _, h, w = img.shape
    small = F.interpolate(img, size=(512, 1024), mode="bilinear", align_corners=False)
    seg = F.interpolate(slide_inference(sd, small, crop, stride, **kw), size=(h, w), mode="bilinear", align_corners=False)
    preds = img.new_zeros((b, 19, h, w))
    cnt = img.new_zeros((b, 1, h, w))
    for j, (y1, y2, x1, x2) in enumerate(grid_boxes(h, w, crop, stride)):
        feats = dinov2_forward(sd, img[:, :, y1:y2, x1:x2], **kw)
        lg = vfm_head_forward(sd, feats, seg[:, :, y1:y2, x1:x2], None if mask_keeps is None else mask_keeps[j])
        preds[:, :, y1:y2, x1:x2] += F.interpolate(lg, size=(y2 - y1, x2 - x1), mode="bilinear", align_corners=False)
        cnt[:, :, y1:y2, x1:x2] += 1
    return preds / cnt


def postprocess_result(seg_logits, metas):
    """mmseg postprocess_result (1.2.2; tools/test.py:96-145 reaches it through predict): per image crop the padding
    (left, right, top, bottom), undo the flip, bilinear to ori_shape, argmax.  Returns [(logits [C,h,w], pred [1,h,w])]."""
    out = []
    B, C, H, W = seg_logits.shape
    for i in range(B):
        m = metas[i]
        pl, pr, pt, pb = m.get("img_padding_size", m.get("padding_size", [0] * 4))
        x = seg_logits[i:i + 1, :, pt:H - pb, pl:W - pr]
        if m.get("flip"):
            x = x.flip(dims=(3,)) if m.get("flip_direction") == "horizontal" else x.flip(dims=(2,))
        x = F.interpolate(x, size=tuple(m["ori_shape"]), mode="bilinear", align_corners=False).squeeze(0)
        out.append((x, x.argmax(dim=0, keepdim=True)))
    return out


# =============================================================================== optimiser
def param_group_options(name, module_is_norm, base_lr=1e-4, base_wd=0.05, custom_keys=None, norm_decay_mult=0.0):
    """peft_optimizer_constructor.py:25-147 for one parameter: custom key (longest first, substring of the full
    name) wins; otherwise norm-module params get wd * norm_decay_mult."""
    custom_keys = custom_keys or {}
    for key in sorted(sorted(custom_keys), key=len, reverse=True):
        if key in name:
            return base_lr * custom_keys[key].get("lr_mult", 1.0), base_wd * custom_keys[key].get("decay_mult", 1.0)
    if module_is_norm and norm_decay_mult is not None:
        return base_lr, base_wd * norm_decay_mult
    return base_lr, base_wd


def poly_lr(base_lr, t, end=40000, power=0.9, eta_min=0.0):
    """mmengine PolyLR closed form (dg_lora_dinov2_ms_masked.py:27-29)."""
    t = min(t, end)
    return (base_lr - eta_min) * (1 - t / end) ** power + eta_min


def adamw_step(p, g, m, v, step, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor update (decoupled decay), returns new (p, m, v)."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    return p - (lr / bc1) * m / denom, m, v


# parameters owned by normalisation MODULES of the reference's heads (isinstance check at peft_optimizer_constructor.py:93-96):
# ConvModule GN (linear_head.py:36-40), nn.SyncBatchNorm (:44), nn.GroupNorm (VFMHead.py:31, :39-49), and the decoder's
# LayerNorms (Transformer.py:167-169, :258) - the latter are hit by the custom key "norm" first anyway.
NORM_MODULE_PREFIXES = (
    "decode_head.fusion_conv.gn.", "decode_head.output_upscaling.1.", "aux_decoder.fuse_conv.1.",
    "aux_decoder.seg_logits_embed.1.", "aux_decoder.seg_logits_embed.4.", "aux_decoder.seg_logits_embed.7.",
)
DG_CUSTOM_KEYS = {"norm": dict(decay_mult=0.0), "query_embed": dict(lr_mult=1.0, decay_mult=0.0),
                  "level_embed": dict(lr_mult=1.0, decay_mult=0.0), "learnable_tokens": dict(lr_mult=1.0, decay_mult=0.0),
                  "reins.scale": dict(lr_mult=1.0, decay_mult=0.0)}   # dg_lora_dinov2_ms_masked.py:10-25


def train_step(sd, opt_state, img, label, hr_box, mask_keep, t, base_lr=1e-4, base_wd=0.05, end=40000, loss_scale=None, **kw):
    """One iteration of the reference's training loop on the oracle: forward_train -> parse_losses -> backward ->
    AdamW with the PEFTOptimWrapperConstructor groups at lr = PolyLR(t) -> SyncBN running stats
    (tools/train.py:64-121 via mmengine Runner; dg_lora_dinov2_ms_masked.py:10-29).  sd is updated IN PLACE (plain
    tensors); opt_state: {key: (m, v)} carried between calls; t = number of optimiser steps already taken.  Returns the
    loss dict of this iteration (floats).
    loss_scale (mmengine AmpOptimWrapper.update_params -> torch GradScaler.scale / unscale_ / step): backward runs on loss * scale,
    the gradients are divided by the scale, and a step whose gradients hold an inf / NaN is skipped (returned dict gets
    "skipped": 1.0; the caller owns the scale's schedule).  Run it inside oracle.amp_emul.cuda_autocast() for the `--amp` arithmetic."""
    tk = trainable_keys(sd)
    work = dict(sd)
    for k in tk:
        work[k] = sd[k].detach().clone().requires_grad_(True)
    bn = {}
    losses = forward_train(work, img, label, hr_box, mask_keep, bn_out=bn, **kw)
    total = total_loss(losses)
    grads = torch.autograd.grad(total if loss_scale is None else total * loss_scale, [work[k] for k in tk], allow_unused=True)
    out = {k: float(v.detach()) for k, v in losses.items()}
    if bn:   # the forward pass moved the running statistics whether or not the optimiser steps
        sd["decode_head.output_upscaling.1.running_mean"] = bn["running_mean"].detach()
        sd["decode_head.output_upscaling.1.running_var"] = bn["running_var"].detach()
    if loss_scale is not None:
        grads = [None if g is None else g.float() / loss_scale for g in grads]
        if not all(bool(torch.isfinite(g).all()) for g in grads if g is not None):
            out["skipped"] = 1.0
            return out
    lr_t = poly_lr(base_lr, t, end=end)
    with torch.no_grad():
        for k, g in zip(tk, grads):
            if g is None:   # never entered the graph (inert adapters): torch's AdamW skips grad-less parameters
                continue
            lr_k, wd_k = param_group_options(k, key_is_norm(k), base_lr=lr_t, base_wd=base_wd, custom_keys=DG_CUSTOM_KEYS)
            m, v = opt_state.get(k, (torch.zeros_like(g), torch.zeros_like(g)))
            sd[k], m, v = adamw_step(sd[k].detach(), g, m, v, t + 1, lr_k, wd_k)
            opt_state[k] = (m, v)
    return out


def key_is_norm(key):
    return key.startswith(NORM_MODULE_PREFIXES)


# =============================================================================== evaluation
def confusion_iou(pred, label, num_classes=19, ignore=IGNORE):
    """mmseg IoUMetric.intersect_and_union + per-class IoU / mIoU (rein/dg_metrics.py:74-102 groups these by dataset)."""
    valid = label != ignore
    p, l = pred[valid], label[valid]
    inter = torch.bincount(p[p == l], minlength=num_classes).double()
    ap = torch.bincount(p, minlength=num_classes).double()
    al = torch.bincount(l, minlength=num_classes).double()
    union = ap + al - inter
    iou = inter / union
    return iou, float(np.nanmean(iou.numpy()) * 100.0)


def intersect_and_union(pred_label, label, num_classes=19, ignore_index=IGNORE):
    """mmseg IoUMetric.intersect_and_union (called per sample at rein/dg_metrics.py:46-52): torch.histc over [0, nc-1] of
    pred[mask], label[mask] and pred[mask][pred == label]; out-of-range values fall outside the histogram."""
    mask = label != ignore_index
    pred_label, label = pred_label[mask].float(), label[mask].float()
    intersect = pred_label[pred_label == label]
    h = lambda t: torch.histc(t, bins=num_classes, min=0, max=num_classes - 1).double()   # noqa: E731
    ai, ap, al = h(intersect), h(pred_label), h(label)
    return ai, ap + al - ai, ap, al


def iou_metric_summary(results):
    """mmseg IoUMetric.compute_metrics for metrics=['mIoU']: totals over the samples -> aAcc, IoU, Acc per class ->
    np.round(nanmean * 100, 2) as aAcc / mIoU / mAcc."""
    ai, au, ap, al = (sum(r[i] for r in results) for i in range(4))
    ret = {"aAcc": (ai.sum() / al.sum()).numpy(), "IoU": (ai / au).numpy(), "Acc": (ai / al).numpy()}
    return {(k if k == "aAcc" else "m" + k): float(np.round(np.nanmean(v) * 100, 2)) for k, v in ret.items()}


def dg_iou_metrics(batches, dataset_keys, mean_used_keys=None, num_classes=19):
    """rein/dg_metrics.py:24-102.  batches: list of batches, each a list of (pred [H,W], label [H,W], seg_map_path); every
    sample of a batch is filed under the key found in the FIRST sample's path (:53-58)."""
    mean_used_keys = mean_used_keys or dataset_keys
    per = {}
    for batch in batches:
        key = "unknown"
        for k in dataset_keys:
            if k in batch[0][2]:
                key = k
                break
        for pred, label, _ in batch:
            per.setdefault(key, []).append(intersect_and_union(pred.long(), label.long(), num_classes))
    metrics, to_mean = {}, {}
    for key, res in per.items():
        for k, v in iou_metric_summary(res).items():
            metrics[f"{key}_{k}"] = v
            if key in mean_used_keys:
                to_mean.setdefault(k, []).append(v)
    for k, v in to_mean.items():
        metrics[f"mean_{k}"] = sum(v) / len(v)
    return metrics
