"""Shared test helpers: synthetic state dict for the full DINOv2-L ms_masked model (no reference needed)."""
import functools

import numpy as np
import torch

from vfmseg_amd.synth import synth_state_dict


def model_shapes(depth=24, dim=1024, r=32, num_classes=19, ch=256, dec_depth=3, n_pos=1025):
    """Key -> shape of MsVFMEncoderDecoder(LoRABackbone(DinoVisionTransformer), LinearHead, VFMHead) state_dict
    (SURVEY.md §8b 'state_dict keys')."""
    s = {}
    bb = "backbone.model.base_model.model."
    s[bb + "cls_token"] = (1, 1, dim)
    s[bb + "pos_embed"] = (1, n_pos, dim)
    s[bb + "mask_token"] = (1, dim)
    s[bb + "patch_embed.proj.weight"] = (dim, 3, 16, 16)
    s[bb + "patch_embed.proj.bias"] = (dim,)
    s[bb + "norm.weight"] = (dim,)
    s[bb + "norm.bias"] = (dim,)
    for i in range(depth):
        q = f"{bb}blocks.{i}."
        for n in ("norm1", "norm2"):
            s[q + n + ".weight"] = (dim,)
            s[q + n + ".bias"] = (dim,)
        s[q + "attn.qkv.base_layer.weight"] = (3 * dim, dim)
        s[q + "attn.qkv.base_layer.bias"] = (3 * dim,)
        s[q + "attn.qkv.lora_A.default.weight"] = (r, dim)
        s[q + "attn.qkv.lora_B.default.weight"] = (3 * dim, r)
        s[q + "attn.proj.weight"] = (dim, dim)
        s[q + "attn.proj.bias"] = (dim,)
        s[q + "ls1.gamma"] = (dim,)
        s[q + "ls2.gamma"] = (dim,)
        s[q + "mlp.fc1.weight"] = (4 * dim, dim)
        s[q + "mlp.fc1.bias"] = (4 * dim,)
        s[q + "mlp.fc2.weight"] = (dim, 4 * dim)
        s[q + "mlp.fc2.bias"] = (dim,)
    d = "decode_head."
    s[d + "conv_seg.weight"] = (num_classes, dim // 4, 1, 1)
    s[d + "conv_seg.bias"] = (num_classes,)
    s[d + "fusion_conv.conv.weight"] = (dim, 4 * dim, 1, 1)
    s[d + "fusion_conv.gn.weight"] = (dim,)
    s[d + "fusion_conv.gn.bias"] = (dim,)
    s[d + "output_upscaling.0.weight"] = (dim, dim // 2, 2, 2)
    s[d + "output_upscaling.0.bias"] = (dim // 2,)
    s[d + "output_upscaling.1.weight"] = (dim // 2,)
    s[d + "output_upscaling.1.bias"] = (dim // 2,)
    s[d + "output_upscaling.1.running_mean"] = (dim // 2,)
    s[d + "output_upscaling.1.running_var"] = (dim // 2,)
    s[d + "output_upscaling.1.num_batches_tracked"] = ((), torch.int64)
    s[d + "output_upscaling.3.weight"] = (dim // 2, dim // 4, 2, 2)
    s[d + "output_upscaling.3.bias"] = (dim // 4,)
    a = "aux_decoder."
    s[a + "conv_seg.weight"] = (num_classes, ch, 1, 1)
    s[a + "conv_seg.bias"] = (num_classes,)
    s[a + "fuse_conv.0.weight"] = (ch, 4 * dim, 1, 1)
    s[a + "fuse_conv.0.bias"] = (ch,)
    s[a + "fuse_conv.1.weight"] = (ch,)
    s[a + "fuse_conv.1.bias"] = (ch,)
    s[a + "seg_logits_embed.0.weight"] = (ch // 4, 19, 2, 2)
    s[a + "seg_logits_embed.0.bias"] = (ch // 4,)
    s[a + "seg_logits_embed.1.weight"] = (ch // 4,)
    s[a + "seg_logits_embed.1.bias"] = (ch // 4,)
    s[a + "seg_logits_embed.3.weight"] = (ch // 2, ch // 4, 2, 2)
    s[a + "seg_logits_embed.3.bias"] = (ch // 2,)
    s[a + "seg_logits_embed.4.weight"] = (ch // 2,)
    s[a + "seg_logits_embed.4.bias"] = (ch // 2,)
    s[a + "seg_logits_embed.6.weight"] = (ch, ch // 2, 1, 1)
    s[a + "seg_logits_embed.6.bias"] = (ch,)
    s[a + "seg_logits_embed.7.weight"] = (ch,)
    s[a + "seg_logits_embed.7.bias"] = (ch,)
    t = a + "transformer_decoder."
    s[t + "mask_token"] = (1, ch, 1, 1)
    s[t + "norm.weight"] = (ch,)
    s[t + "norm.bias"] = (ch,)
    inner = 512
    for i in range(dec_depth):
        q = f"{t}transformer_blocks.{i}."
        for at in ("attn1", "attn2"):
            for w in ("to_q", "to_k", "to_v"):
                s[q + f"{at}.{w}.weight"] = (inner, ch)
            s[q + f"{at}.to_out.0.weight"] = (ch, inner)
            s[q + f"{at}.to_out.0.bias"] = (ch,)
        s[q + "ff.net.0.proj.weight"] = (ch * 8, ch)
        s[q + "ff.net.0.proj.bias"] = (ch * 8,)
        s[q + "ff.net.2.weight"] = (ch, ch * 4)
        s[q + "ff.net.2.bias"] = (ch,)
        for n in ("norm1", "norm2", "norm3"):
            s[q + n + ".weight"] = (ch,)
            s[q + n + ".bias"] = (ch,)
    return s


# The reference's loader fills base weights from a checkpoint in the un-renamed key scheme
# (lora_backbone.py:27-35): the value of `...qkv.base_layer.weight` is synth('...qkv.weight') of the bare backbone.
def full_state_dict(depth=24, dim=1024, **kw):
    shapes = model_shapes(depth, dim, **kw)
    sd = synth_state_dict(shapes)
    bb = "backbone.model.base_model.model."
    bare = {}
    for k, v in shapes.items():
        if k.startswith(bb) and "lora_" not in k:
            bare[k[len(bb):].replace(".base_layer", "")] = v
    bare_sd = synth_state_dict(bare)
    for k in list(sd):
        if k.startswith(bb) and "lora_" not in k:
            sd[k] = bare_sd[k[len(bb):].replace(".base_layer", "")]
    return sd


@functools.lru_cache(maxsize=2)
def cached_full_state_dict(depth=24, dim=1024):
    return full_state_dict(depth, dim)


def stats(t):
    t = t.detach().double().cpu()
    return np.array([t.mean().item(), t.abs().mean().item(), t.std().item(), t.abs().max().item()])


def sl(t, n=8):
    idx = tuple(slice(0, min(n, s)) for s in t.shape)
    return t.detach().cpu()[idx].contiguous().numpy()


def rel_err(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def eva02_shapes(depth=24, dim=1024, r=32, hidden=2730, n_pos=1025):
    """state_dict key -> shape of LoRABackbone(EVA2) under the 'backbone.' prefix (rope cos/sin buffers excluded:
    they are deterministic tables, rebuilt by both sides)."""
    s = {}
    bb = "backbone.model.base_model.model."
    s[bb + "cls_token"] = (1, 1, dim)
    s[bb + "pos_embed"] = (1, n_pos, dim)
    s[bb + "patch_embed.proj.weight"] = (dim, 3, 16, 16)
    s[bb + "patch_embed.proj.bias"] = (dim,)
    for i in range(depth):
        q = f"{bb}blocks.{i}."
        for n in ("norm1", "norm2"):
            s[q + n + ".weight"] = (dim,)
            s[q + n + ".bias"] = (dim,)
        for nm in ("q_proj", "k_proj", "v_proj"):
            s[q + f"attn.{nm}.base_layer.weight"] = (dim, dim)
            s[q + f"attn.{nm}.lora_A.default.weight"] = (r, dim)
            s[q + f"attn.{nm}.lora_B.default.weight"] = (dim, r)
        s[q + "attn.q_bias"] = (dim,)
        s[q + "attn.v_bias"] = (dim,)
        s[q + "attn.proj.base_layer.weight"] = (dim, dim)
        s[q + "attn.proj.base_layer.bias"] = (dim,)
        s[q + "attn.proj.lora_A.default.weight"] = (r, dim)
        s[q + "attn.proj.lora_B.default.weight"] = (dim, r)
        s[q + "mlp.w1.weight"] = (hidden, dim)
        s[q + "mlp.w1.bias"] = (hidden,)
        s[q + "mlp.w2.weight"] = (hidden, dim)
        s[q + "mlp.w2.bias"] = (hidden,)
        s[q + "mlp.ffn_ln.weight"] = (hidden,)
        s[q + "mlp.ffn_ln.bias"] = (hidden,)
        s[q + "mlp.w3.weight"] = (dim, hidden)
        s[q + "mlp.w3.bias"] = (dim,)
    return s


def eva02_state_dict(depth=24, dim=1024):
    """Same construction as the reference loader path: base weights = synth of the un-renamed bare-backbone key,
    LoRA factors = synth of the wrapped key (without the 'backbone.' prefix, as gen_golden builds a bare LoRABackbone)."""
    shapes = eva02_shapes(depth, dim)
    bb = "backbone.model.base_model.model."
    wrapped = {k[len("backbone."):]: v for k, v in shapes.items()}
    sd = {"backbone." + k: v for k, v in synth_state_dict(wrapped).items()}
    bare = {k[len(bb):].replace(".base_layer", ""): v for k, v in shapes.items() if "lora_" not in k}
    bare_sd = synth_state_dict(bare)
    for k in list(sd):
        if "lora_" not in k:
            sd[k] = bare_sd[k[len(bb):].replace(".base_layer", "")]
    return sd


def clip_shapes(depth=24, dim=1024, r=32, n_pos=1025):
    """state_dict key -> shape of LoRABackbone(CLIPVisionTransformer) under 'backbone.' (the fpn layers the reference
    constructs but never calls are left out)."""
    s = {}
    bb = "backbone.model.base_model.model."
    s[bb + "conv1.weight"] = (dim, 3, 16, 16)
    s[bb + "class_embedding"] = (dim,)
    s[bb + "positional_embedding"] = (n_pos, dim)
    s[bb + "ln_pre.weight"] = (dim,)
    s[bb + "ln_pre.bias"] = (dim,)
    for i in range(depth):
        q = f"{bb}transformer.resblocks.{i}."
        for n in ("ln_1", "ln_2"):
            s[q + n + ".weight"] = (dim,)
            s[q + n + ".bias"] = (dim,)
        s[q + "attn.in_proj_weight"] = (3 * dim, dim)
        s[q + "attn.in_proj_bias"] = (3 * dim,)
        for nm, (o, i_) in (("attn.out_proj", (dim, dim)), ("mlp.c_fc", (4 * dim, dim)), ("mlp.c_proj", (dim, 4 * dim))):
            s[q + nm + ".base_layer.weight"] = (o, i_)
            s[q + nm + ".base_layer.bias"] = (o,)
            s[q + nm + ".lora_A.default.weight"] = (r, i_)
            s[q + nm + ".lora_B.default.weight"] = (o, r)
    return s


def clip_state_dict(depth=24, dim=1024):
    """Base weights = synth of the bare-backbone key, LoRA factors = synth of the wrapped key (see eva02_state_dict)."""
    shapes = clip_shapes(depth, dim)
    bb = "backbone.model.base_model.model."
    wrapped = {k[len("backbone."):]: v for k, v in shapes.items()}
    sd = {"backbone." + k: v for k, v in synth_state_dict(wrapped).items()}
    bare = {k[len(bb):].replace(".base_layer", ""): v for k, v in shapes.items() if "lora_" not in k}
    bare_sd = synth_state_dict(bare)
    for k in list(sd):
        if "lora_" not in k:
            sd[k] = bare_sd[k[len(bb):].replace(".base_layer", "")]
    return sd


def sam_shapes(depth=32, dim=1280, heads=16, r=32, grid=32, window=14, global_idx=(7, 15, 23, 31)):
    s = {}
    bb = "backbone.model.base_model.model."
    d = dim // heads
    s[bb + "pos_embed"] = (1, grid, grid, dim)
    s[bb + "patch_embed.proj.weight"] = (dim, 3, 16, 16)
    s[bb + "patch_embed.proj.bias"] = (dim,)
    for i in range(depth):
        q = f"{bb}blocks.{i}."
        for n in ("norm1", "norm2"):
            s[q + n + ".weight"] = (dim,)
            s[q + n + ".bias"] = (dim,)
        s[q + "attn.qkv.base_layer.weight"] = (3 * dim, dim)
        s[q + "attn.qkv.base_layer.bias"] = (3 * dim,)
        s[q + "attn.qkv.lora_A.default.weight"] = (r, dim)
        s[q + "attn.qkv.lora_B.default.weight"] = (3 * dim, r)
        s[q + "attn.proj.weight"] = (dim, dim)
        s[q + "attn.proj.bias"] = (dim,)
        L = (4 * grid - 1) if i in global_idx else (2 * window - 1)
        s[q + "attn.rel_pos_h"] = (L, d)
        s[q + "attn.rel_pos_w"] = (L, d)
        s[q + "mlp.lin1.weight"] = (4 * dim, dim)
        s[q + "mlp.lin1.bias"] = (4 * dim,)
        s[q + "mlp.lin2.weight"] = (dim, 4 * dim)
        s[q + "mlp.lin2.bias"] = (dim,)
    return s


def sam_state_dict(depth=32, global_idx=(7, 15, 23, 31), **kw):
    shapes = sam_shapes(depth, global_idx=global_idx, **kw)
    bb = "backbone.model.base_model.model."
    wrapped = {k[len("backbone."):]: v for k, v in shapes.items()}
    sd = {"backbone." + k: v for k, v in synth_state_dict(wrapped).items()}
    bare = {k[len(bb):].replace(".base_layer", ""): v for k, v in shapes.items() if "lora_" not in k}
    bare_sd = synth_state_dict(bare)
    for k in list(sd):
        if "lora_" not in k:
            sd[k] = bare_sd[k[len(bb):].replace(".base_layer", "")]
    return sd
