"""Checkpoint converters (reference: tools/convert_models/{convert_dinov2.py:34-83, convert_eva2_512x512.py:6-119,
convert_sam.py:21-91, convert_clip.py:21-79}): resize the 14x14 patch-embedding kernel to 16x16 and the positional grid to
the 512/16 = 32x32 training grid, strip the `image_encoder.` / `visual.` prefixes, drop EVA's rope buffers.  The resizes
are F.interpolate(mode='bicubic' | 'bilinear', align_corners=False) in the reference; here they run on the HIP resize
kernels (vfm_resize_bicubic / vfm_resize_bilinear), so a converted checkpoint is produced on the box that trains with it."""
import torch

from . import ops


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def resize_grid(grid_hwc, ho, wo, mode="bicubic"):
    """grid [Hi, Wi, C] (any device, any float dtype) -> fp32 [ho, wo, C] on the original device."""
    hi, wi, c = grid_hwc.shape
    src = grid_hwc.detach().to(_dev(), torch.float32).contiguous()
    out = torch.empty(ho, wo, c, dtype=torch.float32, device=src.device)
    if mode == "bicubic":
        ops.resize_bicubic(src, hi, wi, c, out, ho, wo, hi / ho, wi / wo)
    else:
        ops.resize_bilinear(src, False, 1, hi, wi, c, out, 0, (ho, wo))
    return out.to(grid_hwc.device)


def resize_conv_kernel(w, k):
    """[Co, Ci, kh, kw] -> [Co, Ci, k, k], bicubic over the two kernel axes (interpolate_patch_embed_)."""
    co, ci, kh, kw = w.shape
    g = resize_grid(w.float().permute(2, 3, 0, 1).reshape(kh, kw, co * ci), k, k)
    return g.reshape(k, k, co, ci).permute(2, 3, 0, 1).contiguous()


def _resize_tokens(tok, new_hw, mode="bicubic"):
    """tok [n*n, C] -> [h*w, C]"""
    n = int(round(tok.shape[0] ** 0.5))
    assert n * n == tok.shape[0]
    return resize_grid(tok.float().reshape(n, n, -1), new_hw[0], new_hw[1], mode).reshape(new_hw[0] * new_hw[1], -1)


def convert_dinov2(sd, kernel=16, crop=(512, 512)):
    """convert_dinov2.py:34-83"""
    sd = dict(sd)
    sd["patch_embed.proj.weight"] = resize_conv_kernel(sd["patch_embed.proj.weight"], kernel)
    pe = sd["pos_embed"]
    hw = (crop[0] // kernel, crop[1] // kernel)
    sd["pos_embed"] = torch.cat((pe[:, :1].float(), _resize_tokens(pe[0, 1:], hw)[None]), dim=1)
    return sd


def convert_eva02(sd, kernel=16, grid=32):
    """convert_eva2_512x512.py:6-119: drop rope buffers, 16x16 patch kernel, 32x32 positional grid."""
    if "model" in sd:
        sd = sd["model"]
    sd = {k: v for k, v in sd.items() if "rope" not in k}
    sd["patch_embed.proj.weight"] = resize_conv_kernel(sd["patch_embed.proj.weight"], kernel)
    if "pos_embed" in sd:
        pe = sd["pos_embed"]
        sd["pos_embed"] = torch.cat((pe[:, :1].float(), _resize_tokens(pe[0, 1:], (grid, grid))[None]), dim=1)
    if "positional_embedding" in sd:
        pe = sd["positional_embedding"]
        sd["positional_embedding"] = torch.cat((pe[:1].float(), _resize_tokens(pe[1:], (grid, grid))), dim=0)
    return sd


def convert_sam(sd, kernel=16, crop=(512, 512)):
    """convert_sam.py:21-91: keep image_encoder.*, 16x16 patch kernel, [1, H, W, C] positional grid resized."""
    sd = {k.replace("image_encoder.", ""): v for k, v in sd.items() if "image_encoder." in k}
    sd["patch_embed.proj.weight"] = resize_conv_kernel(sd["patch_embed.proj.weight"], kernel)
    pe = sd["pos_embed"]
    hw = (crop[0] // kernel, crop[1] // kernel)
    sd["pos_embed"] = resize_grid(pe[0].float(), hw[0], hw[1])[None]
    return sd


def convert_clip(sd, resolution=512, patch=16, dim=1024):
    """convert_clip.py:21-79: keep visual.*, BILINEAR positional grid, bicubic patch kernel."""
    sd = {k.replace("visual.", ""): v.float() for k, v in sd.items() if k.startswith("visual.")}
    ss = resolution // patch
    if "positional_embedding" in sd:
        pe = sd["positional_embedding"]
        sd["positional_embedding"] = torch.cat((pe[:1], _resize_tokens(pe[1:], (ss, ss), mode="bilinear")), dim=0)
    sd["conv1.weight"] = resize_conv_kernel(sd["conv1.weight"], patch)
    return sd
