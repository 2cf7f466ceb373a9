"""Deterministic synthetic weights and inputs (no checkpoints or datasets exist offline).

Construction-order independent: every tensor is drawn from its own generator seeded
with crc32(key), so the reference modules (oracle/gen_golden.py), the CPU oracle and
the HIP product path all see identical parameters regardless of how they were built.
SURVEY.md §7 step 0 / §8(d) "Synthetic inputs".
"""
import zlib

import torch


def _std_for(key: str, shape) -> tuple:
    """(mean, std) of the synthetic distribution for a state_dict key."""
    leaf = key.split(".")[-1]
    is_norm = any(t in key for t in (".norm", "norm1", "norm2", "norm3", ".gn.", "ln_")) or (
        len(shape) == 1 and leaf == "weight"
    )
    if "running_mean" in key:
        return 0.0, 0.1
    if "running_var" in key:
        return 1.0, 0.0
    if "num_batches_tracked" in key:
        return 0.0, 0.0
    if leaf == "gamma":  # LayerScale: default 1e-5 would hide the residual branches
        return 1.0, 0.1
    if is_norm and leaf == "weight":
        return 1.0, 0.1
    if leaf == "bias":
        return 0.0, 0.02
    if "mask_token" in key:
        return 0.0, 1.0 if len(shape) == 4 else 0.02
    if "conv_seg.weight" in key:  # decisive logits: argmax / confidence-gate margins >> rounding noise
        return 0.0, (0.15 if "aux_decoder" in key else 2.0)
    if any(t in key for t in ("qkv", "to_q", "to_k", "q_proj", "k_proj")):  # peaky attention
        return 0.0, 0.04
    return 0.0, 0.02


def synth_tensor(key: str, shape, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    mean, std = _std_for(key, tuple(shape))
    if dtype in (torch.int64, torch.int32):
        return torch.zeros(shape, dtype=dtype)
    t = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if "running_var" in key:
        return torch.ones(tuple(shape), dtype=dtype) + 0.1 * t.abs().to(dtype)
    return (t * std + mean).to(dtype)


def synth_state_dict(shapes: dict) -> dict:
    """shapes: {key: (shape, dtype)} or {key: shape} -> {key: tensor}, sorted by key."""
    out = {}
    for k in sorted(shapes):
        v = shapes[k]
        if isinstance(v, tuple) and len(v) == 2 and isinstance(v[1], torch.dtype):
            shape, dt = v
        else:
            shape, dt = v, torch.float32
        out[k] = synth_tensor(k, shape, dt)
    return out


def synth_like(state_dict: dict) -> dict:
    return synth_state_dict({k: (tuple(v.shape), v.dtype) for k, v in state_dict.items()})


def synth_image(batch: int, size, seed: int = 0, cell: int = 64) -> torch.Tensor:
    """Already-normalised image batch: unit noise plus a blocky low-frequency pattern
    (so that different crops of one image have different statistics)."""
    if isinstance(size, int):
        size = (size, size)
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + seed)
    h, w = size
    noise = torch.randn(batch, 3, h, w, generator=g)
    coarse = torch.randn(batch, 3, (h + cell - 1) // cell, (w + cell - 1) // cell, generator=g)
    amp = torch.rand(batch, 1, (h + cell - 1) // cell, (w + cell - 1) // cell, generator=g)
    up = lambda t: t.repeat_interleave(cell, 2).repeat_interleave(cell, 3)[:, :, :h, :w]
    return (noise * (0.25 + 1.5 * up(amp)) + 1.5 * up(coarse)).contiguous()


def synth_label(batch: int, size, num_classes: int = 19, seed: int = 0, cell: int = 32) -> torch.Tensor:
    """Blocky label map [B,1,H,W] int64 with a ~5% band of ignore(255) pixels."""
    if isinstance(size, int):
        size = (size, size)
    g = torch.Generator(device="cpu")
    g.manual_seed(2000 + seed)
    h, w = size
    coarse = torch.randint(0, num_classes, (batch, 1, (h + cell - 1) // cell, (w + cell - 1) // cell), generator=g)
    lab = coarse.repeat_interleave(cell, 2).repeat_interleave(cell, 3)[:, :, :h, :w].contiguous()
    band0 = int(torch.randint(0, max(h - h // 20, 1), (1,), generator=g))
    lab[:, :, band0 : band0 + max(h // 20, 1), :] = 255
    return lab.long()
