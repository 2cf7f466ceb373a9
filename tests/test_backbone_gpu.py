"""DINOv2 + LoRA backbone on the HIP path vs the CPU oracle: taps and LoRA gradients (f32 parity mode and bf16)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402
from tests.helpers import full_state_dict, rel_err  # noqa: E402
from vfmseg_amd import presets  # noqa: E402
from vfmseg_amd.precision import set_compute_dtype  # noqa: E402
from vfmseg_amd.registry import MODELS  # noqa: E402
from vfmseg_amd.synth import synth_image  # noqa: E402
import vfmseg_amd.backbones  # noqa: E402,F401

DEPTH, OUT = 4, (0, 1, 2, 3)


def _build(sd):
    cfg = dict(type="LoRABackbone", backbone=dict(presets.dinov2_backbone(depth=DEPTH), out_indices=list(OUT)),
               Lora_config=presets.lora_cfg(dropout=0.0))
    m = MODELS.build(cfg)
    bsd = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    missing, unexpected = m.load_state_dict(bsd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return m.cuda()


@pytest.mark.parametrize("mode,tol,gtol", [("f32", 2e-4, 2e-3), ("bf16", 3e-2, 8e-2)])
def test_backbone_taps_and_lora_grads(mode, tol, gtol):
    set_compute_dtype(mode)
    try:
        sd = full_state_dict(depth=DEPTH)
        m = _build(sd)
        img = synth_image(2, 512, seed=21)
        # oracle (CPU fp32) with autograd on the LoRA factors
        sdo = dict(sd)
        tk = [k for k in sd if "lora_" in k]
        for k in tk:
            sdo[k] = sd[k].clone().requires_grad_(True)
        taps = R.dinov2_forward(sdo, img, depth=DEPTH, out_indices=OUT)
        gen = torch.Generator().manual_seed(5)
        dts = [torch.randn(t.shape, generator=gen) for t in taps]
        loss = sum((t * d).sum() for t, d in zip(taps, dts))
        og = dict(zip(tk, torch.autograd.grad(loss, [sdo[k] for k in tk])))
        # HIP
        xcat, (hp, wp) = m.forward_tokens([(img.cuda(), None)])
        assert (hp, wp) == (32, 32) and xcat.shape == (2 * 1024, 4 * 1024)
        v = xcat.float().view(2, 32, 32, 4, 1024)
        for i, t in enumerate(taps):
            got = v[:, :, :, i].permute(0, 3, 1, 2).cpu()
            assert rel_err(got, t.detach()) < tol, (i, rel_err(got, t.detach()))
        dx = torch.stack([d.permute(0, 2, 3, 1) for d in dts], dim=3).reshape(2 * 1024, 4 * 1024).to(xcat.dtype).cuda()
        xcat.backward(dx)
        for n, p in m.named_parameters():
            if "lora_" in n:
                e = rel_err(p.grad.cpu(), og["backbone." + n])
                assert e < gtol, (n, e)
    finally:
        set_compute_dtype("bf16")


def test_api_views_match_reference_layout():
    set_compute_dtype("f32")
    try:
        sd = full_state_dict(depth=DEPTH)
        m = _build(sd).eval()
        img = synth_image(1, 512, seed=22)
        with torch.no_grad():
            feats = m(img.cuda())
            ref = R.dinov2_forward(sd, img, depth=DEPTH, out_indices=OUT)
        assert len(feats) == 4 and tuple(feats[0].shape) == (1, 1024, 32, 32)
        assert rel_err(feats[3].cpu(), ref[3]) < 2e-4
    finally:
        set_compute_dtype("bf16")


@pytest.mark.parametrize("mode,tol,gtol", [("f32", 3e-4, 3e-3), ("bf16", 3e-2, 1e-1)])
def test_eva02_taps_and_lora_grads(mode, tol, gtol):
    """EVA02 (rope, SwiGLU with inner LN, LoRA on attn.proj; q/k/v adapters inert) vs the CPU oracle, depth 4."""
    from tests.helpers import eva02_state_dict
    import vfmseg_amd.eva  # noqa: F401
    set_compute_dtype(mode)
    try:
        sd = eva02_state_dict(depth=DEPTH)
        cfg = dict(type="LoRABackbone", backbone=dict(presets.eva02_backbone(depth=DEPTH), out_indices=list(OUT)),
                   Lora_config=presets.eva02_lora_cfg(dropout=0.0))
        m = MODELS.build(cfg)
        bsd = {k[len("backbone."):]: v for k, v in sd.items()}
        missing, unexpected = m.load_state_dict(bsd, strict=False)
        assert not unexpected and all("rope" in k for k in missing), (missing, unexpected)
        m = m.cuda()
        img = synth_image(1, 512, seed=23)
        sdo = dict(sd)
        tk = [k for k in sd if "attn.proj.lora_" in k]
        for k in tk:
            sdo[k] = sd[k].clone().requires_grad_(True)
        taps = R.eva02_forward(sdo, img, depth=DEPTH, out_indices=OUT)
        gen = torch.Generator().manual_seed(5)
        dts = [torch.randn(t.shape, generator=gen) for t in taps]
        loss = sum((t * d).sum() for t, d in zip(taps, dts))
        og = dict(zip(tk, torch.autograd.grad(loss, [sdo[k] for k in tk])))
        xcat, (hp, wp) = m.forward_tokens([(img.cuda(), None)])
        v = xcat.float().view(1, 32, 32, 4, 1024)
        for i, t in enumerate(taps):
            e = rel_err(v[:, :, :, i].permute(0, 3, 1, 2).cpu(), t.detach())
            assert e < tol, (i, e)
        dx = torch.stack([d.permute(0, 2, 3, 1) for d in dts], dim=3).reshape(1024, 4 * 1024).to(xcat.dtype).cuda()
        xcat.backward(dx)
        n_live = 0
        for n, p in m.named_parameters():
            if "attn.proj.lora_" in n:
                e = rel_err(p.grad.cpu(), og["backbone." + n])
                assert e < gtol, (n, e)
                n_live += 1
            elif "lora_" in n:
                assert not p.requires_grad and p.grad is None   # inert adapters
        assert n_live == 2 * DEPTH
    finally:
        set_compute_dtype("bf16")


@pytest.mark.parametrize("mode,tol,gtol", [("f32", 3e-4, 3e-3), ("bf16", 3e-2, 1e-1)])
def test_clip_taps_and_lora_grads(mode, tol, gtol):
    """CLIP ViT-L/16 (double class embedding, ln_pre, packed in_proj, QuickGELU, LoRA on mlp.c_fc / mlp.c_proj; the
    out_proj adapter is inert) vs the CPU oracle, depth 4."""
    from tests.helpers import clip_state_dict
    set_compute_dtype(mode)
    try:
        sd = clip_state_dict(depth=DEPTH)
        cfg = dict(type="LoRABackbone", backbone=presets.clip_backbone(layers=DEPTH, out_indices=OUT),
                   Lora_config=presets.clip_lora_cfg(dropout=0.0))
        m = MODELS.build(cfg)
        bsd = {k[len("backbone."):]: v for k, v in sd.items()}
        missing, unexpected = m.load_state_dict(bsd, strict=False)
        assert not unexpected and all(".fpn" in k for k in missing), (missing, unexpected)
        m = m.cuda()
        img = synth_image(1, 512, seed=53)
        sdo = dict(sd)
        tk = [k for k in sd if "mlp.c_" in k and "lora_" in k]
        for k in tk:
            sdo[k] = sd[k].clone().requires_grad_(True)
        taps = R.clip_forward(sdo, img, depth=DEPTH, out_indices=OUT)
        gen = torch.Generator().manual_seed(8)
        dts = [torch.randn(t.shape, generator=gen) for t in taps]
        loss = sum((t * d).sum() for t, d in zip(taps, dts))
        og = dict(zip(tk, torch.autograd.grad(loss, [sdo[k] for k in tk])))
        xcat, (hp, wp) = m.forward_tokens([(img.cuda(), None)])
        v = xcat.float().view(1, 32, 32, 4, 1024)
        for i, t in enumerate(taps):
            e = rel_err(v[:, :, :, i].permute(0, 3, 1, 2).cpu(), t.detach())
            assert e < tol, (i, e)
        dx = torch.stack([d.permute(0, 2, 3, 1) for d in dts], dim=3).reshape(1024, 4 * 1024).to(xcat.dtype).cuda()
        xcat.backward(dx)
        n_live = 0
        for n, p in m.named_parameters():
            if "mlp.c_" in n and "lora_" in n:
                e = rel_err(p.grad.cpu(), og["backbone." + n])
                assert e < gtol, (n, e)
                n_live += 1
            elif "lora_" in n:
                assert not p.requires_grad and p.grad is None   # the out_proj adapter
        assert n_live == 4 * DEPTH
    finally:
        set_compute_dtype("bf16")


def test_clip_lora_dropout_training_smoke():
    """bf16 training mode with lora_dropout 0.1: both adapter sites draw masks (fused LN+dropout for c_fc, mask kernel for
    c_proj); gradients are finite, non-zero, and reproducible for a fixed seed."""
    from tests.helpers import clip_state_dict
    set_compute_dtype("bf16")
    sd = clip_state_dict(depth=2)
    cfg = dict(type="LoRABackbone", backbone=presets.clip_backbone(layers=2, out_indices=(0, 1)), Lora_config=presets.clip_lora_cfg(0.1))
    m = MODELS.build(cfg)
    m.load_state_dict({k[len("backbone."):]: v for k, v in sd.items()}, strict=False)
    m = m.cuda().train()
    img = synth_image(2, 512, seed=54).cuda()
    outs = []
    for rep in range(2):
        for p in m.parameters():
            p.grad = None
        xcat, _ = m.forward_tokens([(img, None)], seed=77)
        xcat.backward(torch.ones_like(xcat))
        g = torch.cat([p.grad.flatten() for n, p in m.named_parameters() if p.requires_grad])
        assert torch.isfinite(g).all() and g.abs().sum() > 0
        outs.append((xcat.detach().clone(), g.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("mode,tol", [("f32", 3e-4), ("bf16", 4e-2)])
def test_sam_taps(mode, tol):
    """SAM-ViT-H widths, depth 8 (windowed blocks + 2 global blocks, non-zero rel-pos tables, LoRA on qkv) vs the oracle."""
    from tests.helpers import sam_state_dict
    import vfmseg_amd.sam  # noqa: F401
    set_compute_dtype(mode)
    try:
        depth, gidx, oidx = 8, (3, 7), (1, 3, 5, 7)
        sd = sam_state_dict(depth=depth, global_idx=gidx)
        cfg = dict(type="LoRABackbone", backbone=presets.sam_backbone(depth=depth, global_idx=gidx, out_indices=oidx),
                   Lora_config=presets.lora_cfg(dropout=0.0))
        m = MODELS.build(cfg)
        missing, unexpected = m.load_state_dict({k[len("backbone."):]: v for k, v in sd.items()}, strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        m = m.cuda().eval()
        img = synth_image(1, 512, seed=43)
        with torch.no_grad():
            feats = m(img.cuda())
            ref = R.sam_forward(sd, img, depth=depth, global_idx=gidx, out_indices=oidx)
        for i, (f, r) in enumerate(zip(feats, ref)):
            e = rel_err(f.float().cpu(), r)
            assert e < tol, (i, e)
    finally:
        set_compute_dtype("bf16")


@pytest.mark.parametrize("mode,tol,gtol", [("f32", 3e-4, 3e-3), ("bf16", 4e-2, 1.5e-1)])
def test_sam_training_lora_grads(mode, tol, gtol):
    """SAM training (lora_sam_ms_masked.py): backward through the windowed / global attention with decomposed rel-pos bias
    (batched GEMM form) vs autograd of the oracle, SAM-H widths, depth 4 with one global block."""
    from tests.helpers import sam_state_dict
    set_compute_dtype(mode)
    try:
        depth, gidx, oidx = 4, (2,), (0, 1, 2, 3)
        sd = sam_state_dict(depth=depth, global_idx=gidx)
        cfg = dict(type="LoRABackbone", backbone=presets.sam_backbone(depth=depth, global_idx=gidx, out_indices=oidx),
                   Lora_config=presets.lora_cfg(dropout=0.0))
        m = MODELS.build(cfg)
        m.load_state_dict({k[len("backbone."):]: v for k, v in sd.items()}, strict=False)
        m = m.cuda().train()
        img = synth_image(1, 512, seed=44)
        sdo = dict(sd)
        tk = [k for k in sd if "lora_" in k]
        for k in tk:
            sdo[k] = sd[k].clone().requires_grad_(True)
        taps = R.sam_forward(sdo, img, depth=depth, global_idx=gidx, out_indices=oidx)
        gen = torch.Generator().manual_seed(10)
        dts = [torch.randn(t.shape, generator=gen) for t in taps]
        loss = sum((t * d_).sum() for t, d_ in zip(taps, dts))
        og = dict(zip(tk, torch.autograd.grad(loss, [sdo[k] for k in tk])))
        xcat, (hp, wp) = m.forward_tokens([(img.cuda(), None)])
        D = 1280
        v = xcat.float().view(1, 32, 32, 4, D)
        for i, t in enumerate(taps):
            e = rel_err(v[:, :, :, i].permute(0, 3, 1, 2).cpu(), t.detach())
            assert e < tol, (i, e)
        dx = torch.stack([d_.permute(0, 2, 3, 1) for d_ in dts], dim=3).reshape(1024, 4 * D).to(xcat.dtype).cuda()
        xcat.backward(dx)
        n = 0
        for name, p in m.named_parameters():
            if "lora_" in name:
                e = rel_err(p.grad.cpu(), og["backbone." + name])
                assert e < gtol, (name, e)
                n += 1
        assert n == 2 * depth
    finally:
        set_compute_dtype("bf16")
