"""vfm_sam_attn_flash_fwd (csrc/sam_flash.hip) against a float64 restatement of sam_vit.py:273-298 + :301-356 + :359-430 on
random qkv: windowed attention on the zero-padded grid (padded tokens are keys whose k / v equal the projection bias; padded
queries are dropped) and global attention with a re-interpolated 127-entry table; then the whole SAM-H engine with the flash
path on against the materialised path (same goldens as tests/test_fulldepth_gpu.py cover it at depth 32)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import vfmseg_amd  # noqa: E402,F401
from oracle import torch_ref as R  # noqa: E402
from tests.helpers import rel_err  # noqa: E402
from vfmseg_amd import ops  # noqa: E402


def _ref(qkv, bias, rel_h, rel_w, nimg, G, S, H, d):
    """fp64: [nimg*G*G, 3*H*d] -> [nimg*G*G, H*d]"""
    C = H * d
    x = (qkv if qkv.dtype == torch.float64 else qkv.double()).view(nimg, G, G, 3 * C)
    if S < G:
        pad = (S - G % S) % S
        Gp = G + pad
        xp = bias.double().view(1, 1, 1, 3 * C).expand(nimg, Gp, Gp, 3 * C).clone()    # padded tokens: qkv = bias
        xp[:, :G, :G] = x
        w = xp.view(nimg, Gp // S, S, Gp // S, S, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(-1, S * S, 3, H, d)
    else:
        Gp = G
        w = x.reshape(nimg, S * S, 3, H, d)
    q, k, v = w.permute(2, 0, 3, 1, 4).unbind(0)                                         # [nb, H, S*S, d]
    attn = (q * d ** -0.5) @ k.transpose(-2, -1)
    rh, rw = R.sam_rel_pos(S, S, rel_h.double()), R.sam_rel_pos(S, S, rel_w.double())
    rq = q.reshape(q.shape[0], H, S, S, d)
    attn = attn.view(-1, H, S, S, S, S) + torch.einsum("bnhwc,hkc->bnhwk", rq, rh)[..., None] + torch.einsum("bnhwc,wkc->bnhwk", rq, rw)[..., None, :]
    o = attn.view(-1, H, S * S, S * S).softmax(-1) @ v                                    # [nb, H, S*S, d]
    o = o.permute(0, 2, 1, 3).reshape(-1, S, S, C)
    if S < G:
        o = o.view(nimg, Gp // S, Gp // S, S, S, C).permute(0, 1, 3, 2, 4, 5).reshape(nimg, Gp, Gp, C)[:, :G, :G]
    return o.reshape(nimg * G * G, C)


def _tables(rel, S, JP):
    """[L, d] parameter -> the kernel's bf16 [JP, d] relative-index table (get_rel_pos re-interpolation included)."""
    g = R.sam_rel_pos(S, S, rel)                                                          # [S, S, d] gathered
    t = torch.zeros(JP, rel.shape[1])
    for j in range(2 * S - 1):
        t[j] = g[max(j - (S - 1), 0), max(S - 1 - j, 0)]
    return t.bfloat16().cuda()


@pytest.mark.parametrize("S,G,nimg,L", [(14, 32, 2, 27), (32, 32, 2, 127), (14, 32, 1, 27), (14, 20, 3, 27)])
def test_sam_flash_forward_matches_reference_math(S, G, nimg, L):
    H, d = 16, 80
    g = torch.Generator().manual_seed(S + nimg)
    qkv = (torch.randn(nimg * G * G, 3 * H * d, generator=g) * 1.5).bfloat16()
    bias = torch.randn(3 * H * d, generator=g) * 0.5
    rel_h, rel_w = torch.randn(L, d, generator=g) * 0.3, torch.randn(L, d, generator=g) * 0.3
    JP = 32 if S == 14 else 64
    out = torch.full((nimg * G * G, H * d), float("nan"), dtype=torch.bfloat16, device="cuda")
    for rep in range(2):
        ops.sam_attn_flash_fwd(qkv.cuda(), bias.cuda(), _tables(rel_h, S, JP), _tables(rel_w, S, JP), out, nimg, G, S, H, d, d ** -0.5)
    ref = _ref(qkv.float(), bias, rel_h, rel_w, nimg, G, S, H, d)
    assert torch.isfinite(out.float()).all()
    e = rel_err(out.float().cpu(), ref)
    print(f"[sam flash S={S}] rel err {e:.2e}")
    assert e < 2e-2, e     # bf16 P and bf16 bias columns; the materialised path has the same operand precision


@pytest.mark.parametrize("S,G,nimg,L", [(14, 32, 2, 27), (32, 32, 2, 127), (14, 32, 9, 27), (14, 20, 3, 27)])
def test_sam_flash_backward_matches_autograd(S, G, nimg, L):
    """vfm_sam_attn_flash_fwd_train + vfm_sam_attn_flash_bwd (csrc/sam_flash_bwd.hip) against float64 autograd of the same
    restatement: d(qkv) for random d(out), including the path through the decomposed rel-pos bias (frozen tables) and the windows
    with padded tokens (their k / v gradients are dropped, their queries have no output)."""
    H, d = 16, 80
    g = torch.Generator().manual_seed(100 + S + nimg)
    qkv = (torch.randn(nimg * G * G, 3 * H * d, generator=g) * 1.5).bfloat16()
    bias = torch.randn(3 * H * d, generator=g) * 0.5
    rel_h, rel_w = torch.randn(L, d, generator=g) * 0.3, torch.randn(L, d, generator=g) * 0.3
    dout = (torch.randn(nimg * G * G, H * d, generator=g)).bfloat16()
    JP = 32 if S == 14 else 64
    th, tw = _tables(rel_h, S, JP), _tables(rel_w, S, JP)
    out = torch.full((nimg * G * G, H * d), float("nan"), dtype=torch.bfloat16, device="cuda")
    lse, qext = ops.sam_attn_flash_stats(nimg, G, S, H, "cuda")
    ops.sam_attn_flash_fwd_train(qkv.cuda(), bias.cuda(), th, tw, out, lse, qext, nimg, G, S, H, d, d ** -0.5)
    out2 = torch.empty_like(out)
    ops.sam_attn_flash_fwd(qkv.cuda(), bias.cuda(), th, tw, out2, nimg, G, S, H, d, d ** -0.5)
    assert torch.equal(out, out2)                       # the training launch computes the same output
    dqkv = torch.full((nimg * G * G, 3 * H * d), float("nan"), dtype=torch.bfloat16, device="cuda")
    for rep in range(2):
        ops.sam_attn_flash_bwd(qkv.cuda(), bias.cuda(), th, tw, out, dout.cuda(), lse, qext, dqkv, nimg, G, S, H, d, d ** -0.5)
    assert torch.isfinite(dqkv.float()).all()           # every element written
    x = qkv.double().requires_grad_(True)
    ref = _ref(x, bias, rel_h, rel_w, nimg, G, S, H, d)
    ref.backward(dout.double())
    C = H * d
    got = dqkv.float().cpu()
    for name, sl in (("dq", slice(0, C)), ("dk", slice(C, 2 * C)), ("dv", slice(2 * C, 3 * C))):
        e = rel_err(got[:, sl], x.grad[:, sl])
        print(f"[sam flash bwd S={S} nimg={nimg}] {name} rel err {e:.2e}")
        assert e < 3e-2, (name, e)


def test_sam_engine_flash_equals_materialised_path():
    """SAM-H widths, depth 8 (2 global blocks), eval: taps with the flash forward vs taps with the batched-GEMM form."""
    from tests.helpers import sam_state_dict
    from vfmseg_amd import presets
    from vfmseg_amd.precision import set_compute_dtype
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.synth import synth_image
    import vfmseg_amd.sam  # noqa: F401
    set_compute_dtype("bf16")
    depth, gidx, oidx = 8, (3, 7), (1, 3, 5, 7)
    sd = sam_state_dict(depth=depth, global_idx=gidx)
    cfg = dict(type="LoRABackbone", backbone=presets.sam_backbone(depth=depth, global_idx=gidx, out_indices=oidx), Lora_config=presets.lora_cfg(dropout=0.0))
    m = MODELS.build(cfg)
    m.load_state_dict({k[len("backbone."):]: v for k, v in sd.items()}, strict=False)
    m = m.cuda().eval()
    img = synth_image(2, 512, seed=45).cuda()
    outs = {}
    try:
        for flag in ("1", "0"):
            os.environ["VFMSEG_SAM_FLASH"] = flag
            with torch.no_grad():
                outs[flag] = [f.float().cpu() for f in m(img)]
    finally:
        os.environ.pop("VFMSEG_SAM_FLASH", None)
    ref = R.sam_forward(sd, img.cpu(), depth=depth, global_idx=gidx, out_indices=oidx)
    for i in range(4):
        e_f, e_m = rel_err(outs["1"][i], ref[i]), rel_err(outs["0"][i], ref[i])
        print(f"[sam engine tap {i}] flash vs oracle {e_f:.2e}, materialised vs oracle {e_m:.2e}")
        assert e_f < 4e-2 and e_f < 2.0 * e_m + 5e-3


def test_sam_engine_flash_backward_equals_materialised_backward():
    """SAM-H widths, depth 4 (one global block), train mode, LoRA dropout off: the LoRA gradients of the engine with the flash
    forward + backward against the same engine on the materialised path (VFMSEG_SAM_FLASH=0), same input and d(taps)."""
    from tests.helpers import sam_state_dict
    from vfmseg_amd import presets
    from vfmseg_amd.precision import set_compute_dtype
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.synth import synth_image
    import vfmseg_amd.sam  # noqa: F401
    set_compute_dtype("bf16")
    depth, gidx, oidx = 4, (3,), (0, 1, 2, 3)
    sd = sam_state_dict(depth=depth, global_idx=gidx)
    cfg = dict(type="LoRABackbone", backbone=presets.sam_backbone(depth=depth, global_idx=gidx, out_indices=oidx), Lora_config=presets.lora_cfg(dropout=0.0))
    m = MODELS.build(cfg)
    m.load_state_dict({k[len("backbone."):]: v for k, v in sd.items()}, strict=False)
    m = m.cuda().train()
    for n, p_ in m.named_parameters():   # B starts at zero: give it values so that dA is not identically zero
        if "lora_B" in n:
            with torch.no_grad():
                p_.copy_(torch.randn(p_.shape, generator=torch.Generator().manual_seed(len(n))).to(p_.device) * 0.02)
    img = synth_image(2, 512, seed=46).cuda()
    grads = {}
    try:
        for flag in ("1", "0"):
            os.environ["VFMSEG_SAM_FLASH"] = flag
            for p_ in m.parameters():
                p_.grad = None
            feats = m(img)
            g = torch.Generator().manual_seed(5)
            loss = sum((f.float() * torch.randn(f.shape, generator=g).to(f.device)).sum() for f in feats)
            loss.backward()
            grads[flag] = {n: p_.grad.detach().float().cpu().clone() for n, p_ in m.named_parameters() if p_.grad is not None}
    finally:
        os.environ.pop("VFMSEG_SAM_FLASH", None)
    assert len(grads["1"]) == 2 * depth == len(grads["0"])
    for n in grads["1"]:
        e = rel_err(grads["1"][n], grads["0"][n])
        print(f"[sam engine bwd] {n}: flash vs materialised {e:.2e}")
        assert grads["0"][n].abs().max() > 0 and e < 6e-2, (n, e)
