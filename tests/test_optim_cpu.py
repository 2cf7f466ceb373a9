"""Host logic of the optimiser / runner side (CPU): the product's parameter groups against the oracle's restatement of
PEFTOptimWrapperConstructor.add_params (rein/optimizers/peft_optimizer_constructor.py:25-147), PolyLR, gradient-production
order of the flat buffer, checkpoint discovery for --resume."""
import os

import torch

import vfmseg_amd  # noqa: F401
from oracle import torch_ref as R
from vfmseg_amd import presets
from vfmseg_amd.optim import PolyLR, param_options, production_order
from vfmseg_amd.registry import MODELS
from vfmseg_amd.runner import find_latest_checkpoint


def _model(depth=2):
    cfg = presets.dinov2_ms_masked(depth=depth)
    cfg["backbone"]["backbone"]["out_indices"] = [0, 1]
    return MODELS.build(cfg).train()


def test_param_options_match_the_oracle_groups():
    m = _model()
    oc = presets.optim_cfg()["optim_wrapper"]
    opts = param_options(m, oc["optimizer"]["lr"], oc["optimizer"]["weight_decay"], oc["paramwise_cfg"])
    named = {n: p for n, p in m.named_parameters() if p.requires_grad}
    assert set(opts) == set(named) and len(opts) > 80
    assert all(("lora_" in n) for n in named if n.startswith("backbone.")), "only LoRA factors train in the backbone"
    n_wd0 = 0
    for n, (lr_mult, wd) in opts.items():
        want_lr, want_wd = R.param_group_options(n, R.key_is_norm(n), base_lr=1.0, base_wd=0.05, custom_keys=R.DG_CUSTOM_KEYS)
        assert (lr_mult, wd) == (want_lr, want_wd), (n, lr_mult, wd, want_lr, want_wd)
        n_wd0 += wd == 0.0
    # spot checks spelled out: norms decay-free, everything else (biases, LoRA, mask_token) decays
    assert opts["decode_head.fusion_conv.gn.weight"][1] == 0.0 and opts["decode_head.output_upscaling.1.bias"][1] == 0.0
    assert opts["aux_decoder.transformer_decoder.transformer_blocks.0.norm2.weight"][1] == 0.0
    assert opts["aux_decoder.seg_logits_embed.4.weight"][1] == 0.0
    assert opts["decode_head.conv_seg.bias"][1] == 0.05 and opts["aux_decoder.transformer_decoder.mask_token"][1] == 0.05
    assert opts["backbone.model.base_model.model.blocks.0.attn.qkv.lora_A.default.weight"] == (1.0, 0.05)
    assert 20 < n_wd0 < len(opts) // 2


def test_polylr_matches_closed_form():
    s = PolyLR(1e-4, power=0.9, eta_min=0.0, begin=0, end=40000)
    for t in (0, 1, 50, 20000, 39999, 40000, 50000):
        assert abs(s.lr(t) - R.poly_lr(1e-4, t)) < 1e-18
    assert s.lr(0) == 1e-4 and s.lr(40000) == 0.0


def test_production_order_follows_backward():
    m = _model(depth=3)
    names = production_order([n for n, p in m.named_parameters() if p.requires_grad])
    groups = ["aux" if n.startswith("aux_decoder") else "lin" if n.startswith("decode_head") else "lora" for n in names]
    first = {g: groups.index(g) for g in ("aux", "lin", "lora")}
    last = {g: len(groups) - 1 - groups[::-1].index(g) for g in ("aux", "lin", "lora")}
    assert last["aux"] < first["lin"] and last["lin"] < first["lora"]          # VFMHead -> LinearHead -> LoRA
    blocks = [int(n.split("blocks.")[1].split(".")[0]) for n in names if "lora_" in n]
    assert blocks == sorted(blocks, reverse=True)                                # LoRA layers L-1 .. 0


def test_find_latest_checkpoint_orders_numerically(tmp_path):
    assert find_latest_checkpoint(str(tmp_path / "missing")) is None
    for it in (4000, 8000, 12000, 40000):
        torch.save({}, tmp_path / f"iter_{it}.pth")
    (tmp_path / "iter_notanumber.pth").write_text("x")
    assert os.path.basename(find_latest_checkpoint(str(tmp_path))) == "iter_40000.pth"   # a string sort would pick iter_8000
    (tmp_path / "last_checkpoint").write_text(str(tmp_path / "iter_12000.pth"))
    assert os.path.basename(find_latest_checkpoint(str(tmp_path))) == "iter_12000.pth"   # the pointer file wins (mmengine)
    (tmp_path / "last_checkpoint").write_text(str(tmp_path / "gone.pth"))
    assert os.path.basename(find_latest_checkpoint(str(tmp_path))) == "iter_40000.pth"


class _StubOpt:
    """What AmpOptimWrapper needs of FusedAdamW: the flat gradient buffer, step(lr, grad_scale, zero_grad), zero_grad()."""

    def __init__(self, n=8):
        self.w = torch.zeros(n)
        self.gflat = torch.zeros(n)
        self.param_groups = [dict(lr=0.0)]
        self.steps = []
        self.step_count = 0

    def step(self, lr=None, grad_scale=1.0, zero_grad=False, skip_flag=None):
        self.step_count += 1
        if skip_flag is not None and int(skip_flag.item()):   # vfm_adamw_guarded: parameters and moments untouched, gradients cleared
            if zero_grad:
                self.gflat.zero_()
            return
        self.steps.append(float(grad_scale))
        self.w -= 0.1 * self.gflat * grad_scale
        if zero_grad:
            self.gflat.zero_()

    def zero_grad(self):
        self.gflat.zero_()


class _Loss:
    """Stands in for the loss tensor: backward() deposits grad * (whatever the loss was multiplied by) into the stub's buffer."""

    def __init__(self, opt, grad, mult=1.0):
        self.opt, self.grad, self.mult = opt, grad, mult

    def __mul__(self, k):
        return _Loss(self.opt, self.grad, self.mult * k)

    def backward(self):
        self.opt.gflat += self.grad * self.mult


def test_amp_optim_wrapper_follows_gradscaler():
    # mmengine AmpOptimWrapper / torch GradScaler semantics (tools/train.py:87-102): scaled backward, un-scaled step, a step with
    # inf / NaN gradients is skipped and halves the scale, `growth_interval` good steps double it; the iteration counter (PolyLR)
    # advances either way
    from vfmseg_amd.optim import AmpOptimWrapper
    opt = _StubOpt()
    ow = AmpOptimWrapper(opt, None, None, loss_scale=dict(init_scale=1024.0, growth_interval=3))
    g = torch.arange(8, dtype=torch.float32)
    ow.update_params(_Loss(opt, g))
    assert opt.steps == [1.0 / 1024.0] and torch.allclose(opt.w, -0.1 * g) and ow.scale == 1024.0 and ow.iter == 1
    bad = g.clone()
    bad[3] = float("inf")
    w_before = opt.w.clone()
    ow.update_params(_Loss(opt, bad))
    assert len(opt.steps) == 1 and torch.equal(opt.w, w_before) and ow.scale == 512.0 and ow.skipped == 1 and ow.iter == 2
    assert float(opt.gflat.abs().sum()) == 0.0          # the poisoned gradients are cleared
    for _ in range(3):
        ow.update_params(_Loss(opt, g))
    assert ow.scale == 1024.0 and ow.growth_tracker == 0 and len(opt.steps) == 4 and opt.steps[-1] == 1.0 / 512.0
    sd = ow.state_dict()
    ow2 = AmpOptimWrapper(_StubOpt(), None, None, loss_scale="dynamic")
    assert ow2.scale == 65536.0
    ow2.load_state_dict(sd)
    assert ow2.scale == 1024.0 and ow2.iter == 5
    # in-process resume (Runner.resume: optimizer.load_state_dict, then the wrapper's): a wrapper that has already trained holds a
    # device-side state with ITS step count; loading must not read that back over the count the optimiser has just restored
    ow._state, ow._stale = torch.tensor([4096.0, 2.0, 777.0, 9.0]), True
    opt.step_count = 5
    ow.load_state_dict(sd)
    assert opt.step_count == 5 and ow.scale == 1024.0 and ow.growth_tracker == 0 and ow._state is None
    # FusedAdamW.state_dict() goes through the wrapper's sync (the stub borrows the hook the wrapper installs on its optimiser)
    ow._state, ow._stale = torch.tensor([1024.0, 0.0, 6.0, 1.0]), True
    opt._amp_sync()
    assert opt.step_count == 6
    fixed = AmpOptimWrapper(_StubOpt(), None, None, loss_scale=128.0)
    fixed.update_params(_Loss(fixed.optimizer, bad))
    assert fixed.scale == 128.0 and fixed.skipped == 1   # a static scale never moves
    import pytest
    # autocast dtype as in mmengine: None / 'float16' -> fp16 (what the reference's `--amp` runs: precision mode "fp16", the twin
    # library), 'bfloat16' -> bf16; anything else is not an autocast dtype
    assert AmpOptimWrapper(_StubOpt(), None, None).mode == "fp16" and AmpOptimWrapper(_StubOpt(), None, None, dtype="float16").dtype == torch.float16
    assert AmpOptimWrapper(_StubOpt(), None, None, dtype="bfloat16").mode == "bf16"
    with pytest.raises(NotImplementedError):
        AmpOptimWrapper(_StubOpt(), None, None, dtype="float64")


def test_amp_loss_scale_schedule_equals_torch_gradscaler():
    """The wrapper's scale / growth-tracker trajectory against torch.amp.GradScaler itself (what mmengine's AmpOptimWrapper holds)
    over a sequence of good and overflowing steps, and the skip decisions (the parameter moves exactly when torch's optimiser steps)."""
    from vfmseg_amd.optim import AmpOptimWrapper
    seq = [0, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0]
    scaler = torch.amp.GradScaler("cpu", init_scale=1024.0, growth_interval=3)
    p = torch.nn.Parameter(torch.ones(8))
    topt = torch.optim.SGD([p], lr=0.1)
    opt = _StubOpt()
    ow = AmpOptimWrapper(opt, None, None, loss_scale=dict(init_scale=1024.0, growth_interval=3))
    for i, bad in enumerate(seq):
        g = torch.arange(8, dtype=torch.float32) + 1
        if bad:
            g[i % 8] = float("inf") if i % 2 else float("nan")
        topt.zero_grad()
        scaler.scale((p * g).sum()).backward()
        before = p.detach().clone()
        scaler.step(topt)
        scaler.update()
        nsteps = len(opt.steps)
        ow.update_params(_Loss(opt, g))
        assert ow.scale == scaler.get_scale() and ow.growth_tracker == int(scaler._growth_tracker.item()), (i, ow.scale, scaler.get_scale())
        assert (len(opt.steps) > nsteps) == (not torch.equal(before, p.detach())) == (not bad)
    assert ow.skipped == sum(seq) and ow.iter == len(seq)
