"""Optimiser side of the hot path: PEFTOptimWrapperConstructor (rein/optimizers/peft_optimizer_constructor.py:18-170),
a fused multi-tensor AdamW over ONE flat fp32 buffer (vfm_adamw) and PolyLR
(configs/dg/gta2citys/dg_lora_dinov2_ms_masked.py:10-29).

All trainable parameters are re-pointed into a flat buffer laid out in gradient-production order
(VFMHead -> LinearHead -> LoRA layers L-1..0), and so are their gradients: the data-parallel all-reduce
(vfmseg_amd.parallel) then works on contiguous bucket slices with no packing copies.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .registry import OPTIM_WRAPPER_CONSTRUCTORS

_NORM_TYPES = (nn.modules.batchnorm._BatchNorm, nn.modules.instancenorm._InstanceNorm, nn.GroupNorm, nn.LayerNorm)


def param_options(model, base_lr, base_wd, paramwise_cfg):
    """-> {param_name: (lr_mult, weight_decay)} for every trainable parameter, following add_params():
    a custom key (longest first, substring of the full name) wins; otherwise parameters of norm modules get
    weight_decay * norm_decay_mult; everything else (incl. biases, LoRA factors, mask_token) the base values."""
    custom = (paramwise_cfg or {}).get("custom_keys", {})
    keys = sorted(sorted(custom.keys()), key=len, reverse=True)
    norm_decay_mult = (paramwise_cfg or {}).get("norm_decay_mult", None)
    bias_lr_mult = (paramwise_cfg or {}).get("bias_lr_mult", None)
    bias_decay_mult = (paramwise_cfg or {}).get("bias_decay_mult", None)
    out = {}

    def visit(module, prefix):
        is_norm = isinstance(module, _NORM_TYPES)
        for name, p in module.named_parameters(recurse=False):
            if not p.requires_grad:
                continue
            full = f"{prefix}.{name}" if prefix else name
            lr_mult, wd = 1.0, base_wd
            for k in keys:
                if k in f"{prefix}.{name}":
                    lr_mult = custom[k].get("lr_mult", 1.0)
                    wd = base_wd * custom[k].get("decay_mult", 1.0)
                    break
            else:
                if name == "bias" and not is_norm and bias_lr_mult is not None:
                    lr_mult = bias_lr_mult
                if is_norm and norm_decay_mult is not None:
                    wd = base_wd * norm_decay_mult
                elif name == "bias" and bias_decay_mult is not None:
                    wd = base_wd * bias_decay_mult
            out[full] = (lr_mult, wd)
        for cn, child in module.named_children():
            visit(child, f"{prefix}.{cn}" if prefix else cn)

    visit(model, "")
    return out


def production_order(names):
    """Gradient-production order of backward: aux_decoder, decode_head, then LoRA blocks from the last to the first."""
    def key(n):
        if "lora_" in n:
            try:
                blk = int(n.split("blocks.")[1].split(".")[0])
            except Exception:
                blk = 0
            return (2, -blk, n)
        if n.startswith("aux_decoder"):
            return (0, 0, n)
        return (1, 0, n)
    return sorted(names, key=key)


# bumped by every FusedAdamW.step(): caches derived from parameter values (merged LoRA weights for inference) key on it
PARAM_EPOCH = [0]

class PolyLR:
    """mmengine PolyLR(by_epoch=False): lr_t = (base - eta_min) * (1 - t/T)^power + eta_min, t = steps taken."""

    def __init__(self, base_lr, power=0.9, eta_min=0.0, begin=0, end=40000, **kw):
        self.base_lr, self.power, self.eta_min, self.begin, self.end = base_lr, power, eta_min, begin, end

    def lr(self, t):
        t = min(max(t - self.begin, 0), self.end - self.begin)
        return (self.base_lr - self.eta_min) * (1 - t / (self.end - self.begin)) ** self.power + self.eta_min


class FusedAdamW:
    def __init__(self, model, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8, paramwise_cfg=None):
        self.lr, self.betas, self.eps = lr, tuple(betas), eps
        opts = param_options(model, lr, weight_decay, paramwise_cfg)
        named = dict(model.named_parameters())
        self.names = production_order([n for n in named if named[n].requires_grad])
        assert set(self.names) == set(opts), "option table / trainable set mismatch"
        self.params = [named[n] for n in self.names]
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        # every parameter starts on a 64-byte boundary of the flat buffers: bias / gamma slices then satisfy the 16-byte
        # operand alignment of the vector GEMM epilogues (a 19-float conv_seg bias used to knock every later bias off it and
        # sent a dozen head GEMMs per step down the scalar epilogue); the pad elements stay zero (zero grad -> zero update)
        self.sizes = sizes
        self.offsets = [0]
        for s in sizes:
            self.offsets.append((self.offsets[-1] + s + 15) // 16 * 16)
        n = self.offsets[-1]
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, a, sz in zip(self.params, self.offsets[:-1], sizes):
                b = a + sz
                view = self.flat[a:b].view(p.shape)
                ops.cast(p.detach().reshape(1, -1).contiguous(), self.flat[a:b].view(1, -1))
                p.data = view
                p.grad = self.gflat[a:b].view(p.shape)
                p._vfm_direct_grad = True   # backward kernels accumulate straight into the flat buffer
        self.seg_start = torch.tensor(self.offsets[:-1], dtype=torch.int64, device=dev)
        self.seg_lr = torch.tensor([opts[nm][0] for nm in self.names], dtype=torch.float32, device=dev)
        self.seg_wd = torch.tensor([opts[nm][1] for nm in self.names], dtype=torch.float32, device=dev)
        self.step_count = 0
        self.param_groups = [dict(lr=lr)]

    def bucket_slices(self):
        """[(name, start, end)] contiguous gradient slices in production order: aux_decoder, decode_head, lora."""
        cuts, cur = [], None
        for nm, a in zip(self.names, self.offsets[:-1]):
            g = "lora" if "lora_" in nm else ("aux_decoder" if nm.startswith("aux_decoder") else "decode_head")
            if g != cur:
                cuts.append([g, a, a])
                cur = g
            cuts[-1][2] = a + dict(zip(self.names, [p.numel() for p in self.params]))[nm]
        return [tuple(c) for c in cuts]

    def step(self, lr=None, grad_scale=1.0, zero_grad=False, skip_flag=None, amp_state=None):
        """zero_grad=True clears the flat gradient buffer in the same pass (OptimWrapper.update_params: step, then zero_grad).
        skip_flag / amp_state (AmpOptimWrapper, device tensors): a non-zero flag makes the launch a no-op on parameters and moments;
        with amp_state the loss scale and the step count of the bias corrections are read on the device and `step_count` is the
        wrapper's to maintain (AmpOptimWrapper.sync)."""
        if amp_state is None:
            self.step_count += 1
        PARAM_EPOCH[0] += 1  # the fused kernel rewrites the parameters behind torch's version counters
        lr = self.lr if lr is None else lr
        self.param_groups[0]["lr"] = lr
        ops.adamw(self.flat, self.gflat, self.m, self.v, self.seg_start, self.seg_lr, self.seg_wd, lr, self.betas, self.eps,
                  self.step_count, grad_scale, zero_grad=zero_grad, vec4=True, skip=skip_flag, amp_state=amp_state)   # offsets are multiples of 16 floats
        self._grads_cleared = bool(zero_grad)

    def zero_grad(self):
        if not getattr(self, "_grads_cleared", False):
            self.gflat.zero_()
        self._grads_cleared = False
        for p, a, sz in zip(self.params, self.offsets[:-1], self.sizes):
            if p.grad is None or p.grad.data_ptr() != self.gflat.data_ptr() + a * 4:
                p.grad = self.gflat[a:a + sz].view(p.shape)

    def state_dict(self):
        sync = getattr(self, "_amp_sync", None)
        if sync is not None:   # under AmpOptimWrapper on the GPU the step count lives in the device-side scaler state
            sync()
        return dict(step=self.step_count, m=self.m, v=self.v, names=self.names)

    def load_state_dict(self, sd):
        if list(sd.get("names", self.names)) != list(self.names) or sd["m"].numel() != self.m.numel():
            raise ValueError("optimizer state does not match this model: the flat m/v buffers are laid out by parameter name "
                             f"({len(sd.get('names', []))} saved names / {sd['m'].numel()} elements vs {len(self.names)} / {self.m.numel()})")
        self.step_count = sd["step"]
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])


class OptimWrapper:
    """mmengine OptimWrapper.update_params: backward -> (grad sync) -> step -> zero_grad; scheduler stepped by iter."""

    def __init__(self, optimizer, scheduler=None, grad_sync=None):
        self.optimizer, self.scheduler, self.grad_sync = optimizer, scheduler, grad_sync
        self.iter = 0

    @staticmethod
    def _backward(loss):
        # one device per process: the autograd engine's hand-off to its device thread buys nothing and costs ~15 % of the step's
        # host enqueue time (10.4 vs 12.2 ms of a 15.6-ms step; tools/scratch/_host_profile_step.py)
        with torch.autograd.set_multithreading_enabled(False):
            loss.backward()
        from .functional import join_wgrad_stream
        join_wgrad_stream()   # the heads' weight gradients were issued on a side stream: everything below reads the flat gradient buffer

    def update_params(self, loss):
        self._backward(loss)
        if self.grad_sync is not None:
            self.grad_sync()
        lr = self.scheduler.lr(self.iter) if self.scheduler is not None else None
        self.optimizer.step(lr, grad_scale=getattr(self.grad_sync, "post_scale", 1.0), zero_grad=True)
        self.optimizer.zero_grad()
        self.iter += 1

    def get_lr(self):
        return self.optimizer.param_groups[0]["lr"]

    def state_dict(self):
        return dict(iter=self.iter)

    def load_state_dict(self, sd):
        self.iter = int(sd.get("iter", self.iter))


class AmpOptimWrapper(OptimWrapper):
    """mmengine AmpOptimWrapper (what tools/train.py:87-102 switches to under `--amp`): the loss is multiplied by a loss scale
    before backward, the gradients are un-scaled inside the optimiser step, a step whose gradients hold an inf / NaN is skipped,
    and with `loss_scale='dynamic'` the scale follows torch.cuda.amp.GradScaler's schedule (init 2**16, x0.5 after a skipped
    step, x2 after `growth_interval` = 2000 consecutive good ones).  A number or a dict(init_scale=, growth_factor=,
    backoff_factor=, growth_interval=) is accepted as in mmengine.

    Autocast dtype (`dtype`, as in mmengine): None / 'float16' = fp16, the CUDA autocast default and what the reference's `--amp`
    runs in - here the precision mode "fp16" (libvfmseg_hip_f16.so: the same kernels with fp16 storage and the fp16 MFMA, fp32
    accumulation, residual stream, statistics and losses); 'bfloat16' = the bf16 mode, where the scale never has to back off in
    practice.  The wrapper records the dtype (`self.dtype`, `self.mode`); the Runner / tools/train.py put the engine into that mode
    (the role of mmengine's optim_context autocast).  The un-scale costs nothing: 1/scale rides on the fused AdamW kernel's grad_scale."""

    _DTYPES = {None: torch.float16, "float16": torch.float16, "fp16": torch.float16, "half": torch.float16, torch.float16: torch.float16,
               "bfloat16": torch.bfloat16, "bf16": torch.bfloat16, torch.bfloat16: torch.bfloat16}

    def __init__(self, optimizer, scheduler=None, grad_sync=None, loss_scale="dynamic", dtype=None):
        super().__init__(optimizer, scheduler, grad_sync)
        if dtype not in self._DTYPES:
            raise NotImplementedError("AmpOptimWrapper: autocast dtype %r (float16 and bfloat16 are the ones torch.autocast offers on a GPU)" % (dtype,))
        self.dtype = self._DTYPES[dtype]
        self.mode = "fp16" if self.dtype == torch.float16 else "bf16"
        self._tracker, self._skipped = 0, 0
        self._state, self._stale = None, False   # device-side scaler state (built at the first GPU update_params)
        optimizer._amp_sync = self.sync          # FusedAdamW.state_dict() reads the device-side step count back through this
        self.growth_factor, self.backoff_factor, self.growth_interval = 2.0, 0.5, 2000
        self.dynamic = True
        if loss_scale == "dynamic":
            self.scale = 2.0 ** 16
        elif isinstance(loss_scale, dict):
            self.scale = float(loss_scale.get("init_scale", 2.0 ** 16))
            self.growth_factor = float(loss_scale.get("growth_factor", 2.0))
            self.backoff_factor = float(loss_scale.get("backoff_factor", 0.5))
            self.growth_interval = int(loss_scale.get("growth_interval", 2000))
        elif isinstance(loss_scale, (int, float)):
            self.scale, self.dynamic = float(loss_scale), False
        else:
            raise TypeError("loss_scale must be 'dynamic', a number or a dict, got %r" % (loss_scale,))

    def update_params(self, loss):
        """GradScaler.scale(loss).backward(); unscale_; step (skipped on inf / NaN); update.
        On the GPU the scaler's state {scale, growth tracker, optimiser steps, skipped steps} lives in a 4-float DEVICE tensor: the loss
        is multiplied by the device scale, the overflow flag is formed on the device, the fused AdamW launch reads flag, scale and step
        count there (vfm_adamw_guarded) and vfm_amp_update applies GradScaler.update() - the host never waits for the backward pass
        (torch's GradScaler.step pays a found_inf.item() per step with a non-fused optimiser: the host then cannot run ahead of the
        GPU, which cost ~1 ms of a 15.3-ms step here).  `scale`, `skipped`, `growth_tracker` and optimizer.step_count are read back
        on demand (sync())."""
        g = self.optimizer.gflat
        if not g.is_cuda:                    # host path (CPU tensors: tests of the schedule against torch's GradScaler)
            self._backward(loss * self._scale)
            if self.grad_sync is not None:
                self.grad_sync()
            found_inf = not bool(torch.isfinite(g.sum()).item())
            flag = torch.tensor([int(found_inf)], dtype=torch.int32)
            lr = self.scheduler.lr(self.iter) if self.scheduler is not None else None
            self.optimizer.step(lr, grad_scale=getattr(self.grad_sync, "post_scale", 1.0) / self._scale, zero_grad=True, skip_flag=flag)
            self.optimizer.zero_grad()
            if found_inf:
                self._skipped += 1
                self.optimizer.step_count -= 1   # torch's AdamW does not count a skipped step (bias correction)
            self._update_scale(found_inf)
            self.iter += 1
            return
        if self._state is None:
            self._state = torch.tensor([self._scale, float(self._tracker), float(self.optimizer.step_count), float(self._skipped)],
                                       dtype=torch.float32, device=g.device)
        self._backward(loss * self._state[0])
        if self.grad_sync is not None:
            self.grad_sync()
        # after the all-reduce every rank sees the same sums, hence the same decision (an inf / NaN survives the reduction)
        flag = (~torch.isfinite(g.sum())).to(torch.int32).reshape(1)
        lr = self.scheduler.lr(self.iter) if self.scheduler is not None else None
        self.optimizer.step(lr, grad_scale=getattr(self.grad_sync, "post_scale", 1.0), zero_grad=True, skip_flag=flag, amp_state=self._state)
        self.optimizer.zero_grad()
        ops.amp_update(flag, self._state, self.growth_factor, self.backoff_factor, self.growth_interval, self.dynamic)
        self._stale = True
        self.iter += 1

    def sync(self):
        """Read the device-side scaler state back (one host wait): scale, growth tracker, skipped steps, optimizer.step_count."""
        if self._stale:
            st = self._state.cpu().tolist()
            self._scale, self._tracker, self._skipped = float(st[0]), int(st[1]), int(st[3])
            self.optimizer.step_count = int(st[2])
            self._stale = False

    def _push(self):
        """The host copies changed (load_state_dict, a scale set by hand): the device tensor is rebuilt at the next update_params."""
        self._state, self._stale = None, False

    @property
    def scale(self):
        self.sync()
        return self._scale

    @scale.setter
    def scale(self, v):
        self.sync()
        self._scale = float(v)
        self._push()

    @property
    def skipped(self):
        self.sync()
        return self._skipped

    @skipped.setter
    def skipped(self, v):
        self.sync()
        self._skipped = int(v)
        self._push()

    @property
    def growth_tracker(self):
        self.sync()
        return self._tracker

    @growth_tracker.setter
    def growth_tracker(self, v):
        self.sync()
        self._tracker = int(v)
        self._push()

    def _update_scale(self, found_inf):
        """torch GradScaler.update(): x backoff after a skipped step, x growth after growth_interval consecutive good ones."""
        if not self.dynamic:
            return
        if found_inf:
            self._scale *= self.backoff_factor
            self._tracker = 0
        else:
            self._tracker += 1
            if self._tracker == self.growth_interval:
                self._scale *= self.growth_factor
                self._tracker = 0

    def state_dict(self):
        self.sync()
        return dict(iter=self.iter, loss_scaler=dict(scale=self.scale, growth_factor=self.growth_factor, backoff_factor=self.backoff_factor,
                                                     growth_interval=self.growth_interval, _growth_tracker=self.growth_tracker))

    def load_state_dict(self, sd):
        # The device-side state of THIS process is dropped first, un-read: reading it back here (the `scale` / `growth_tracker`
        # setters do) would write the old run's step count over the one Runner.resume has just restored into the optimiser.
        self._push()
        super().load_state_dict(sd)
        ls = sd.get("loss_scaler")
        if ls:
            self._scale, self._tracker = float(ls["scale"]), int(ls.get("_growth_tracker", 0))
        # (the device state is rebuilt from the host copies - scale, tracker, optimizer.step_count - at the next update_params)


@OPTIM_WRAPPER_CONSTRUCTORS.register_module()
class PEFTOptimWrapperConstructor:
    def __init__(self, optim_wrapper_cfg, paramwise_cfg=None):
        self.cfg = dict(optim_wrapper_cfg)
        self.cfg.pop("constructor", None)
        self.paramwise_cfg = paramwise_cfg if paramwise_cfg is not None else self.cfg.pop("paramwise_cfg", None)
        self.optimizer_cfg = dict(self.cfg["optimizer"])

    def __call__(self, model, param_scheduler=None):
        model.train()
        oc = dict(self.optimizer_cfg)
        typ = oc.pop("type", "AdamW")
        if typ != "AdamW":
            raise NotImplementedError("the hot path ships the reference's optimiser (AdamW) only")
        opt = FusedAdamW(model, oc.get("lr", 1e-3), oc.get("weight_decay", 0.01), oc.get("betas", (0.9, 0.999)),
                         oc.get("eps", 1e-8), self.paramwise_cfg)
        sched = None
        if param_scheduler:
            sc = dict(param_scheduler[0] if isinstance(param_scheduler, (list, tuple)) else param_scheduler)
            assert sc.pop("type", "PolyLR") == "PolyLR"
            sc.pop("by_epoch", None)
            sched = PolyLR(opt.lr, **sc)
        wtype = self.cfg.get("type", "OptimWrapper")
        if wtype == "AmpOptimWrapper":
            return AmpOptimWrapper(opt, sched, loss_scale=self.cfg.get("loss_scale", "dynamic"), dtype=self.cfg.get("dtype"))
        if wtype != "OptimWrapper":
            raise NotImplementedError("optim_wrapper type %r (OptimWrapper and AmpOptimWrapper are the ones tools/train.py uses)" % (wtype,))
        return OptimWrapper(opt, sched)
