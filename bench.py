#!/usr/bin/env python
"""Headline benchmark: train images/sec of configs/dg/gta2citys/dg_lora_dinov2_ms_masked.py (DINOv2-L + LoRA,
LinearHead + VFMHead), synthetic 19-class data, bs=2 per GPU, bf16 MFMA, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one full training iteration of the hot path on one batch: MsVFMEncoderDecoder.forward_train (one 1024^2
sample -> LR 512^2 pass + HR 512^2 crop through the 24-block backbone, both heads, two fused upsample-CE losses),
backward (LoRA + head gradients), DP gradient all-reduce, fused AdamW, zero_grad.  One "image" = one dataloader
sample.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

np = torch = None   # imported by _heavy(): the launcher parent (`python bench.py --gpus N`, no WORLD_SIZE) must stay GPU-free


def _heavy():
    global np, torch
    if torch is None:
        import numpy as _np
        import torch as _torch
        np, torch = _np, _torch

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense")
PEAK_HBM_GBS = 8000.0


def build(device, batch, seed=0, workload="ms1024", depth=None, amp=False):
    _heavy()
    import vfmseg_amd  # noqa: F401  (registers the model classes)
    from vfmseg_amd import presets
    from vfmseg_amd.optim import PEFTOptimWrapperConstructor
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.synth import synth_like

    torch.manual_seed(seed)
    kw = {} if depth is None else dict(depth=depth)   # depth: launcher / DP rehearsals only, never the headline line
    cfg = presets.dinov2_ms_masked(**kw) if workload == "ms1024" else presets.dinov2_linear(**kw)
    if depth is not None:
        assert depth % 4 == 0, "--depth: a multiple of 4 (the four feature taps sit at the quarter points)"
        cfg["backbone"]["backbone"]["out_indices"] = [depth * (i + 1) // 4 - 1 for i in range(4)]
    model = MODELS.build(cfg)
    # random-init weights of the named architecture (no checkpoints offline): key-hashed synthetic values
    sd = synth_like(model.state_dict())
    model.load_state_dict(sd)
    model = model.to(device)
    model.train()
    oc = presets.optim_cfg()
    ocw = oc["optim_wrapper"]
    if amp:   # --dtype fp16: the step the reference's `--amp` runs (fp16 autocast + dynamic loss scale, tools/train.py:87-102)
        ocw = dict(ocw, type="AmpOptimWrapper", loss_scale="dynamic")
    ow = PEFTOptimWrapperConstructor(ocw)(model, oc["param_scheduler"])
    return model, ow


def make_batch(batch, rank, step, device, size=1024):
    _heavy()
    from vfmseg_amd.segmentors import SegDataSample
    from vfmseg_amd.synth import synth_image, synth_label
    img = synth_image(batch, size, seed=100 + rank).to(device)
    lab = synth_label(batch, size, seed=100 + rank).to(device)
    return dict(inputs=img, data_samples=[SegDataSample(gt_sem_seg=lab[i]) for i in range(batch)])


class KernelTimer:
    """HIP-event timing of the two MFMA kernel families (bf16 GEMM, flash attention) on the stream they are launched on, through
    the measurement hook of vfmseg_amd.ops: every `every`-th launch is bracketed by an event pair (sampling keeps the event cost off
    the step time; a prime period cycles through every launch site over the steps).  Sums are kept per (kind, region)."""

    def __init__(self, every=31):
        self._c_every = None
        self.recs, self.plan_recs = [], []
        self.every = every
        self.on = False
        self.calls = 0

    def install(self):
        from vfmseg_amd import ops
        timer = self

        def hook(kind, flops, region):
            if not timer.on:
                return None
            timer.calls += 1
            if timer.calls % timer.every:
                return None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()

            def fin():
                e1.record()
                timer.recs.append((kind, region, flops, e0, e1))
            return fin

        ops.PROFILE = hook
        self.plan_recs = []
        self._c_every = -1

    def uninstall(self):
        from vfmseg_amd import ops
        ops.PROFILE = None
        ops.prof_config(0)

    def reset(self):
        self.recs, self.calls = [], 0
        self._pull()
        self.plan_recs = []

    def _sync_c(self):
        """The launches a launch plan issues from C (the backbone of the train step: vfm_run_plan) are sampled by the library's own
        event sampler at the same period; it follows `on`."""
        from vfmseg_amd import ops
        want = self.every if self.on else 0
        if want != self._c_every:
            ops.prof_config(want)
            self._c_every = want

    def _pull(self):
        from vfmseg_amd import ops
        try:
            self.plan_recs += [(k, "backbone", fl, ms) for k, fl, ms in ops.prof_read()]
        except Exception:   # noqa: BLE001  (library not loaded yet)
            pass

    def __setattr__(self, k, v):
        object.__setattr__(self, k, v)
        if k == "on" and getattr(self, "_c_every", None) is not None:
            self._sync_c()

    def summary(self, kinds=None, region=None, min_flops=0.0):
        self._pull()
        ok = lambda r: (kinds is None or r[0] in kinds) and (region is None or r[1] == region) and r[2] >= min_flops   # noqa: E731
        sel = [r for r in self.recs if ok(r)]
        selc = [r for r in self.plan_recs if ok(r)]
        if not sel and not selc:
            return None
        fl = sum(r[2] for r in sel) + sum(r[2] for r in selc)
        ms = sum(r[3].elapsed_time(r[4]) for r in sel) + sum(r[3] for r in selc)
        return dict(flops=fl, ms=ms, launches=len(sel) + len(selc), tflops=fl / (ms * 1e-3) / 1e12)


# forward FLOPs of the eval workloads (SURVEY 8d arithmetic): DINOv2-L 512^2 pass 722.4 G; the coarse 512x1024 pass has N = 2049
# tokens: linear parts 2 x 25.8 G, attention 4 N^2 D = 17.2 G per layer -> 24 x 68.8 = 1651 G; SAM-H 512^2 crop 1.63 T
EVAL_FLOPS = {"dinov2_ms": 1651e9 + 9 * 722.4e9, "sam_slide": 9 * 1.63e12}


def eval_leg(device, timer, iters=5):
    """BASELINE.json metric, second half: eval ms/img @1024x1024, batch 1, bf16, synthetic, inputs resident.
      dinov2 : dg_lora_dinov2_ms_masked test_cfg (ms_slide_inference: coarse 512x1024 pass + confidence-gated refinement of the
               3x3 windows; random-init logits are never confident, conf forced so that all 9 refine: the worst case)
      sam    : configs[4] = lora_sam_linear semantics (SAM-ViT-H + LoRA + LinearHead, mode slide, stride 320, crop 512: 3x3 windows)"""
    _heavy()
    import vfmseg_amd  # noqa: F401
    from vfmseg_amd import presets
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.synth import synth_image, synth_like
    res = {"metric": "eval ms/img @1024x1024 (batch 1, bf16, synthetic)", "unit": "ms/img", "higher_is_better": False}
    cfg_d = presets.dinov2_ms_masked()
    cfg_d["test_cfg"]["conf"] = 2.0
    for name, cfg, fkey in (("dinov2_ms_slide_inference", cfg_d, "dinov2_ms"), ("sam_h_slide", presets.sam_linear(), "sam_slide")):
        print(f"[bench] eval leg: {name}", file=sys.stderr, flush=True)
        model = MODELS.build(cfg)
        model.load_state_dict(synth_like(model.state_dict()), strict=False)
        model = model.to(device).eval()
        img = synth_image(1, 1024, seed=77).to(device)
        with torch.no_grad():
            for _ in range(2):
                model.predict(img)
            torch.cuda.synchronize()
            if timer is not None:
                timer.reset()
                timer.on = True
            t0 = time.perf_counter()
            for _ in range(iters):
                model.predict(img)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / iters
            if timer is not None:
                timer.on = False
        r = {"value": round(ms, 2), "iters": iters, "model_tflops": round(EVAL_FLOPS[fkey] / (ms * 1e-3) / 1e12, 1),
             "model_frac": round(EVAL_FLOPS[fkey] / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)}
        if hasattr(model, "last_refined"):
            r["refined_windows"] = len(model.last_refined)
        if timer is not None:
            blk = timer.summary(kinds=("gemm", "attn_fwd"), region="backbone")
            allg = timer.summary(kinds=("gemm",))
            if blk is not None:
                r["roofline"] = {"bound": "mfma", "achieved": round(blk["tflops"], 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(blk["tflops"] / PEAK_BF16_TFLOPS, 4),
                                 "kernel": "backbone GEMM + flash-attention launches (sampled HIP events)" if fkey == "dinov2_ms" else
                                           "backbone GEMM + SAM flash-attention launches (sampled HIP events; attention FLOPs = q.k and p.v only)",
                                 "kernel_ms_per_img_est": round(blk["ms"] * timer.every / iters, 2),
                                 "all_gemm_tflops": None if allg is None else round(allg["tflops"], 1)}
        res[name] = r
        del model
        torch.cuda.empty_cache()
    return res


def other_backbones_leg(device, batch, timer, steps=10, warmup=3):
    """BASELINE configs[3]: EVA02-ViT-L/14 + LoRA + LinearHead + VFMHead (lora_eva02_ms_masked), the same train step as the headline
    (synthetic 1024^2 samples -> LR + HR 512^2 passes, bs 2, bf16): images/s and the MFMA-kernel fraction of its backbone."""
    _heavy()
    import vfmseg_amd  # noqa: F401
    from vfmseg_amd import functional as Fh, presets
    from vfmseg_amd.optim import PEFTOptimWrapperConstructor
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.synth import synth_like
    out = {}
    for key, preset, label in (("eva02_ms_masked", "eva02_ms_masked", "configs[3]: EVA02-L/14 + LoRA(q,k,v,attn.proj) + LinearHead + VFMHead, 512^2 passes"),):
        print(f"[bench] other-backbones leg: {key}", file=sys.stderr, flush=True)
        model = MODELS.build(getattr(presets, preset)())
        sd = {k: v for k, v in synth_like(model.state_dict()).items() if "rope." not in k}
        model.load_state_dict(sd, strict=False)
        model = model.to(device).train()
        oc = presets.optim_cfg()
        ow = PEFTOptimWrapperConstructor(oc["optim_wrapper"])(model, oc["param_scheduler"])
        Fh.manual_seed(7)
        data = make_batch(batch, 0, 0, device)
        for _ in range(warmup):
            model.train_step(data, ow)
        torch.cuda.synchronize()
        if timer is not None:
            timer.reset()
            timer.on = True
        t0 = time.perf_counter()
        for _ in range(steps):
            model.train_step(data, ow)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        r = {"what": label, "train_images_per_s": round(batch / dt, 2), "ms_per_step": round(dt * 1e3, 3), "steps": steps, "batch": batch, "dtype": "bf16"}
        if timer is not None:
            timer.on = False
            blk = timer.summary(kinds=("gemm", "gemm_tn", "attn_fwd", "attn_bwd"), region="backbone")
            if blk is not None:
                r["blocks_tflops"] = round(blk["tflops"], 1)
                r["blocks_frac"] = round(blk["tflops"] / PEAK_BF16_TFLOPS, 4)
                r["kernel_ms_per_step_est"] = round(blk["ms"] * timer.every / steps, 3)
        out[key] = r
        del model, ow, data
        torch.cuda.empty_cache()
    return out


def parity_leg(device, batch, steps=3, iters=2, modes=("bf16x3", "f32")):
    """The two modes whose results meet north_star's tolerance (logits <= 1e-3 rel, argmax mismatches only on near-ties against the
    reference's fp32 CPU path: tests/test_model_gpu.py::test_ms_inference_matches_reference_golden[f32|bf16x3],
    test_train_step_matches_reference_goldens[f32|bf16x3]) on the same two workloads as the bf16 headline:
      f32    - exact-fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate) and exact-fp32 attention
      bf16x3 - the same fp32 data path with every GEMM as ONE bf16 MFMA GEMM over split operands (hi / lo halves, K' = 3 K)"""
    _heavy()
    import vfmseg_amd  # noqa: F401
    from vfmseg_amd import presets
    from vfmseg_amd.precision import set_compute_dtype
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.synth import synth_image, synth_like
    out = {"what": "in-tolerance modes (the configurations the 1e-3 / argmax parity tests run in), same workloads as the bf16 lines"}
    for mode in modes:
        res = {}
        set_compute_dtype(mode)
        try:
            print(f"[bench] parity-mode leg: {mode} train step", file=sys.stderr, flush=True)
            model, ow = build(device, batch, amp=(mode == "fp16"))
            data = make_batch(batch, 0, 0, device)
            for _ in range(2 if steps < 10 else 4):
                model.train_step(data, ow)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                model.train_step(data, ow)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res["train_images_per_s"] = round(batch * steps / dt, 3)
            res["train_ms_per_step"] = round(1e3 * dt / steps, 2)
            res["train_steps"] = steps
            del model, ow, data
            torch.cuda.empty_cache()
            print(f"[bench] parity-mode leg: {mode} eval", file=sys.stderr, flush=True)
            cfg_d = presets.dinov2_ms_masked()
            cfg_d["test_cfg"]["conf"] = 2.0
            model = MODELS.build(cfg_d)
            model.load_state_dict(synth_like(model.state_dict()), strict=False)
            model = model.to(device).eval()
            img = synth_image(1, 1024, seed=77).to(device)
            with torch.no_grad():
                for _ in range(2):   # (the second prediction is the first one with the coarse pass on its side stream: that stream's allocator pool fills here)
                    model.predict(img)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    model.predict(img)
                torch.cuda.synchronize()
            res["eval_dinov2_ms_slide_inference_ms_per_img"] = round(1e3 * (time.perf_counter() - t0) / iters, 2)
            res["eval_iters"] = iters
            del model
            torch.cuda.empty_cache()
        except Exception as e:   # noqa: BLE001
            res["error"] = repr(e)
        finally:
            set_compute_dtype("bf16")
        out[mode] = res
    return out


def host_core_budget():
    """(cores in the affinity mask, cores the cgroup's CPU quota allows - the mask's size where there is no quota)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    share = avail
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()[:2]
        if q != "max":
            share = max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:   # cgroup v1
                q, per = int(f.read()), int(g.read())
            if q > 0:
                share = max(1, int(q / per + 0.5))
        except (OSError, ValueError):
            pass
    return avail, min(share, avail)


def cpu_baseline_guarded(timeout_s=150.0):
    """cpu_baseline() in a child interpreter under a time limit: a baseline that runs away (a thread count the box cannot serve) must not
    take the GPU numbers of the line down with it.  Falls back to 16 threads, then gives up with a stated reason."""
    import subprocess
    tried = []
    for threads in (None, "16"):
        env = dict(os.environ)
        if threads is not None:
            env["VFMSEG_CPU_THREADS"] = threads
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], env=env, capture_output=True, text=True,
                               timeout=timeout_s)
            sys.stderr.write(r.stderr[-2000:])
            for ln in reversed(r.stdout.splitlines()):
                if ln.startswith("{"):
                    out = json.loads(ln)
                    if tried:
                        out["note"] = "; ".join(tried)
                    return out
            tried.append(f"threads={threads or 'all'}: rc {r.returncode}, no result")
        except subprocess.TimeoutExpired:
            tried.append(f"threads={threads or 'all'}: stopped after {timeout_s:.0f} s")
    return {"value": None, "unit": "images/s", "cores": None, "kind": "port", "sample": "failed: " + "; ".join(tried)}


def cpu_baseline(seconds_cap=40.0, model_kw=None):
    """The oracle (CPU fp32 restatement of the reference path) timed on this box's host cores: one train step
    (forward_train + backward + AdamW on the trainable tensors), B=1, 1024^2 input -> 2 x 512^2 passes."""
    _heavy()
    from oracle import torch_ref as R
    from vfmseg_amd import presets
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.synth import synth_image, synth_label, synth_like
    import vfmseg_amd  # noqa: F401
    avail, share = host_core_budget()
    # every host core this process may USE - its affinity mask, cut to the cgroup's CPU quota where there is one (a
    # GPU box grants a share of the host: 256 runnable threads on a 16-core quota take minutes per step) - one warm-up step, one timed
    # step.  VFMSEG_CPU_THREADS is an explicit override only (stated in the output when it is in force).
    override = os.environ.get("VFMSEG_CPU_THREADS")
    cores = max(1, min(avail, int(override))) if override else max(1, min(avail, share))
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle train step on {cores} threads (affinity mask: {avail} cores, cgroup CPU share: {share}"
          + (f", VFMSEG_CPU_THREADS={override}" if override else "") + ") ...", file=sys.stderr, flush=True)
    cfg = presets.dinov2_ms_masked(**(model_kw or {}))  # model_kw: tests shrink the model, not the workload
    bb = cfg["backbone"]["backbone"]
    model = MODELS.build(cfg)
    sd = synth_like(model.state_dict())
    del model
    tk = R.trainable_keys(sd)
    for k in tk:
        sd[k] = sd[k].clone().requires_grad_(True)
    img, lab = synth_image(1, 1024, seed=7), synth_label(1, 1024, seed=7)
    keep = torch.rand(1, 1, 32, 32, generator=torch.Generator().manual_seed(1)) > 0.2
    def one_step():
        losses = R.forward_train(sd, img, lab, (256, 768, 256, 768), keep, depth=bb["depth"], heads=bb["num_heads"],
                                 out_indices=tuple(bb.get("out_indices", (7, 11, 15, 23))))
        grads = torch.autograd.grad(R.total_loss(losses), [sd[k] for k in tk])
        with torch.no_grad():
            for k, g in zip(tk, grads):
                R.adamw_step(sd[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, 1e-4, 0.05)
        return grads
    t0 = time.time()
    grads = one_step()                 # warm-up (thread pool start, allocator growth, first-touch of the weights)
    t_warm = time.time() - t0
    t0 = time.time()
    grads = one_step()
    dt = time.time() - t0
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    out = dict(value=1.0 / dt, unit="images/s", cores=cores, cpu_model=cpu_model, host_cpus=os.cpu_count(), affinity_cores=avail,
               cgroup_cpu_share=share, threads_override=override, kind="port",
               sample=f"1 warm-up ({t_warm:.1f} s) + 1 timed train step (fwd+bwd+AdamW), B=1, 1024^2 -> 2x512^2 passes, fp32 torch CPU, {dt:.1f} s")
    if model_kw is None and os.environ.get("VFMSEG_CPU_EXTRAS", "1") != "0":
        # the other CPU timings BASELINE.md section 3 lists (bounded samples): configs[0] = DINOv2-L + LinearHead, 1x512^2, forward + CE
        # in eval mode; and ONE of the nine 512^2 crops of the SAM-H 1024^2 sliding-window inference (configs[4])
        del grads
        with torch.no_grad():
            sdd = {k: v.detach() for k, v in sd.items()}
            im, lb = synth_image(1, 512, seed=8), synth_label(1, 512, seed=8)
            t0 = time.time()
            lg = R.linear_head_forward(sdd, R.dinov2_forward(sdd, im, depth=bb["depth"], heads=bb["num_heads"]), training=False)
            R.head_loss(lg, lb)
            t_cfg1 = time.time() - t0
            out["cfg1_fwd_ce"] = dict(seconds_per_image=round(t_cfg1, 2), sample="configs[0]: DINOv2-L + LinearHead, 1x512^2, fwd + CE, eval")
            del sdd, sd
            from vfmseg_amd.synth import synth_state_dict
            sam = MODELS.build(presets.sam_linear())
            sds = synth_like(sam.state_dict())
            del sam
            t0 = time.time()
            R.whole_inference(sds, im, (512, 512), backbone="sam")
            t_sam = time.time() - t0
            out["sam_crop_fwd"] = dict(seconds_per_crop=round(t_sam, 2), ms_per_1024_image_est=round(9e3 * t_sam, 0),
                                       sample="configs[4]: SAM-ViT-H + LinearHead, ONE 512^2 crop forward (a 1024^2 slide = 9 crops)")
    return out


def _pin_this_rank():
    """Started by torchrun (the driver's form) rather than by tools/dist_launch.py: this rank pins itself - before torch and the HIP
    runtime start their threads - to the cores the launcher would have given it (tools/dist_launch.py:rank_cpus)."""
    if "VFMSEG_RANK_CPUS" in os.environ or os.environ.get("VFMSEG_PIN_RANKS", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location("dist_launch", os.path.join(ROOT, "tools", "dist_launch.py"))
    dl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dl)
    lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    try:
        cpus = dl.rank_cpus(lw)[lr]
        os.sched_setaffinity(0, cpus)
        os.environ["VFMSEG_RANK_CPUS"] = ",".join(str(c) for c in cpus)
        if os.environ.get("OMP_NUM_THREADS", "1") == "1":    # torchrun's default of 1 starves the (few) CPU-side torch ops
            os.environ["OMP_NUM_THREADS"] = str(max(1, min(8, len(cpus))))
    except (OSError, IndexError):
        pass


def _reserve_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner from rank 0 when its
    first communicator is built), so file descriptor 1 is pointed at stderr for the whole run and the JSON line goes to a private
    duplicate of the original stdout."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def _launcher_rehearsal(a, rank, world, real_stdout):
    """--rehearse-launcher: everything of the N-rank bench EXCEPT the model - process start, rendezvous, the product's bucketed
    gradient all-reduce (parallel.GradSync) on a flat fp32 buffer of the real size (16.6 M values), the barrier-bracketed timing,
    MAX over ranks, rank 0's one JSON line.  Runs on gloo/CPU where there is no GPU (the product path has no CPU fallback, so a
    model step cannot run there); the line it prints is labelled as a rehearsal and is never the headline metric."""
    _heavy()
    import torch.distributed as dist
    from vfmseg_amd import parallel
    n = 16_600_000 // 64
    g = torch.full((n,), float(rank + 1))
    names = ["aux_decoder.w", "decode_head.w"] + [f"backbone.model.base_model.model.blocks.{i}.attn.qkv.lora_A.default.weight" for i in (1, 0)]
    offs = [0, n // 4, n // 2, 3 * n // 4, n]
    gs = parallel.GradSync(g, parallel.make_buckets(names, offs), None)

    def step():
        g.fill_(float(rank + 1))
        gs.ready(0)
        gs.finish()
    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ok = bool(torch.allclose(g, torch.full((n,), (world + 1) / 2.0)))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        real_stdout.write(json.dumps({
            "metric": "launcher rehearsal: bucketed gradient all-reduce steps/s (no model; NOT the headline metric)",
            "value": round(a.steps / dt, 3), "unit": "steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "allreduce_mean_ok": ok,
            "config": {"workload": "rehearsal of bench.py's N-rank plumbing on %s" % dist.get_backend() if world > 1 else "single rank",
                       "parallelism": "dp%d" % world}}) + "\n")
        real_stdout.flush()
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0 if ok else 1


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2, help="samples per GPU (reference: 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-eval", action="store_true", help="skip the eval ms/img @1024x1024 leg (N=1 only)")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the f32 parity-mode speed leg (N=1 only)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "bf16x3", "fp16"])
    ap.add_argument("--workload", default="ms1024", choices=["ms1024", "single512"],
                    help="ms1024: BASELINE configs[1] (1024^2 sample -> LR + HR 512^2 passes, both heads; the headline metric); "
                         "single512: one 512^2 pass per sample, DINOv2-L + LoRA + LinearHead (SURVEY 8: the labelled single-pass step)")
    ap.add_argument("--tune", action="append", default=[], help="kernel tuning knob KEY=INT (vfm_tune), repeatable")
    ap.add_argument("--depth", type=int, default=None, help="backbone depth for launcher / DP rehearsals (default: the config's 24; "
                                                            "any other value marks the line as not the headline metric)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="internal: print the cpu_baseline object as one JSON line (no GPU work)")
    ap.add_argument("--rehearse-launcher", action="store_true", help="N-rank plumbing without the model (runs on CPU/gloo): see _launcher_rehearsal")
    ap.add_argument("--launch-timeout", type=float, default=None, help="launcher parent: stop all ranks after this many seconds")
    return ap.parse_args(argv)


def main():
    a = parse_args()
    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (the reference's tools/dist_train.sh:9-17 role).  It starts N
        # fresh interpreters of this file with the torchrun env, relays rank 0's JSON line and returns the worst exit code.  It has not
        # imported torch and never touches the GPU; under `python -m torch.distributed.run ... bench.py` WORLD_SIZE is set and this branch
        # is not taken.
        import importlib.util
        spec = importlib.util.spec_from_file_location("dist_launch", os.path.join(ROOT, "tools", "dist_launch.py"))
        dl = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(dl)
        sys.exit(dl.launch([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], a.gpus, timeout=a.launch_timeout))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        _pin_this_rank()
        os.environ.setdefault("NCCL_DEBUG", "VERSION")   # RCCL's version banner (rank 0; lands on stderr: fd 1 is re-pointed below)
    real_stdout = _reserve_stdout()
    _heavy()

    from vfmseg_amd import parallel
    from vfmseg_amd.precision import set_compute_dtype
    if a.rehearse_launcher:
        rank, world, local = parallel.init_from_env("gloo")
        sys.exit(_launcher_rehearsal(a, rank, world, real_stdout))
    rank, world, local = parallel.init_from_env("nccl")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: start with plain `python bench.py --gpus N` (self-launching) or "
                         f"`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    ndev = torch.cuda.device_count()
    dev_index = local if local < ndev else local % max(ndev, 1)   # rehearsal: several gloo ranks may share one GPU
    torch.cuda.set_device(dev_index)
    from vfmseg_amd import lib as _L
    _L.set_device_index(dev_index)
    device = torch.device("cuda", dev_index)
    set_compute_dtype(a.dtype)
    np.random.seed(rank)
    for kv in a.tune:
        from vfmseg_amd import ops as _ops
        k, v = kv.split("=")
        _ops.tune(k, int(v))

    model, ow = build(device, a.batch, workload=a.workload, depth=a.depth, amp=(a.dtype == "fp16"))
    parallel.attach(model, ow)
    from vfmseg_amd import functional as Fh
    Fh.manual_seed(1234 + rank)
    data = make_batch(a.batch, rank, 0, device, size=1024 if a.workload == "ms1024" else 512)
    timer = KernelTimer()
    if not a.no_roofline:
        timer.install()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    print(f"[bench] rank {rank}: model built, warming up", file=sys.stderr, flush=True)
    for _ in range(a.warmup):
        model.train_step(data, ow)
    barrier()
    print(f"[bench] rank {rank}: timing {a.steps} steps", file=sys.stderr, flush=True)
    timer.on = not a.no_roofline
    gs = getattr(ow, "grad_sync", None)
    if gs is not None and world > 1:
        gs.measure = True          # event pair around the join of the all-reduce stream: what backward did NOT hide
    host_s = 0.0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        h0 = time.perf_counter()
        model.train_step(data, ow)
        host_s += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0      # this rank's own clock, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    timer.on = False
    ranks = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        # per-rank diagnostics for rank 0's line: a slow or host-bound rank, an exposed all-reduce and a bad core assignment show here
        exposed = gs.exposed_ms() if gs is not None and hasattr(gs, "exposed_ms") else float("nan")
        try:
            ncpu = len(os.sched_getaffinity(0))
        except Exception:
            ncpu = -1
        mine = torch.tensor([1e3 * dt_own / a.steps, 1e3 * host_s / a.steps, exposed / max(a.steps, 1), float(ncpu)], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        cols = list(zip(*[x.tolist() for x in allr]))
        ranks = {"ms_per_step": [round(v, 3) for v in cols[0]], "ms_per_step_min": round(min(cols[0]), 3), "ms_per_step_max": round(max(cols[0]), 3),
                 "host_enqueue_ms_per_step": [round(v, 3) for v in cols[1]],
                 "exposed_allreduce_ms_per_step": [round(v, 4) for v in cols[2]],
                 "affinity_cores": [int(v) for v in cols[3]],
                 "what": "per rank, own clock: step time before the closing barrier; host time spent inside train_step (enqueue); time the "
                         "compute stream waited for the gradient all-reduce stream after backward (HIP events around GradSync.finish's join)"}
    ms_per_step = dt * 1000.0 / a.steps
    value = a.batch * world * a.steps / dt

    out = {
        "metric": "train images/sec @512x512 DINOv2-L+LoRA (dg_lora_dinov2_ms_masked, 1024^2 sample = LR+HR 512^2 passes)",
        "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "configs[1]: dg_lora_dinov2_ms_masked.py, synthetic 1024^2 19-class samples -> 2x512^2 passes, "
                               "bs=%d/GPU, full train step (fwd+bwd+allreduce+AdamW), LoRA dropout on" % a.batch,
                   "global_batch": a.batch * world, "parallelism": "dp%d" % world},
    }
    if ranks is not None:
        out["ranks"] = ranks
        try:
            out["comm"] = {"backend": torch.distributed.get_backend(), "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()),
                           "grad_bytes_per_step": int(ow.optimizer.gflat.numel() * 4), "buckets": len(gs.buckets) if gs is not None else 0,
                           "NCCL_DEBUG": os.environ.get("NCCL_DEBUG")}
        except Exception as e:   # noqa: BLE001
            out["comm"] = {"error": repr(e)}
    if a.depth is not None and a.depth != 24:
        out["metric"] = "REHEARSAL at backbone depth %d (not the headline metric): " % a.depth + out["metric"]
        out["config"]["depth"] = a.depth
    if a.workload == "single512":
        out["metric"] = "train images/sec @512x512 DINOv2-L+LoRA + LinearHead, single 512^2 pass per sample (not the headline metric)"
        out["config"]["workload"] = "single-pass variant of configs[1]: EncoderDecoder(LoRABackbone(DINOv2-L), LinearHead), 512^2 inputs, bs=%d/GPU" % a.batch
    if rank == 0:
        GEMMS = ("gemm", "gemm_tn")
        allg = timer.summary(kinds=GEMMS)
        # the dominant kernel family: k_gemm_w4 (LDS-DMA chunk ring; 128x128 and 256x256 tile forms) = the backbone's QKV / proj / fc1 / fc2
        # forward and input-gradient launches (>= 8.6 GFLOP each; 193 launches and ~47 % of the step's GPU time in profiles/*_kernel_stats.csv)
        g = timer.summary(kinds=("gemm",), region="backbone", min_flops=8e9) or allg
        if g is not None:
            traffic, traffic_src = None, None
            for tj_name in ("r04_pmc_gemm_traffic.json", "r03_pmc_gemm_traffic.json", "r02_pmc_gemm_traffic.json", "r01_pmc_gemm_traffic.json"):
                tj = os.path.join(ROOT, "profiles", tj_name)
                if os.path.exists(tj) and a.workload == "ms1024":  # PMC counters cannot be read from inside the timed run: committed rocprofv3 --pmc passes
                    with open(tj) as f:
                        tr = json.load(f)
                    traffic = tr.get("mean_bytes_per_launch")
                    traffic_src = f"profiles/{tj_name}: mean FETCH_SIZE*2 + WRITE_SIZE bytes per launch over the four backbone GEMM shapes"
                    break
            est = lambda r: round(r["ms"] * timer.every / a.steps, 3)   # noqa: E731
            out["roofline"] = {"bound": "mfma", "achieved": round(g["tflops"], 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(g["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                               "kernel": "the backbone's QKV / proj / fc1 / fc2 forward + input-gradient GEMMs (193 launches per step, ~47 % of its GPU time): "
                                         "k_gemm_w4<true,2,1,2,4,2,0> (128x128 tiles, LDS-DMA chunk ring; the N = 1024 shapes and the qkv input gradient) and "
                                         "k_gemm_w4<true,4,2,2,4,2,0> (256x256 tiles, 8 waves; fc1 forward, fc2 input gradient, qkv projection) "
                                         "(algorithmic FLOPs of the sampled launches / their HIP-event time)",
                               "launches_timed": g["launches"], "sampled_every": timer.every,
                               "avg_launch_us": round(g["ms"] * 1e3 / g["launches"], 2),
                               "avg_launch_gflop": round(g["flops"] / g["launches"] / 1e9, 2), "ms_per_step_est": est(g),
                               "all_gemms": None if allg is None else {
                                   "what": "every bf16 GEMM launch of the timed steps (backbone + heads + weight gradients)",
                                   "tflops": round(allg["tflops"], 2), "frac": round(allg["tflops"] / PEAK_BF16_TFLOPS, 4),
                                   "launches_timed": allg["launches"], "ms_per_step_est": est(allg)}}
            # the block-level figure north_star's 40 % target is stated on: the ViT-L attention + MLP kernels of the backbone
            # (QKV / proj / fc1 / fc2 GEMMs forward + dgrad, LoRA GEMMs, flash attention forward + backward) as ONE family:
            # algorithmic FLOPs (attention backward counted as its five products) / summed kernel time of the sampled launches
            bg = timer.summary(kinds=GEMMS, region="backbone")
            af = timer.summary(kinds=("attn_fwd",), region="backbone")
            ab = timer.summary(kinds=("attn_bwd",), region="backbone")
            blk = timer.summary(kinds=GEMMS + ("attn_fwd", "attn_bwd"), region="backbone")
            if blk is not None:
                out["roofline"]["blocks_frac"] = round(blk["tflops"] / PEAK_BF16_TFLOPS, 4)
                out["roofline"]["blocks"] = {
                    "what": "backbone attention+MLP kernels (bf16 GEMMs + flash attention fwd/bwd), HIP events on sampled launches",
                    "achieved": round(blk["tflops"], 2), "ms_per_step_est": est(blk),
                    "gemm": None if bg is None else {"tflops": round(bg["tflops"], 1), "ms_per_step_est": est(bg)},
                    "attn_fwd": None if af is None else {"tflops": round(af["tflops"], 1), "ms_per_step_est": est(af)},
                    "attn_bwd": None if ab is None else {"tflops": round(ab["tflops"], 1), "ms_per_step_est": est(ab)}}
        # end-to-end model-FLOP utilisation (722.4 GFLOP fwd per image-pass, train ~2.2x, 2 passes/sample)
        out["model_tflops"] = round(value / world * (2 if a.workload == "ms1024" else 1) * 722.4e9 * 2.2 / 1e12, 2)
        out["model_frac"] = round(out["model_tflops"] / PEAK_BF16_TFLOPS, 4)
        if world == 1 and not a.no_eval and a.workload == "ms1024":
            del model, ow, data
            torch.cuda.empty_cache()
            try:
                out["eval"] = eval_leg(device, timer if not a.no_roofline else None)
            except Exception as e:
                out["eval"] = {"error": repr(e)}
        if world == 1 and not a.no_eval and a.workload == "ms1024" and a.dtype == "bf16" and a.depth is None:
            try:
                out["other_backbones"] = other_backbones_leg(device, a.batch, timer if not a.no_roofline else None)
            except Exception as e:
                out["other_backbones"] = {"error": repr(e)}
        if world == 1 and not a.no_parity_mode and a.workload == "ms1024" and a.dtype == "bf16" and a.depth is None:
            try:
                out["parity_mode"] = parity_leg(device, a.batch)
            except Exception as e:
                out["parity_mode"] = {"error": repr(e)}
            try:   # the reference's `--amp` step: fp16 autocast (the fp16 twin library) + dynamic loss scale, same workloads
                out["amp_fp16"] = dict(parity_leg(device, a.batch, steps=10, iters=5, modes=("fp16",))["fp16"],
                                       what="tools/train.py --amp: fp16 storage + v_mfma_f32_32x32x16_f16 (libvfmseg_hip_f16.so), AmpOptimWrapper")
            except Exception as e:
                out["amp_fp16"] = {"error": repr(e)}
        if world == 1 and not a.no_cpu_baseline and a.workload == "ms1024":
            try:
                out["cpu_baseline"] = cpu_baseline_guarded()
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
