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

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense")
PEAK_HBM_GBS = 8000.0


def build(device, batch, seed=0, workload="ms1024"):
    import vfmseg_amd  # noqa: F401  (registers the model classes)
    from vfmseg_amd import presets
    from vfmseg_amd.optim import PEFTOptimWrapperConstructor
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.synth import synth_like

    torch.manual_seed(seed)
    model = MODELS.build(presets.dinov2_ms_masked() if workload == "ms1024" else presets.dinov2_linear())
    # random-init weights of the named architecture (no checkpoints offline): key-hashed synthetic values
    sd = synth_like(model.state_dict())
    model.load_state_dict(sd)
    model = model.to(device)
    model.train()
    oc = presets.optim_cfg()
    ow = PEFTOptimWrapperConstructor(oc["optim_wrapper"])(model, oc["param_scheduler"])
    return model, ow


def make_batch(batch, rank, step, device, size=1024):
    from vfmseg_amd.segmentors import SegDataSample
    from vfmseg_amd.synth import synth_image, synth_label
    img = synth_image(batch, size, seed=100 + rank).to(device)
    lab = synth_label(batch, size, seed=100 + rank).to(device)
    return dict(inputs=img, data_samples=[SegDataSample(gt_sem_seg=lab[i]) for i in range(batch)])


class GemmTimer:
    """HIP-event timing of the dominant kernel (the bf16 MFMA GEMM) on the stream it is launched on."""

    def __init__(self):
        self.recs = []
        self.on = False
        self.calls = 0
        self.every = 7   # prime: cycles through every GEMM site of a layer over the steps

    def install(self):
        from vfmseg_amd import ops
        orig = ops.gemm
        timer = self

        def timed(a, b, c, **kw):
            if not timer.on or a.dtype != torch.bfloat16:
                return orig(a, b, c, **kw)
            timer.calls += 1
            if timer.calls % timer.every:  # sample every Nth launch: event pairs cost host time
                return orig(a, b, c, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a2 = a[0] if a.dim() == 3 else a
            b2 = b[0] if b.dim() == 3 else b
            m, k = (a2.shape[1], a2.shape[0]) if kw.get("trans_a") else a2.shape
            n = b2.shape[1] if kw.get("trans_b") else b2.shape[0]
            batch = a.shape[0] if a.dim() == 3 else 1
            e0.record()
            r = orig(a, b, c, **kw)
            e1.record()
            timer.recs.append((2.0 * m * n * k * batch, e0, e1))
            return r

        ops.gemm = timed
        import vfmseg_amd.backbones as bb
        import vfmseg_amd.functional as fn
        bb.ops.gemm = timed
        fn.ops.gemm = timed

    def summary(self):
        if not self.recs:
            return None
        fl = sum(r[0] for r in self.recs)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.recs)
        return dict(flops=fl, ms=ms, launches=len(self.recs), sampled_every=self.every)


def cpu_baseline(seconds_cap=40.0, model_kw=None):
    """The oracle (CPU fp32 restatement of the reference path) timed on this box's host cores: one train step
    (forward_train + backward + AdamW on the trainable tensors), B=1, 1024^2 input -> 2 x 512^2 passes."""
    from oracle import torch_ref as R
    from vfmseg_amd import presets
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.synth import synth_image, synth_label, synth_like
    import vfmseg_amd  # noqa: F401
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("VFMSEG_CPU_THREADS", "16"))))  # a GPU box grants a 16-core share
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle train step on {cores} threads ...", file=sys.stderr, flush=True)
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
    t0 = time.time()
    losses = R.forward_train(sd, img, lab, (256, 768, 256, 768), keep, depth=bb["depth"], heads=bb["num_heads"],
                             out_indices=tuple(bb.get("out_indices", (7, 11, 15, 23))))
    grads = torch.autograd.grad(R.total_loss(losses), [sd[k] for k in tk])
    with torch.no_grad():
        for k, g in zip(tk, grads):
            R.adamw_step(sd[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, 1e-4, 0.05)
    dt = time.time() - t0
    return dict(value=1.0 / dt, unit="images/s", cores=cores, kind="port",
                sample=f"1 train step (fwd+bwd+AdamW), B=1, 1024^2 -> 2x512^2 passes, fp32 torch CPU, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=2, help="samples per GPU (reference: 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--workload", default="ms1024", choices=["ms1024", "single512"],
                    help="ms1024: BASELINE configs[1] (1024^2 sample -> LR + HR 512^2 passes, both heads; the headline metric); "
                         "single512: one 512^2 pass per sample, DINOv2-L + LoRA + LinearHead (SURVEY 8: the labelled single-pass step)")
    ap.add_argument("--tune", action="append", default=[], help="kernel tuning knob KEY=INT (vfm_tune), repeatable")
    a = ap.parse_args()

    from vfmseg_amd import parallel
    from vfmseg_amd.precision import set_compute_dtype
    rank, world, local = parallel.init_from_env("nccl")
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
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

    model, ow = build(device, a.batch, workload=a.workload)
    parallel.attach(model, ow)
    from vfmseg_amd import functional as Fh
    Fh.manual_seed(1234 + rank)
    data = make_batch(a.batch, rank, 0, device, size=1024 if a.workload == "ms1024" else 512)
    timer = GemmTimer()
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
    t0 = time.perf_counter()
    for _ in range(a.steps):
        model.train_step(data, ow)
    barrier()
    dt = time.perf_counter() - t0
    timer.on = False
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
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
    if a.workload == "single512":
        out["metric"] = "train images/sec @512x512 DINOv2-L+LoRA + LinearHead, single 512^2 pass per sample (not the headline metric)"
        out["config"]["workload"] = "single-pass variant of configs[1]: EncoderDecoder(LoRABackbone(DINOv2-L), LinearHead), 512^2 inputs, bs=%d/GPU" % a.batch
    if rank == 0:
        s = timer.summary()
        if s is not None:
            ach = s["flops"] / (s["ms"] * 1e-3) / 1e12
            traffic, traffic_src = None, None
            tj = os.path.join(ROOT, "profiles", "r01_pmc_gemm_traffic.json")
            if os.path.exists(tj) and a.workload == "ms1024":  # PMC counters cannot be read from inside the timed run: committed rocprofv3 --pmc passes
                with open(tj) as f:
                    tr = json.load(f)
                traffic = tr.get("mean_bytes_per_launch")
                traffic_src = "profiles/r01_pmc_gemm_traffic.json: mean FETCH_SIZE*2 + WRITE_SIZE bytes per launch over the four backbone GEMM shapes"
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                               "kernel": "bf16 MFMA GEMM (k_gemm_w4 128x128 ring tiles + k_gemm_bf16: every GEMM launch of the timed steps)",
                               "launches_timed": s["launches"], "sampled_every": s["sampled_every"],
                               "gemm_ms_per_step_est": round(s["ms"] * s["sampled_every"] / a.steps, 3)}
        # end-to-end model-FLOP utilisation (722.4 GFLOP fwd per image-pass, train ~2.2x, 2 passes/sample)
        out["model_tflops"] = round(value / world * (2 if a.workload == "ms1024" else 1) * 722.4e9 * 2.2 / 1e12, 2)
        if world == 1 and not a.no_cpu_baseline and a.workload == "ms1024":
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
