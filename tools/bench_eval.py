#!/usr/bin/env python
"""Eval latency of the two inference paths BASELINE.json names: ms/img at 1024x1024, bf16, batch 1, synthetic.
  dinov2 : MsVFMEncoderDecoder 'ms_slide_inference' (coarse 512x1024 pass + confidence-gated 512^2 refinement, 3x3 windows)
  sam    : EncoderDecoder + LoRA SAM-ViT-H + LinearHead, 'slide' (3x3 windows of 512^2)  [BASELINE config 5]
Prints one JSON line per model."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vfmseg_amd  # noqa: F401
from vfmseg_amd import presets
from vfmseg_amd.registry import MODELS
from vfmseg_amd.synth import synth_image, synth_like


def run(name, cfg, iters=5):
    model = MODELS.build(cfg)
    model.load_state_dict(synth_like(model.state_dict()), strict=False)
    model = model.cuda().eval()
    img = synth_image(1, 1024, seed=77).cuda()
    with torch.no_grad():
        for _ in range(2):
            model.predict(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            model.predict(img)
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    extra = {}
    if hasattr(model, "last_refined"):
        extra["refined_windows"] = len(model.last_refined)
    print(json.dumps(dict(metric="eval ms/img @1024x1024", model=name, value=round(ms, 2), unit="ms/img", higher_is_better=False,
                          dtype="bf16", data="synthetic", **extra)), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["dinov2", "sam"]
    if "dinov2" in which:
        cfg = presets.dinov2_ms_masked()
        cfg["test_cfg"]["conf"] = 2.0  # random-init logits are never "confident": force all 9 refinements (worst case)
        run("DINOv2-L+LoRA ms_slide_inference (9/9 windows refined)", cfg)
    if "sam" in which:
        run("SAM-ViT-H+LoRA + LinearHead, slide 3x3", presets.sam_linear())
