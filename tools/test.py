#!/usr/bin/env python
"""Evaluation entry point (tools/test.py:96-145): CONFIG CHECKPOINT [--backbone PTH --work-dir --cfg-options].
Runs the configured test mode (ms_slide_inference / slide) over synthetic images and reports mIoU against the
labels (mmseg IoUMetric semantics) and ms/img."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("checkpoint", nargs="?")
    ap.add_argument("--backbone", help="backbone .pth merged under 'backbone.' (LoadBackboneHook semantics)")
    ap.add_argument("--work-dir")
    ap.add_argument("--cfg-options", nargs="+")
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--size", type=int, nargs=2, default=[1024, 1024])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    a = ap.parse_args()
    import torch
    import vfmseg_amd  # noqa: F401
    from vfmseg_amd.config import Config, parse_cfg_options
    from vfmseg_amd.metrics import IoUMetric
    from vfmseg_amd.precision import set_compute_dtype
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.synth import synth_image, synth_label, synth_like
    set_compute_dtype(a.dtype)
    cfg = Config.fromfile(a.config)
    cfg.merge_from_dict(parse_cfg_options(a.cfg_options))
    model = MODELS.build(cfg["model"])
    sd = synth_like(model.state_dict())
    if a.checkpoint:
        ck = torch.load(a.checkpoint, map_location="cpu")
        sd.update(ck.get("state_dict", ck))
    if a.backbone:
        sd.update({"backbone." + k: v for k, v in torch.load(a.backbone, map_location="cpu").items()})
    model.load_state_dict(sd, strict=False)
    model = model.cuda().eval()
    metric = IoUMetric(num_classes=model.num_classes)
    t = 0.0
    for i in range(a.images):
        img, lab = synth_image(1, tuple(a.size), seed=500 + i).cuda(), synth_label(1, tuple(a.size), seed=500 + i).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.predict(img)
        torch.cuda.synchronize()
        t += time.perf_counter() - t0
        metric.process(out[0].pred_sem_seg.data[0], lab[0, 0])
    res = metric.compute()
    res["ms_per_img"] = 1e3 * t / a.images
    print(res)


if __name__ == "__main__":
    main()
