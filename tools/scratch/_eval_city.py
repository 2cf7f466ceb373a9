"""Cityscapes-sized (1024 x 2048) inference smoke + latency for both eval paths (18 crops at 1024 x 2048)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vfmseg_amd  # noqa: F401
from vfmseg_amd import presets
from vfmseg_amd.registry import MODELS
from vfmseg_amd.synth import synth_image, synth_like
for name, cfg in (("dinov2 ms_slide_inference", presets.dinov2_ms_masked()), ("sam slide", presets.sam_linear())):
    if "ms_slide" in name:
        cfg["test_cfg"]["conf"] = 2.0
    model = MODELS.build(cfg)
    model.load_state_dict(synth_like(model.state_dict()), strict=False)
    model = model.cuda().eval()
    img = torch.cat([synth_image(1, 1024, seed=5), synth_image(1, 1024, seed=6)], dim=3).cuda()   # [1, 3, 1024, 2048]
    with torch.no_grad():
        out = model.predict(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = model.predict(img)
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 3
    seg = out[0].pred_sem_seg.data if hasattr(out[0], "pred_sem_seg") else out[0]
    print(json.dumps(dict(model=name, input="1024x2048", ms_per_img=round(ms, 2), out_shape=list(getattr(seg, "shape", [])))), flush=True)
    del model
    torch.cuda.empty_cache()
