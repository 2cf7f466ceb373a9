"""One DINOv2-L ms_slide_inference prediction (all nine windows refined) in a given precision mode, for profiling:
    python tools/eval_once.py MODE [ITERS]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import vfmseg_amd  # noqa: F401
from vfmseg_amd import presets
from vfmseg_amd.precision import set_compute_dtype
from vfmseg_amd.registry import MODELS
from vfmseg_amd.synth import synth_image, synth_like

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
set_compute_dtype(mode)
cfg = presets.dinov2_ms_masked()
cfg["test_cfg"]["conf"] = 2.0
model = MODELS.build(cfg)
model.load_state_dict(synth_like(model.state_dict()), strict=False)
model = model.cuda().eval()
img = synth_image(1, 1024, seed=77).cuda()
with torch.no_grad():
    model.predict(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        model.predict(img)
    torch.cuda.synchronize()
print(f"{mode}: {1e3 * (time.perf_counter() - t0) / iters:.2f} ms/img")
