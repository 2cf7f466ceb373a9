"""Forward repeatability: same model, same batch, losses of repeated forwards with / without a backward in between."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vfmseg_amd  # noqa
from tests.helpers import full_state_dict
from vfmseg_amd import presets, functional as Fh
from vfmseg_amd.optim import PEFTOptimWrapperConstructor
from vfmseg_amd.precision import set_compute_dtype
from vfmseg_amd.registry import MODELS
from vfmseg_amd.segmentors import SegDataSample
from vfmseg_amd.synth import synth_image, synth_label
set_compute_dtype(sys.argv[1])
depth = 2
cfg = presets.dinov2_ms_masked(depth=depth)
cfg["backbone"]["backbone"]["out_indices"] = [0, 0, 1, 1]
model = MODELS.build(cfg)
model.load_state_dict(full_state_dict(depth=depth))
model = model.cuda().train()
for m in model.modules():
    if hasattr(m, "dropout_ratio"):
        m.dropout_ratio = 0.0
    if hasattr(m, "p") and isinstance(getattr(m, "p"), float):
        m.p = 0.0
oc = presets.optim_cfg()
ow = PEFTOptimWrapperConstructor(oc["optim_wrapper"])(model, oc["param_scheduler"])
opt = ow.optimizer
keep = torch.rand(2, 1, 32, 32, generator=torch.Generator().manual_seed(12)) > 0.2
imgs = synth_image(2, 1024, seed=500).cuda()
labs = synth_label(2, 1024, seed=500)
data = dict(inputs=imgs, data_samples=[SegDataSample(gt_sem_seg=labs[k]) for k in range(2)])
def fwd():
    Fh.manual_seed(4321)
    model.fixed_crop_box = (256, 768, 128, 640)
    model.aux_decoder.transformer_decoder.fixed_keep = keep
    d = model.data_preprocessor(data, True) if getattr(model, "data_preprocessor", None) is not None else data
    losses = model.loss(d["inputs"], d["data_samples"])
    print("   " + ", ".join("%s=%.7f" % (k, float(v)) for k, v in losses.items()), "| local_iter", getattr(model, "local_iter", None))
    return sum(v for k, v in losses.items() if "loss" in k)
xs = torch.nn.functional.interpolate(imgs, size=(512, 512), mode="bilinear", align_corners=False)
def probe():
    with torch.no_grad():
        feats = model.backbone(xs)
        print("   probe backbone taps: " + ", ".join("%.9e" % f.float().abs().sum().item() for f in feats))
        lg = model.decode_head(feats) if callable(getattr(model.decode_head, "forward", None)) else None
        if lg is not None:
            print("   probe decode_head logits: %.9e" % lg.float().abs().sum().item())
for step in sys.argv[2].split(","):
    if step == "p":
        probe()
        continue
    if step == "f":
        with torch.no_grad():
            fwd()
    else:
        opt.zero_grad()
        (fwd() * float(step)).backward()
        torch.cuda.synchronize()
        print("   (backward at scale %s)" % step)
