"""Time of the backbone alone (fwd + bwd, 4 images = LR+HR passes of bs 2) vs the full train step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vfmseg_amd  # noqa
from vfmseg_amd import presets
from vfmseg_amd.registry import MODELS
from vfmseg_amd.precision import set_compute_dtype
set_compute_dtype("bf16")
cfg = dict(type="LoRABackbone", backbone=presets.dinov2_backbone(), Lora_config=presets.lora_cfg())
m = MODELS.build(cfg).cuda().train()
img = torch.randn(4, 3, 512, 512, device="cuda")
def step():
    for p in m.parameters(): p.grad = None
    xcat, _ = m.forward_tokens([(img, None)], seed=1)
    xcat.backward(torch.ones_like(xcat))
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.time()
n = 10
for _ in range(n): step()
torch.cuda.synchronize()
print(f"backbone fwd+bwd: {(time.time()-t0)/n*1e3:.2f} ms")
