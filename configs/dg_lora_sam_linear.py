# SAM-ViT-H + LoRA(qkv) + LinearHead, sliding-window test (reference: configs/_base_/models/lora_sam_linear.py; BASELINE config 5).
from vfmseg_amd import presets

crop_size = (512, 512)
num_classes = 19
model = presets.sam_linear()
_o = presets.optim_cfg()
optim_wrapper = _o["optim_wrapper"]
param_scheduler = _o["param_scheduler"]
randomness = dict(seed=0)
env_cfg = dict(dist_cfg=dict(backend="nccl"))
