# Hot-path training config in the reference's schema (same keys as configs/dg/gta2citys/dg_lora_sam_ms_masked.py
# + its _base_ model file); a user's own reference configs load through the same Config.fromfile.
from vfmseg_amd import presets

crop_size = (1024, 1024)
num_classes = 19
model = presets.sam_ms_masked()
_o = presets.optim_cfg()
optim_wrapper = _o["optim_wrapper"]
param_scheduler = _o["param_scheduler"]
train_dataloader = dict(batch_size=2, num_workers=4, sampler=dict(type="InfiniteSampler", shuffle=True))
train_cfg = dict(type="IterBasedTrainLoop", max_iters=40000, val_interval=8000)
default_hooks = dict(logger=dict(type="LoggerHook", interval=50), checkpoint=dict(type="CheckpointHook", by_epoch=False, interval=4000, max_keep_ckpts=3))
randomness = dict(seed=0)
log_config = dict(interval=50, img_interval=500)
env_cfg = dict(dist_cfg=dict(backend="nccl"))
