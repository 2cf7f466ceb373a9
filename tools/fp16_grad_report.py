"""Per-parameter gradient error of the fp16 (`--amp`) HIP path against the fp32 oracle, next to the same error of the oracle run under the
emulated CUDA autocast policy (oracle/amp_emul.py) - i.e. "is the HIP fp16 step as close to exact as the reference's own fp16 step".
    python tools/fp16_grad_report.py [MODE=fp16] [LOSS_SCALE=65536] [DEPTH=4] [BATCH=1] [golden]
`golden`: the inputs of tests/golden/train_step.npz (depth 24, batch 2).  The depth-24 oracle passes take ~15 min of host time.
Output of the round-3 runs: profiles/r03_fp16_grad_report_*.log."""
import sys, contextlib, torch
sys.path.insert(0, "/root/repo")
import vfmseg_amd  # noqa
from oracle import torch_ref as R
from oracle.amp_emul import cuda_autocast
from tests.helpers import full_state_dict
from vfmseg_amd import presets
from vfmseg_amd.precision import set_compute_dtype
from vfmseg_amd.registry import MODELS
from vfmseg_amd.segmentors import SegDataSample
from vfmseg_amd.synth import synth_image, synth_label
MODE = sys.argv[1] if len(sys.argv) > 1 else "fp16"
SCALE = float(sys.argv[2]) if len(sys.argv) > 2 else 65536.0
DEPTH = int(sys.argv[3]) if len(sys.argv) > 3 else 4
NB = int(sys.argv[4]) if len(sys.argv) > 4 else 1
depth, out_idx = DEPTH, ((0, 1, 2, 3) if DEPTH == 4 else (7, 11, 15, 23))
sd0 = full_state_dict(depth=depth)
keep = torch.rand(NB, 1, 32, 32, generator=torch.Generator().manual_seed(3)) > 0.2
img, lab = synth_image(NB, 1024, seed=60), synth_label(NB, 1024, seed=60)
box = (256, 768, 128, 640)
if len(sys.argv) > 5 and sys.argv[5] == "golden":
    import numpy as np
    from tests.helpers import cached_full_state_dict
    G = np.load("tests/golden/train_step.npz")
    sd0 = cached_full_state_dict()
    img, lab = synth_image(2, 1024, seed=3), synth_label(2, 1024, seed=3)
    box = tuple(int(v) for v in G["hr_crop_box"]); keep = torch.from_numpy(G["mask_rand"]) > 0.2
def ograds(emul, scale):
    tk = R.trainable_keys(sd0)
    work = dict(sd0)
    for k in tk: work[k] = sd0[k].detach().clone().requires_grad_(True)
    with (cuda_autocast() if emul else contextlib.nullcontext()):
        losses = R.forward_train(work, img, lab, box, keep, depth=depth, out_indices=out_idx)
        g = torch.autograd.grad(R.total_loss(losses) * scale, [work[k] for k in tk], allow_unused=True)
    return {k: (None if x is None else x.float() / scale) for k, x in zip(tk, g)}
g32, g16 = ograds(False, 1.0), ograds(True, SCALE)
set_compute_dtype(MODE)
cfg = presets.dinov2_ms_masked(depth=depth)
cfg["backbone"]["backbone"]["out_indices"] = list(out_idx)
model = MODELS.build(cfg); model.load_state_dict(sd0); model = model.cuda().train()
for m in model.modules():
    if hasattr(m, "dropout_ratio"): m.dropout_ratio = 0.0
    if hasattr(m, "p") and isinstance(getattr(m, "p"), float): m.p = 0.0
model.aux_decoder.transformer_decoder.fixed_keep = keep
model.fixed_crop_box = box
losses = model.loss(img.cuda(), [SegDataSample(gt_sem_seg=lab[i]) for i in range(NB)])
total, _ = model.parse_losses(losses)
(total * SCALE).backward()
named = dict(model.named_parameters())
rows = []
for k, g in g32.items():
    if g is None or named[k].grad is None: continue
    gh = named[k].grad.float().cpu() / SCALE
    eh = ((gh - g).norm() / g.norm()).item(); ee = ((g16[k] - g).norm() / g.norm()).item()
    ph = ((gh.double().flatten() @ g.double().flatten()) / g.double().pow(2).sum() - 1).item()
    pe = ((g16[k].double().flatten() @ g.double().flatten()) / g.double().pow(2).sum() - 1).item()
    rows.append((eh / max(ee, 1e-12), eh, ee, k + f"  proj-1: HIP {ph:+.1e} emu {pe:+.1e}", g.norm().item()))
import collections
grp = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
for k, g in g32.items():
    if g is None or named[k].grad is None: continue
    j = "lora" if "lora_" in k else ("decode_head" if k.startswith("decode_head") else "aux")
    grp[j][0] += g.double().pow(2).sum().item(); grp[j][1] += (named[k].grad.double().cpu() / SCALE).pow(2).sum().item(); grp[j][2] += g16[k].double().pow(2).sum().item()
for j, (a, b, c) in grp.items():
    print(f"group {j}: norm rel err HIP {abs((b / a) ** 0.5 - 1):.2e}  emulated {abs((c / a) ** 0.5 - 1):.2e}")
rows.sort(reverse=True)
print(f"mode {MODE} scale {SCALE}: per-parameter gradient error (L2 relative) HIP vs fp32 | emulated fp16 vs fp32 | ratio")
for r, eh, ee, k, n in rows:
    print(f"{eh:9.2e} {ee:9.2e} x{r:6.1f}  |g|={n:.2e}  {k}")
