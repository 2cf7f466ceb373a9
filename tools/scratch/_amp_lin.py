"""Is backward linear in the loss scale?  One forward/backward at scale 1 and at 2**16 on the same model and batch; per-parameter
relative difference of grad / scale."""
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
set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "f32")
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
if len(sys.argv) > 3:
    from vfmseg_amd import lib as L
    for kv in sys.argv[3].split(","):
        k, v = kv.split("=")
        assert L.load().vfm_tune(k.encode(), int(v)) == 0
STASH = {}
_cf, _cb = Fh.UpsampleCEFn.forward, Fh.UpsampleCEFn.backward
def cfwd(ctx, logits_low, label, ignore_index, loss_weight):
    r = _cf(ctx, logits_low, label, ignore_index, loss_weight)
    return r
def cbwd(ctx, dloss, _):
    pre = ctx.saved_tensors[0].abs().sum().item()
    r = _cb(ctx, dloss, _)
    print("   CE bwd: shape %s sum|dl| before %.6e dloss %.6e  sum|g| %.6e" % (tuple(r[0].shape), pre, float(dloss), r[0].abs().sum().item()))
    return r
Fh.UpsampleCEFn.forward, Fh.UpsampleCEFn.backward = staticmethod(cfwd), staticmethod(cbwd)
_f, _b = Fh.CrossAttnFn.forward, Fh.CrossAttnFn.backward
def fwd(ctx, q, kv, B, Nq, Nk, H, d):
    o = _f(ctx, q, kv, B, Nq, Nk, H, d)
    STASH[id(ctx)] = [t.clone() for t in ctx.saved_tensors]
    return o
def bwd(ctx, do):
    for nm, a, b in zip(("q", "kv", "o", "lse"), STASH.pop(id(ctx)), ctx.saved_tensors):
        if not torch.equal(a, b):
            print("   !! CrossAttnFn saved tensor %s changed between forward and backward: max diff %.3e" % (nm, (a.float() - b.float()).abs().max().item()))
    r = _b(ctx, do)
    print("   cross-attn bwd: |do| %.3e |dq| %.3e |dkv| %.3e" % (do.abs().max().item(), r[0].abs().max().item(), r[1].abs().max().item()))
    return r
Fh.CrossAttnFn.forward, Fh.CrossAttnFn.backward = staticmethod(fwd), staticmethod(bwd)
grads = []
SCALES = [float(x) for x in sys.argv[2].split(",")]
for scale in SCALES:
    Fh.manual_seed(4321)
    model.fixed_crop_box = (256, 768, 128, 640)
    model.aux_decoder.transformer_decoder.fixed_keep = keep
    opt.zero_grad()
    losses = model.loss_from_data(data) if hasattr(model, "loss_from_data") else None
    if losses is None:
        d = model.data_preprocessor(data, True) if hasattr(model, "data_preprocessor") and model.data_preprocessor is not None else data
        losses = model.loss(d["inputs"], d["data_samples"])
    loss = sum(v for k, v in losses.items() if "loss" in k)
    print("   forward: " + ", ".join("%s=%.7f" % (k, float(v)) for k, v in losses.items()))
    def walk(o, path, out, seen):
        if id(o) in seen:
            return
        seen.add(id(o))
        if torch.is_tensor(o):
            out[path] = o
        elif isinstance(o, dict):
            for k, v in o.items():
                walk(v, path + "." + str(k), out, seen)
        elif isinstance(o, (list, tuple)):
            for i, v in enumerate(o):
                walk(v, path + "[%d]" % i, out, seen)
        elif isinstance(o, torch.nn.Module):
            return
        elif hasattr(o, "__dict__"):
            for k, v in vars(o).items():
                walk(v, path + "." + k, out, seen)
    cache = {}
    for mn, m in model.named_modules():
        if getattr(m, "_engine", None) is not None:
            walk(vars(m._engine), mn + "._engine", cache, set([id(m)]))
    walk(Fh.PACKS, "PACKS", cache, set())
    csnap = {k: v.detach().clone() for k, v in cache.items()}
    snap = {k: v.detach().clone() for k, v in list(model.named_parameters()) + list(model.named_buffers())}
    flat_snap = opt.flat.clone()
    (loss * scale).backward()
    torch.cuda.synchronize()
    for k, v in list(model.named_parameters()) + list(model.named_buffers()):
        if not torch.equal(snap[k], v.detach()):
            print("   !! backward changed %s: max diff %.3e (max|p| %.3e)" % (k, (snap[k].float() - v.detach().float()).abs().max().item(), snap[k].float().abs().max().item()))
    for k, v in cache.items():
        if v.shape == csnap[k].shape and not torch.equal(csnap[k], v.detach()):
            print("   !! backward changed cached tensor %s %s: max diff %.3e" % (k, tuple(v.shape), (csnap[k].float() - v.detach().float()).abs().max().item()))
    if not torch.equal(flat_snap, opt.flat):
        print("   !! backward changed the flat parameter buffer")
    grads.append(opt.gflat.clone() / scale)
def worst(a, b):
    rows = []
    for nm, o0, o1 in zip(opt.names, opt.offsets[:-1], opt.offsets[1:]):
        x, y = a[o0:o1], b[o0:o1]
        if nm == "decode_head.output_upscaling.0.bias":
            continue
        rows.append(((x - y).abs().max().item() / max(x.abs().max().item(), 1e-30), nm, x.abs().max().item()))
    rows.sort(reverse=True)
    return rows
for i in range(1, min(len(grads), 2)):
    r = worst(grads[0], grads[i])
    print("run %d vs run 0: " % i + "; ".join("%.2e %s" % (x[0], x[1][-40:]) for x in r[:3]))

sys.exit(0)
for nm, o0, o1 in zip(opt.names, opt.offsets[:-1], opt.offsets[1:]):
    if nm.startswith("aux_decoder"):
        x, y = grads[1][o0:o1], grads[2][o0:o1]
        print("  %.2e  %s" % ((x - y).abs().max().item() / max(x.abs().max().item(), 1e-30), nm))
