"""SURVEY 8 row f1 on the HIP path: confusion-histogram kernel, IoUMetric / DGIoUMetric (per-dataset grouping, mean_* keys),
postprocess_result (un-pad, flip undo, resize to ori_shape), the lr / msfull sliding modes, and the north_star target
"mIoU within +-0.1 of reference on a fixed synthetic batch" stated from HIP predictions vs oracle predictions."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vfmseg_amd  # noqa: E402,F401
from oracle import torch_ref as R  # noqa: E402
from tests.helpers import full_state_dict, rel_err, sl  # noqa: E402
from vfmseg_amd import functional as Fh, metrics as M, ops, presets  # noqa: E402
from vfmseg_amd.precision import set_compute_dtype  # noqa: E402
from vfmseg_amd.registry import METRICS, MODELS  # noqa: E402
from vfmseg_amd.segmentors import PixelData, SegDataSample  # noqa: E402
from vfmseg_amd.synth import synth_image, synth_label  # noqa: E402


def _noisy_pred(lab, seed, flip=0.3):
    g = torch.Generator().manual_seed(seed)
    pred = lab.clone()
    pred[pred == 255] = 3
    noise = torch.randint(0, 19, lab.shape, generator=g)
    take = torch.rand(lab.shape, generator=g) < flip
    return torch.where(take, noise, pred).to(torch.uint8)


@pytest.mark.parametrize("n,ldt", [(0, torch.int64), (7, torch.int64), (16, torch.uint8), (1000, torch.int64), (512 * 512 + 5, torch.uint8),
                                   (1024 * 2048, torch.int64)])
def test_confusion_hist_kernel_exact(n, ldt):
    """bit-exact against a torch.bincount of label * nc + pred over the valid pixels, incl. empty / ragged sizes, labels
    outside the class range (row nc) and the ignore value."""
    nc = 19
    g = torch.Generator().manual_seed(n + 1)
    pred = torch.randint(0, nc, (n,), generator=g, dtype=torch.uint8)
    lab = torch.randint(0, nc, (n,), generator=g)
    r = torch.rand(n, generator=g)
    lab[r < 0.07] = 255
    lab[(r > 0.07) & (r < 0.09)] = 200          # not ignore, not a class: counts into the prediction areas only
    want = torch.zeros((nc + 1) * nc, dtype=torch.int64)
    v = lab != 255
    rows = torch.where(lab[v] < nc, lab[v], torch.full_like(lab[v], nc))
    if v.any():
        want += torch.bincount(rows * nc + pred[v].long(), minlength=(nc + 1) * nc)
    hist = torch.zeros((nc + 1) * nc, dtype=torch.int64, device="cuda")
    ops.confusion_hist(pred.cuda(), lab.to(ldt).cuda(), hist, nc, 255)
    ops.confusion_hist(pred.cuda(), lab.to(ldt).cuda(), hist, nc, 255)   # accumulates
    assert torch.equal(hist.cpu(), 2 * want)
    # the areas mmseg derives from it
    if n:
        ai, au, ap, al = R.intersect_and_union(pred.long(), lab, nc)
        gi, gu, gp, gl = M.areas_from_confusion(want.numpy(), nc)
        np.testing.assert_array_equal(gi, ai.numpy()), np.testing.assert_array_equal(gu, au.numpy())
        np.testing.assert_array_equal(gp, ap.numpy()), np.testing.assert_array_equal(gl, al.numpy())


def test_dg_iou_metric_groups_like_the_reference():
    """rein/dg_metrics.py:24-102 semantics vs the oracle restatement: three datasets + an unknown path, batches of 1 and 2
    (a batch is filed under its FIRST sample's key), mean over mean_used_keys only."""
    keys, used = ["citys", "bdd", "map"], ["citys", "bdd"]
    metric = METRICS.build(dict(type="DGIoUMetric", dataset_keys=keys, mean_used_keys=used))
    paths = [["data/citys/a.png"], ["data/bdd/b.png", "data/citys/zz.png"], ["data/map/c.png"], ["elsewhere/d.png"], ["x/citys/e.png"]]
    batches = []
    k = 0
    for ps in paths:
        batch, samples = [], []
        for pth in ps:
            lab = synth_label(1, 256, seed=80 + k)[0, 0]
            pred = _noisy_pred(lab, 90 + k, flip=0.2 + 0.1 * (k % 3))
            batch.append((pred, lab, pth))
            ds = SegDataSample(gt_sem_seg=lab.unsqueeze(0).cuda(), metainfo=dict(seg_map_path=pth))
            ds.pred_sem_seg = PixelData(pred.unsqueeze(0).cuda())
            samples.append(ds)
            k += 1
        metric.process(None, samples)
        batches.append(batch)
    got = metric.evaluate(k)
    want = R.dg_iou_metrics(batches, keys, used)
    assert set(got) == set(want), (sorted(got), sorted(want))
    for kk in want:
        assert abs(got[kk] - want[kk]) < 1e-9, (kk, got[kk], want[kk])
    assert "unknown_mIoU" in got and "mean_mIoU" in got and "map_mIoU" in got
    assert abs(got["mean_mIoU"] - (got["citys_mIoU"] + got["bdd_mIoU"]) / 2) < 1e-9
    # dict-form samples (what mmengine's evaluator passes) through the plain IoUMetric
    m2 = METRICS.build(dict(type="IoUMetric"))
    lab = synth_label(1, 256, seed=99)[0, 0]
    pred = _noisy_pred(lab, 98)
    m2.process(None, [dict(pred_sem_seg=dict(data=pred.cuda()), gt_sem_seg=dict(data=lab.cuda()))])
    res = m2.evaluate(1)
    assert res == R.iou_metric_summary([R.intersect_and_union(pred.long(), lab)])


def _small_model(mode):
    set_compute_dtype(mode)
    depth = 4
    cfg = presets.dinov2_ms_masked(depth=depth)
    cfg["backbone"]["backbone"]["out_indices"] = [0, 1, 2, 3]
    sd = full_state_dict(depth=depth)
    model = MODELS.build(cfg)
    model.load_state_dict(sd)
    return model.cuda().eval(), sd, dict(depth=depth, out_indices=(0, 1, 2, 3))


def test_postprocess_result_unpads_flips_and_resizes():
    model, _, _ = _small_model("f32")
    try:
        g = torch.Generator().manual_seed(5)
        logits = torch.randn(2, 19, 96, 128, generator=g)
        metas = [dict(padding_size=(0, 28, 0, 16), ori_shape=(160, 200), flip=True, flip_direction="horizontal"),
                 dict(img_padding_size=(4, 0, 8, 0), ori_shape=(88, 124), flip=True, flip_direction="vertical")]
        out = model.postprocess_result(logits.cuda().clone(), [SegDataSample(), SegDataSample()], metas)
        ref = R.postprocess_result(logits, metas)
        for o, (rl, rp) in zip(out, ref):
            assert tuple(o.seg_logits.data.shape) == tuple(rl.shape)
            assert rel_err(o.seg_logits.data.cpu(), rl) < 1e-5
            mism = (o.pred_sem_seg.data.cpu().long() != rp)
            top2 = rl.topk(2, dim=0)[0]
            assert mism.float().mean() < 1e-3 and (mism.sum() == 0 or (top2[0] - top2[1])[mism[0]].max() < 1e-4)
        # no padding / flip / resize: the batch path returns the logits themselves
        out = model.postprocess_result(logits.cuda(), None, [dict(ori_shape=(96, 128), padding_size=[0] * 4)] * 2)
        assert torch.equal(out[1].pred_sem_seg.data[0].cpu().long(), logits[1].argmax(0))
    finally:
        set_compute_dtype("bf16")


def test_lr_and_msfull_slide_modes_match_oracle():
    """Ms_VFM_encoder_decoder.py:276-332: `lr_slide_inference` (the config default) and `msfull_slide_inference` on a 1024^2
    image, depth 4, fp32 parity mode, with the msfull query masks injected on both sides."""
    model, sd, kw = _small_model("f32")
    try:
        img = synth_image(1, 1024, seed=71)
        model.test_cfg["mode"] = "lr_slide_inference"
        with torch.no_grad():
            got = model.inference(img.cuda(), [{}]).cpu()
            ref = R.lr_slide_inference(sd, img, **kw)
        assert got.shape == ref.shape and rel_err(got, ref) < 1e-3, rel_err(got, ref)
        model.test_cfg["mode"] = "msfull_slide_inference"
        keeps = torch.rand(9, 1, 1, 32, 32, generator=torch.Generator().manual_seed(6)) > 0.2
        model.aux_decoder.transformer_decoder.fixed_keep = keeps.reshape(9, 1, 32, 32)
        with torch.no_grad():
            got = model.inference(img.cuda(), [{}]).cpu()
            ref = R.msfull_slide_inference(sd, img, mask_keeps=[keeps[j] for j in range(9)], **kw)
        assert rel_err(got, ref) < 1e-3, rel_err(got, ref)
    finally:
        set_compute_dtype("bf16")


# (bf16 bounds = 3x the measured values: logits 8.6e-3, mismatches 8.3e-3, margin 3.2e-3 - profiles/r03_parity_gpu_suite.log)
@pytest.mark.gpu
def test_coarse_pass_beside_the_window_pass_is_bit_identical(monkeypatch):
    """msfull_slide_inference, and ms_slide_inference with a gate that cannot fire (conf > 1), run the coarse 512 x 1024 pass on a side
    stream beside the windows' backbone pass (segmentors._coarse_beside): same kernels on the same data, so the predictions must equal
    the one-stream path bit for bit - including the first prediction after a weight change, which stays on one stream."""
    model, _, _ = _small_model("bf16")
    try:
        img = synth_image(1, 1024, seed=5).cuda()
        for mode, conf in (("ms_slide_inference", 2.0), ("msfull_slide_inference", 0.8)):
            model.test_cfg["mode"], model.test_cfg["conf"] = mode, conf
            monkeypatch.setenv("VFMSEG_EVAL_OVERLAP", "0")
            with torch.no_grad():
                Fh.manual_seed(11)     # (msfull draws the decoder's query mask per call: Ms_VFM_encoder_decoder.py:286-328)
                ref = model.inference(img, [{}])
            monkeypatch.setenv("VFMSEG_EVAL_OVERLAP", "1")
            model._overlap_key = None
            assert not model._overlap_ok() and model._overlap_ok()       # first call after a change: one stream; then beside
            with torch.no_grad():
                for _ in range(3):
                    Fh.manual_seed(11)
                    got = model.inference(img, [{}])
                    assert torch.equal(got, ref), mode
            if mode == "ms_slide_inference":
                assert len(model.last_refined) == 9
        # a live gate keeps the order coarse -> gate -> windows
        model.test_cfg["mode"], model.test_cfg["conf"], model.test_cfg["threadshod"] = "ms_slide_inference", 0.8, 0.5
        with torch.no_grad():
            a = model.inference(img, [{}])
            monkeypatch.setenv("VFMSEG_EVAL_OVERLAP", "0")
            b = model.inference(img, [{}])
        assert torch.equal(a, b)
    finally:
        set_compute_dtype("bf16")


@pytest.mark.parametrize("prec,ltol,mtol", [("f32", 1e-3, 2e-4), ("bf16x3", 1e-3, 2e-4), ("bf16", 2.6e-2, 2.5e-2),
                                                ("fp16", 1e-3, 5e-3)])
def test_slide_modes_match_reference_goldens(golden_dir, prec, ltol, mtol):
    """lr_slide_inference / hr_slide_inference / msfull_slide_inference at full depth on the HIP path against the reference's OWN
    MsVFMEncoderDecoder.inference output in those modes (tests/golden/slide_modes.npz; Ms_VFM_encoder_decoder.py:278-332), msfull with
    the reference's recorded query masks.  f32 and bf16x3 claim north_star's tolerance; bf16 (the timed mode) is measured and bounded."""
    from tests.helpers import cached_full_state_dict
    G = np.load(os.path.join(golden_dir, "slide_modes.npz"))
    set_compute_dtype(prec)
    try:
        model = MODELS.build(presets.dinov2_ms_masked())
        model.load_state_dict(cached_full_state_dict(), strict=False)
        model = model.cuda().eval()
        img = synth_image(1, 1024, seed=11).cuda()
        for mode in ("lr_slide_inference", "hr_slide_inference", "msfull_slide_inference"):
            model.test_cfg["mode"] = mode
            model.aux_decoder.transformer_decoder.fixed_keep = (
                torch.from_numpy(G["msfull_mask_rand"]).reshape(9, 1, 32, 32) > 0.2 if mode == "msfull_slide_inference" else None)
            with torch.no_grad():
                out = model.predict(img)
            logits = out[0].seg_logits.data.unsqueeze(0)
            e1, e2 = rel_err(sl(logits), G[mode + "::logits_slice"]), rel_err(logits[0, :, 500:504, 636:644], G[mode + "::logits_center"])
            pred = out[0].pred_sem_seg.data[0].cpu().numpy().astype(np.uint8)
            diff = pred[::4, ::4] != G[mode + "::pred_sub4"]
            top2 = torch.topk(logits[0, :, ::4, ::4], 2, dim=0).values
            margin = ((top2[0] - top2[1]) / (logits.max() - logits.min())).cpu().numpy()
            worst = float(margin[diff].max()) if diff.any() else 0.0
            print(f"[parity] {mode} {prec}: logits rel err {max(e1, e2):.2e}, argmax mismatches {diff.mean():.2e}, largest top-2 margin among them {worst:.2e}")
            assert max(e1, e2) < ltol and diff.mean() < mtol
            assert worst < {"bf16": 1e-2, "fp16": 2e-3}.get(prec, 1e-4)   # every flipped pixel is a near-tie at that mode's precision
    finally:
        set_compute_dtype("bf16")


_ORACLE_MS = {}   # image index -> argmax of the oracle's ms_inference (fp32 CPU; the same for every precision mode of the HIP path)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_miou_within_0p1_of_oracle_on_fixed_synthetic_batch(mode):
    """north_star: "mIoU within +-0.1 of reference on a fixed synthetic batch".  Predictions of the HIP path (ms_slide_inference,
    depth 4, 3 fixed 1024^2 images) vs predictions of the oracle on the same images, both scored against the same synthetic
    labels with the reference's metric (DGIoUMetric semantics)."""
    model, sd, kw = _small_model(mode)
    try:
        model.test_cfg["mode"] = "ms_slide_inference"
        metric = METRICS.build(dict(type="DGIoUMetric", dataset_keys=["citys"]))
        ref_batches = []
        for i in range(3):
            img, lab = synth_image(1, 1024, seed=300 + i), synth_label(1, 1024, seed=300 + i)
            with torch.no_grad():
                out = model.predict(img.cuda(), [SegDataSample(gt_sem_seg=lab[0].cuda(), metainfo=dict(seg_map_path=f"citys/{i}.png", ori_shape=(1024, 1024)))])
                if i not in _ORACLE_MS:
                    _ORACLE_MS[i] = R.ms_inference(sd, img, thr=model.test_cfg["threadshod"], conf=model.test_cfg["conf"], **kw).argmax(1)[0]
            metric.process(None, out)
            ref_batches.append([(_ORACLE_MS[i], lab[0, 0], f"citys/{i}.png")])
        got, want = metric.evaluate(3), R.dg_iou_metrics(ref_batches, ["citys"])
        print(f"[mIoU {mode}] HIP {got} | oracle {want}")
        for k in ("citys_mIoU", "citys_mAcc", "citys_aAcc", "mean_mIoU"):
            assert abs(got[k] - want[k]) <= 0.1, (k, got[k], want[k])
    finally:
        set_compute_dtype("bf16")


def _colour_coded(seed, size=1024):
    """A synthetic sample whose image carries its label: every class has a colour, plus noise - something a few dozen optimiser steps
    can learn, so that mIoU is measured at an operating point where predictions carry signal (random-init weights score ~2 %)."""
    lab = synth_label(1, size, seed=seed)
    g = torch.Generator().manual_seed(9000 + seed)
    table = torch.randn(256, 3, generator=torch.Generator().manual_seed(77)) * 1.5
    img = table[lab[0, 0].clamp(max=255)].permute(2, 0, 1).unsqueeze(0) + 0.3 * torch.randn(1, 3, size, size, generator=g)
    return img.contiguous(), lab


_TRAINED = {}   # the 60-step training run (bf16) and the oracle's predictions with its weights: shared by the precision modes under test


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16", "fp16"])
def test_miou_after_training_within_0p1_of_oracle(mode):
    """Round-2 verdict: the +-0.1 mIoU target was only tested at random-init weights (mIoU 2.3 % on both sides).  Here a depth-4 model is
    first trained for 60 optimiser steps (product train_step, bf16, lr 1e-3) on three colour-coded 1024^2 samples; the trained weights
    then score the same images through ms_slide_inference on the HIP path (mode under test) and through the oracle, both against the
    labels with DGIoUMetric semantics."""
    from vfmseg_amd.optim import PEFTOptimWrapperConstructor
    set_compute_dtype("bf16")
    depth = 4
    kw = dict(depth=depth, out_indices=(0, 1, 2, 3))
    try:
        cfg = presets.dinov2_ms_masked(depth=depth)
        cfg["backbone"]["backbone"]["out_indices"] = [0, 1, 2, 3]
        if "sd" not in _TRAINED:
            model = MODELS.build(cfg)
            model.load_state_dict(full_state_dict(depth=depth))
            model = model.cuda().train()
            oc = presets.optim_cfg()
            oc["optim_wrapper"]["optimizer"]["lr"] = 1e-3
            ow = PEFTOptimWrapperConstructor(oc["optim_wrapper"])(model, oc["param_scheduler"])
            np.random.seed(5)
            samples = [_colour_coded(400 + i) for i in range(3)]
            for t in range(60):
                img, lab = samples[t % 3]
                model.train_step(dict(inputs=img.cuda(), data_samples=[SegDataSample(gt_sem_seg=lab[0])]), ow)
            _TRAINED["sd"] = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
            _TRAINED["samples"] = samples
            _TRAINED["ref"] = {}
            del model, ow
        sd, samples = _TRAINED["sd"], _TRAINED["samples"]
        set_compute_dtype(mode)
        model = MODELS.build(cfg)
        model.load_state_dict(sd)
        model = model.cuda().eval()
        model.test_cfg["mode"] = "ms_slide_inference"
        metric = METRICS.build(dict(type="DGIoUMetric", dataset_keys=["citys"]))
        ref_batches, flips = [], []
        for i, (img, lab) in enumerate(samples):
            with torch.no_grad():
                out = model.predict(img.cuda(), [SegDataSample(gt_sem_seg=lab[0].cuda(), metainfo=dict(seg_map_path=f"citys/{i}.png", ori_shape=(1024, 1024)))])
                if i not in _TRAINED["ref"]:
                    _TRAINED["ref"][i] = R.ms_inference(sd, img, thr=model.test_cfg["threadshod"], conf=model.test_cfg["conf"], **kw).argmax(1)[0]
            ref_am = _TRAINED["ref"][i]
            metric.process(None, out)
            ref_batches.append([(ref_am, lab[0, 0], f"citys/{i}.png")])
            flips.append((out[0].pred_sem_seg.data[0].cpu().long() != ref_am).float().mean().item())
        got, want = metric.evaluate(3), R.dg_iou_metrics(ref_batches, ["citys"])
        print(f"[parity] mIoU after 60 train steps, {mode}: HIP {got} | oracle {want} | argmax mismatches {max(flips):.2e}")
        assert want["citys_mIoU"] > 10.0, "the trained model must carry signal for this test to mean anything"
        for k in ("citys_mIoU", "citys_mAcc", "citys_aAcc", "mean_mIoU"):
            assert abs(got[k] - want[k]) <= 0.1, (k, got[k], want[k])
    finally:
        set_compute_dtype("bf16")
