"""tools/test.py with the reference's CLI flags (tools/test.py:19-61 there): --out (prediction PNGs, mmseg IoUMetric output_dir), --show-dir
(painted predictions), --tta (SegTTAModel semantics: mean of the views' softmax) on the synthetic stream, through the real entry point."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(tmp_path, tta):
    p = tmp_path / "cfg.py"
    p.write_text(
        "from vfmseg_amd import presets as _p\n"
        "model = _p.dinov2_ms_masked(depth=4)\n"
        "model['backbone']['backbone']['out_indices'] = [0, 1, 2, 3]\n"
        "test_evaluator = dict(type='IoUMetric')\n"
        + ("tta_pipeline = [dict(type='LoadImageFromFile'), dict(type='TestTimeAug', transforms=[[dict(type='Resize', scale_factor=r, keep_ratio=True) "
           "for r in (0.5, 1.0)], [dict(type='RandomFlip', prob=0., direction='horizontal'), dict(type='RandomFlip', prob=1., direction='horizontal')], "
           "[dict(type='LoadAnnotations')], [dict(type='PackSegInputs')]])]\n" if tta else ""))
    return str(p)


def test_test_py_out_and_show_dir(tmp_path):
    from PIL import Image
    out, show = tmp_path / "out", tmp_path / "show"
    r = subprocess.run([sys.executable, "tools/test.py", _cfg(tmp_path, False), "--data", "synthetic", "--images", "2", "--size", "512", "512",
                        "--out", str(out), "--show-dir", str(show), "--launcher", "none"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "'evaluated_samples': 2" in r.stdout and "mIoU" in r.stdout
    pngs = sorted(os.listdir(out))
    assert len(pngs) == 2 and sorted(os.listdir(show)) == pngs
    pred = np.asarray(Image.open(out / pngs[0]))
    assert pred.shape == (512, 512) and pred.dtype == np.uint8 and pred.max() < 19
    assert np.asarray(Image.open(show / pngs[0])).shape == (512, 512, 3)


def test_test_py_tta_needs_a_tta_pipeline_like_the_reference(tmp_path):
    r = subprocess.run([sys.executable, "tools/test.py", _cfg(tmp_path, False), "--data", "synthetic", "--images", "1", "--size", "512", "512", "--tta"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "tta_pipeline" in (r.stderr + r.stdout)


def test_predict_tta_is_the_mean_of_the_views_softmax(tmp_path):
    import vfmseg_amd  # noqa: F401
    from tests.helpers import full_state_dict
    from vfmseg_amd import presets
    from vfmseg_amd.precision import set_compute_dtype
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.segmentors import SegDataSample, predict_tta, tta_views
    from vfmseg_amd.synth import synth_image
    set_compute_dtype("f32")
    try:
        cfg = presets.dinov2_ms_masked(depth=4)
        cfg["backbone"]["backbone"]["out_indices"] = [0, 1, 2, 3]
        cfg["test_cfg"]["mode"] = "hr_slide_inference"
        model = MODELS.build(cfg)
        model.load_state_dict(full_state_dict(depth=4))
        model = model.cuda().eval()
        img = synth_image(1, 512, seed=31).cuda()
        views = tta_views([dict(type="TestTimeAug", transforms=[[dict(type="Resize", scale_factor=1.0)],
                                                               [dict(type="RandomFlip", prob=0.0), dict(type="RandomFlip", prob=1.0)]])])
        assert views == [(1.0, False), (1.0, True)]
        got = predict_tta(model, img, None, views)[0]
        with torch.no_grad():
            plain = model.predict(img)[0].seg_logits.data
            flipped = model.predict(torch.flip(img, dims=[3]).contiguous(), [SegDataSample(metainfo=dict(
                ori_shape=(512, 512), img_shape=(512, 512), padding_size=[0, 0, 0, 0], flip=True, flip_direction="horizontal"))])[0].seg_logits.data
        want = 0.5 * (plain.softmax(0) + flipped.softmax(0))
        assert (got.seg_logits.data - want).abs().max().item() < 1e-5
        assert torch.equal(got.pred_sem_seg.data, want.argmax(0, keepdim=True))
    finally:
        set_compute_dtype("bf16")


def test_train_py_amp_runs_in_fp16_with_a_loss_scale(tmp_path):
    """tools/train.py --amp (tools/train.py:87-102 there: OptimWrapper -> AmpOptimWrapper, loss_scale='dynamic', fp16 autocast): the
    real entry point switches the wrapper, the engine runs on the fp16 twin library, the checkpoint carries the loss scaler."""
    p = tmp_path / "cfg.py"
    p.write_text(
        "from vfmseg_amd import presets as _p\n"
        "model = _p.dinov2_ms_masked(depth=4)\n"
        "model['backbone']['backbone']['out_indices'] = [0, 1, 2, 3]\n"
        "_o = _p.optim_cfg()\n"
        "optim_wrapper = _o['optim_wrapper']\n"
        "param_scheduler = _o['param_scheduler']\n"
        "train_dataloader = dict(batch_size=1)\n"
        "default_hooks = dict(logger=dict(interval=2), checkpoint=dict(interval=4))\n")
    wd = tmp_path / "wd"
    r = subprocess.run([sys.executable, "tools/train.py", str(p), "--amp", "--data", "synthetic", "--max-iters", "4", "--work-dir", str(wd)],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "precision mode fp16" in r.stdout
    ck = torch.load(wd / "iter_4.pth", map_location="cpu", weights_only=False)
    assert ck["optim_wrapper"]["loss_scaler"]["scale"] == 65536.0 and ck["optim_wrapper"]["iter"] == 4
    losses = [float(x) for x in __import__("re").findall(r"decode_lr\.loss_ce'?[=: ]+([0-9.]+)", r.stdout)]
    assert losses and all(np.isfinite(losses)), r.stdout[-1500:]
