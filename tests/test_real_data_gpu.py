"""Real-data path end to end on a tiny Cityscapes-shaped tree: DGDataset (rare class sampling) -> mmseg transform chain ->
uint8 batches -> SegDataPreProcessor (HIP kernel: BGR->RGB, normalise, pad) -> MsVFMEncoderDecoder.train_step through the Runner,
then the test pipeline (keep-ratio resize, labels at original size) -> predict -> DGIoUMetric."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tree(root, n=3, h=1100, w=1200):
    from PIL import Image
    rng = np.random.RandomState(0)
    os.makedirs(os.path.join(root, "images"))
    os.makedirs(os.path.join(root, "labels"))
    stats, swc = [], {}
    for i in range(n):
        base = rng.randint(0, 256, (h // 50 + 1, w // 50 + 1, 3), dtype=np.uint8)
        img = np.kron(base, np.ones((50, 50, 1), np.uint8))[:h, :w]
        lab = (rng.randint(0, 19, (h // 100 + 1, w // 100 + 1), dtype=np.uint8))
        lab = np.kron(lab, np.ones((100, 100), np.uint8))[:h, :w].copy()
        lab[:40] = 255
        Image.fromarray(img).save(os.path.join(root, "images", f"citys_{i}.png"))
        fn = os.path.join(root, "labels", f"citys_{i}_labelTrainIds.png")
        Image.fromarray(lab).save(fn)
        st = {"file": fn}
        for c in np.unique(lab):
            if c != 255:
                st[str(int(c))] = int((lab == c).sum())
                swc.setdefault(str(int(c)), []).append([fn, int((lab == c).sum())])
        stats.append(st)
    json.dump(stats, open(os.path.join(root, "sample_class_stats.json"), "w"))
    json.dump(swc, open(os.path.join(root, "samples_with_class.json"), "w"))


def test_train_and_eval_on_a_folder_dataset(tmp_path):
    import vfmseg_amd  # noqa: F401
    from vfmseg_amd import presets
    from vfmseg_amd.datasets import DataLoaderIter
    from vfmseg_amd.precision import set_compute_dtype
    from vfmseg_amd.registry import METRICS
    from vfmseg_amd.runner import Runner
    root = str(tmp_path / "data")
    _tree(root)
    set_compute_dtype("bf16")
    depth = 4
    model_cfg = presets.dinov2_ms_masked(depth=depth)
    model_cfg["backbone"]["backbone"]["out_indices"] = [0, 1, 2, 3]
    train_pipe = [dict(type="LoadImageFromFile"), dict(type="LoadAnnotations"), dict(type="Resize", scale=(1280, 1152)),
                  dict(type="RandomCrop", crop_size=(1024, 1024), cat_max_ratio=0.75), dict(type="RandomFlip", prob=0.5),
                  dict(type="PhotoMetricDistortion"), dict(type="PackSegInputs")]
    test_pipe = [dict(type="LoadImageFromFile"), dict(type="Resize", scale=(1024, 1024), keep_ratio=True), dict(type="LoadAnnotations"),
                 dict(type="PackSegInputs")]
    src = dict(type="CityscapesDataset", data_root=root, data_prefix=dict(img_path="images", seg_map_path="labels"), img_suffix=".png",
               seg_map_suffix="_labelTrainIds.png")
    oc = presets.optim_cfg()
    cfg = dict(model=model_cfg, optim_wrapper=oc["optim_wrapper"], param_scheduler=oc["param_scheduler"], randomness=dict(seed=0),
               work_dir=str(tmp_path / "work"), train_cfg=dict(max_iters=2),
               train_dataloader=dict(batch_size=2, num_workers=0, sampler=dict(type="InfiniteSampler", shuffle=True),
                                     dataset=dict(type="DGDataset", source=dict(src, pipeline=train_pipe),
                                                  rare_class_sampling=dict(class_temp=0.01, min_crop_ratio=2, min_pixels=3000))))
    runner = Runner.from_cfg(cfg, synthetic=False)
    batch = next(runner.loader)
    assert batch["inputs"][0].dtype == torch.uint8 and tuple(batch["inputs"][0].shape) == (3, 1024, 1024)
    pre = runner.model.data_preprocessor(dict(inputs=[t.clone() for t in batch["inputs"]], data_samples=batch["data_samples"]), True)
    x = pre["inputs"]
    assert x.dtype == torch.float32 and tuple(x.shape) == (2, 3, 1024, 1024)
    mean, std = model_cfg["data_preprocessor"]["mean"], model_cfg["data_preprocessor"]["std"]
    ref = (batch["inputs"][0].float().flip(0) - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)   # BGR -> RGB
    assert (x[0].cpu() - ref).abs().max().item() < 1e-4
    runner.train(max_iters=2, log_interval=1, ckpt_interval=0)
    rec = [json.loads(l) for l in open(os.path.join(cfg["work_dir"], "scalars_rank0.jsonl"))]
    assert len(rec) == 2 and all(np.isfinite(r["decode_lr.loss_ce"]) and np.isfinite(r["decode_hr.loss_ce"]) for r in rec)
    # evaluation: labels stay at the original 1100 x 1200, the image is resized to fit 1024 x 1024, predictions come back at ori_shape
    model = runner.model.eval()
    metric = METRICS.build(dict(type="DGIoUMetric", iou_metrics=["mIoU"], dataset_keys=["citys"]))
    it = DataLoaderIter(dict(src, pipeline=test_pipe), 1, 0, shuffle=False, infinite=False)
    n = 0
    with torch.no_grad():
        for b in it.loader:
            assert tuple(b["inputs"][0].shape) == (3, 939, 1024) and tuple(b["data_samples"][0].gt_sem_seg.data.shape) == (1, 1100, 1200)
            data = model.data_preprocessor(b, False)
            out = model.predict(data["inputs"], data["data_samples"])
            assert tuple(out[0].pred_sem_seg.data.shape[-2:]) == (1100, 1200)
            metric.process(None, out)
            n += 1
    res = metric.evaluate(n)
    assert n == 3 and "citys_mIoU" in res and 0.0 <= res["citys_mIoU"] <= 100.0
