#!/usr/bin/env python
"""Evaluation entry point (tools/test.py:19-145): CONFIG CHECKPOINT [--backbone PTH --work-dir --out DIR --show --show-dir DIR
--wait-time S --cfg-options ... --launcher ... --tta].
Runs the configured test mode (ms_slide_inference / slide) over the config's test_dataloader.dataset when its data_root exists
(test pipeline: LoadImageFromFile, Resize(keep_ratio), LoadAnnotations, PackSegInputs -> SegDataPreProcessor -> predict ->
postprocess to ori_shape), else over synthetic images, and reports mIoU against the labels (mmseg IoUMetric / DGIoUMetric
semantics) and ms/img."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("checkpoint", nargs="?")
    ap.add_argument("--backbone", help="backbone .pth merged under 'backbone.' (LoadBackboneHook semantics)")
    ap.add_argument("--work-dir")
    ap.add_argument("--cfg-options", nargs="+")
    ap.add_argument("--out", type=str, help="directory for the predicted label maps (PNG, one per image; mmseg IoUMetric output_dir)")
    ap.add_argument("--show", action="store_true", help="accepted for CLI parity; there is no display here: use --show-dir")
    ap.add_argument("--show-dir", help="directory for colour-painted predictions (Cityscapes palette)")
    ap.add_argument("--wait-time", type=float, default=2)
    ap.add_argument("--launcher", choices=["none", "pytorch", "slurm", "mpi"], default="none")
    ap.add_argument("--tta", action="store_true", help="test-time augmentation from the config's tta_pipeline (scales x flip, mean of softmax)")
    ap.add_argument("--local_rank", "--local-rank", type=int, default=0)
    ap.add_argument("--images", type=int, default=None,
                    help="evaluate only the first N samples (default: the whole test set on real data, 4 synthetic images otherwise)")
    ap.add_argument("--size", type=int, nargs=2, default=[1024, 1024])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "bf16x3", "fp16"])
    ap.add_argument("--data", choices=["auto", "synthetic", "real"], default="auto")
    a = ap.parse_args()
    if "LOCAL_RANK" not in os.environ:
        os.environ["LOCAL_RANK"] = str(a.local_rank)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # Evaluation here is one process (a 1024^2 prediction takes 15 ms): there is no sharded test loop and no reduction of the
        # confusion matrix, so several ranks would each score the whole test set on the default device and print their own result.
        raise SystemExit("tools/test.py: distributed evaluation (--launcher %s with WORLD_SIZE=%s) is not implemented; run ONE process "
                         "(python tools/test.py CONFIG CHECKPOINT)" % (a.launcher, os.environ["WORLD_SIZE"]))
    import numpy as np
    import torch
    import vfmseg_amd  # noqa: F401
    from vfmseg_amd.config import Config, parse_cfg_options
    from vfmseg_amd.registry import METRICS
    from vfmseg_amd.segmentors import SegDataSample
    from vfmseg_amd.precision import set_compute_dtype
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.synth import synth_image, synth_label, synth_like
    set_compute_dtype(a.dtype)
    cfg = Config.fromfile(a.config)
    cfg.merge_from_dict(parse_cfg_options(a.cfg_options))
    model = MODELS.build(cfg["model"])
    sd = synth_like(model.state_dict())
    if a.checkpoint:
        ck = torch.load(a.checkpoint, map_location="cpu")
        sd.update(ck.get("state_dict", ck))
    if a.backbone:
        sd.update({"backbone." + k: v for k, v in torch.load(a.backbone, map_location="cpu").items()})
    model.load_state_dict(sd, strict=False)
    torch.cuda.set_device(int(os.environ["LOCAL_RANK"]) % max(torch.cuda.device_count(), 1))
    model = model.cuda().eval()
    tta = None
    if a.tta:   # tools/test.py:131-134 swaps in cfg.tta_pipeline / cfg.tta_model (mmseg SegTTAModel: mean of the views' softmax)
        if "tta_pipeline" not in cfg:
            raise SystemExit("--tta: the config defines no tta_pipeline (the reference raises on cfg.tta_pipeline too)")
        from vfmseg_amd.segmentors import tta_views
        tta = tta_views(cfg["tta_pipeline"])
    for d in (a.out, a.show_dir):
        if d:
            os.makedirs(d, exist_ok=True)

    def predict(inputs, samples):
        if tta is None:
            return model.predict(inputs, samples)
        from vfmseg_amd.segmentors import predict_tta
        return predict_tta(model, inputs, samples, tta)

    def dump(out, idx):
        if not (a.out or a.show_dir):
            return
        from PIL import Image
        from vfmseg_amd.datasets import CITYSCAPES_PALETTE
        for j, o in enumerate(out):
            pred = o.pred_sem_seg.data.squeeze().to(torch.uint8).cpu().numpy()
            name = os.path.splitext(os.path.basename(str((o.metainfo or {}).get("img_path") or (o.metainfo or {}).get("seg_map_path") or f"{idx}_{j}")))[0]
            if a.out:
                Image.fromarray(pred).save(os.path.join(a.out, name + ".png"))
            if a.show_dir:
                pal = np.asarray(CITYSCAPES_PALETTE, dtype=np.uint8)
                Image.fromarray(pal[np.minimum(pred, len(pal) - 1)]).save(os.path.join(a.show_dir, name + ".png"))
    ev = cfg.get("test_evaluator") or cfg.get("val_evaluator") or dict(type="IoUMetric")
    ev = dict(ev[0] if isinstance(ev, (list, tuple)) else ev)
    metric = METRICS.build(ev)      # DGIoUMetric (rein/dg_metrics.py) with the config's dataset_keys, or mmseg's IoUMetric
    keys = list(getattr(metric, "dataset_keys", [])) or ["synthetic"]
    t = 0.0
    dl = dict(cfg.get("test_dataloader") or cfg.get("val_dataloader") or {})
    ds_cfg = dl.get("dataset")
    root = (ds_cfg.get("source", ds_cfg) if isinstance(ds_cfg, dict) else {}).get("data_root") if ds_cfg else None
    real = a.data == "real" or (a.data == "auto" and root and os.path.isdir(root))
    if real:
        from vfmseg_amd.datasets import DataLoaderIter
        it = DataLoaderIter(ds_cfg, 1, dl.get("num_workers", 0), shuffle=False, infinite=False)
        if hasattr(it.dataset, "metainfo"):
            metric.dataset_meta = dict(classes=it.dataset.metainfo["classes"])
        n = 0
        for batch in it.loader:
            data = model.data_preprocessor(batch, False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = predict(data["inputs"], data["data_samples"])
            torch.cuda.synchronize()
            t += time.perf_counter() - t0
            metric.process(None, out)
            dump(out, n)
            n += 1
            if a.images and n >= a.images:     # only when --images was given: the reference evaluates the whole test_dataloader
                break
        res = dict(metric.evaluate(n))
        res["ms_per_img"] = 1e3 * t / max(n, 1)
        res["evaluated_samples"] = n
        res["dataset_size"] = len(it.dataset)
        print(res)
        return
    a.images = a.images or 4
    for i in range(a.images):
        img, lab = synth_image(1, tuple(a.size), seed=500 + i).cuda(), synth_label(1, tuple(a.size), seed=500 + i).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sample = SegDataSample(gt_sem_seg=lab[0], metainfo=dict(seg_map_path=f"{keys[i % len(keys)]}/synthetic_{i}.png",
                                                                 ori_shape=tuple(a.size), img_shape=tuple(a.size), padding_size=[0, 0, 0, 0]))
        out = predict(img, [sample])
        torch.cuda.synchronize()
        t += time.perf_counter() - t0
        metric.process(None, out)
        dump(out, i)
    res = dict(metric.evaluate(a.images))
    res["ms_per_img"] = 1e3 * t / a.images
    res["evaluated_samples"] = a.images
    print(res)


if __name__ == "__main__":
    main()
