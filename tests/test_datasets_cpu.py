"""Input pipeline (vfmseg_amd.datasets): folder dataset, the mmseg transform chain of configs/_base_/datasets/*.py, rare class
sampling (rein/datasets/uda_dataset.py:15-107) and the infinite rank-sharded loader - on a tiny Cityscapes-shaped tree written
to a temp dir."""
import json
import os

import numpy as np
import pytest
import torch

import vfmseg_amd  # noqa: F401
from vfmseg_amd import datasets as D
from vfmseg_amd.registry import DATASETS


def _tree(root, n=6, h=96, w=160):
    from PIL import Image
    rng = np.random.RandomState(0)
    os.makedirs(os.path.join(root, "images", "a"))
    os.makedirs(os.path.join(root, "labels", "a"))
    stats, swc = [], {}
    for i in range(n):
        img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        lab = np.full((h, w), 0, np.uint8)
        lab[:, w // 2:] = 1 + i % 3              # classes 1..3 on the right half
        lab[: h // 8] = 255                      # ignore band
        if i == 4:
            lab[h // 2:, : w // 4] = 18          # the rare class lives in one file only
        Image.fromarray(img).save(os.path.join(root, "images", "a", f"s{i}.png"))
        fn = os.path.join(root, "labels", "a", f"s{i}_labelTrainIds.png")
        Image.fromarray(lab).save(fn)
        st = {"file": fn}
        for c in np.unique(lab):
            if c != 255:
                st[str(int(c))] = int((lab == c).sum())
                swc.setdefault(str(int(c)), []).append([fn, int((lab == c).sum())])
        stats.append(st)
    json.dump(stats, open(os.path.join(root, "sample_class_stats.json"), "w"))
    json.dump(swc, open(os.path.join(root, "samples_with_class.json"), "w"))


def _cfg(root, crop=(64, 64), rcs=True):
    pipeline = [dict(type="LoadImageFromFile"), dict(type="LoadAnnotations"), dict(type="Resize", scale=(240, 144)),
                dict(type="RandomCrop", crop_size=crop, cat_max_ratio=0.75), dict(type="RandomFlip", prob=0.5),
                dict(type="PhotoMetricDistortion"), dict(type="PackSegInputs")]
    src = dict(type="CityscapesDataset", data_root=root, data_prefix=dict(img_path="images", seg_map_path="labels"), img_suffix=".png",
               seg_map_suffix="_labelTrainIds.png", pipeline=pipeline, serialize_data=False)
    cfg = dict(type="DGDataset", source=src)
    if rcs:
        cfg["rare_class_sampling"] = dict(class_temp=0.01, min_crop_ratio=2, min_pixels=100)
    return cfg


def test_folder_dataset_and_transform_chain(tmp_path):
    root = str(tmp_path)
    _tree(root)
    ds = DATASETS.build(_cfg(root, rcs=False))
    assert len(ds) == 6 and ds.CLASSES[0] == "road" and ds.ignore_index == 255
    assert ds.source.data_list[2]["seg_map_path"].endswith(os.path.join("labels", "a", "s2_labelTrainIds.png"))
    np.random.seed(0)
    s = ds[1]
    assert s["inputs"].dtype == torch.uint8 and tuple(s["inputs"].shape) == (3, 64, 64)
    gt = s["data_samples"].gt_sem_seg.data
    assert gt.dtype == torch.int64 and tuple(gt.shape) == (1, 64, 64) and set(gt.unique().tolist()) <= {0, 2, 255}
    m = s["data_samples"].metainfo
    assert m["ori_shape"] == (96, 160) and m["img_shape"] == (64, 64) and m["scale_factor"] == (1.5, 1.5) and m["flip"] in (True, False)


def test_resize_crop_flip_keep_image_and_label_aligned(tmp_path):
    root = str(tmp_path)
    _tree(root)
    res = dict(img_path=os.path.join(root, "images", "a", "s1.png"), seg_map_path=os.path.join(root, "labels", "a", "s1_labelTrainIds.png"),
               seg_fields=[])
    res = D.LoadAnnotations()(D.LoadImageFromFile()(res))
    res["img"][..., 0] = res["gt_seg_map"]        # paint the label into the blue channel: the two must move together
    np.random.seed(3)
    res = D.RandomCrop((48, 80), cat_max_ratio=0.75)(D.Resize(scale=(320, 192))(res))
    labels, cnt = np.unique(res["gt_seg_map"], return_counts=True)
    cnt = cnt[labels != 255]
    assert len(cnt) > 1 and cnt.max() / cnt.sum() < 0.75
    res = D.RandomFlip(prob=1.0)(res)
    assert res["flip"] and res["flip_direction"] == "horizontal"
    # nearest label resize and bilinear image resize agree wherever the image is away from a label edge
    same = res["img"][..., 0] == res["gt_seg_map"]
    assert same.mean() > 0.9
    kr = D.Resize(scale=(2048, 64), keep_ratio=True)(dict(img=np.zeros((96, 160, 3), np.uint8), seg_fields=[]))
    assert kr["img_shape"] == (64, 107)            # mmcv.rescale_size: the short edge decides, int(x + 0.5)


def test_hsv_round_trip_and_photometric_range():
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, (40, 50, 3), dtype=np.uint8)
    back = D.hsv2bgr_u8(D.bgr2hsv_u8(img))
    assert np.abs(back.astype(int) - img.astype(int)).max() <= 4      # 8-bit HSV quantisation (H in 2-degree steps), as with OpenCV
    hsv = D.bgr2hsv_u8(np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128]]], np.uint8))   # BGR: blue, green, red, grey
    assert hsv[0, :, 0].tolist() == [120, 60, 0, 0] and hsv[0, :3, 1].tolist() == [255, 255, 255] and hsv[0, 3].tolist() == [0, 0, 128]
    np.random.seed(5)
    outs = [D.PhotoMetricDistortion()(dict(img=img.copy()))["img"] for _ in range(8)]
    assert all(o.dtype == np.uint8 and o.shape == img.shape for o in outs) and any((o != img).any() for o in outs)


def test_rare_class_sampling(tmp_path):
    root = str(tmp_path)
    _tree(root)
    classes, probs = D.get_rcs_class_probs(root, 0.01)
    stats = json.load(open(os.path.join(root, "sample_class_stats.json")))
    tot = {}
    for s in stats:
        for c, n in s.items():
            if c != "file":
                tot[int(c)] = tot.get(int(c), 0) + n
    order = sorted(tot, key=lambda c: tot[c])
    freq = np.array([tot[c] for c in order], np.float64)
    ref = np.exp((1 - freq / freq.sum()) / 0.01)
    ref /= ref.sum()
    assert classes == order and np.allclose(probs, ref, rtol=1e-4) and classes[0] == 18   # the rarest class leads
    ds = DATASETS.build(_cfg(root, crop=(96, 160)))
    np.random.seed(0)
    got = [ds[0] for _ in range(12)]
    names = {os.path.basename(g["data_samples"].metainfo["seg_map_path"]) for g in got}
    assert "s4_labelTrainIds.png" in names       # class 18 dominates p(c) at T = 0.01 and only s4 holds it
    rare = [g for g in got if g["data_samples"].metainfo["seg_map_path"].endswith("s4_labelTrainIds.png")]
    assert all(int((g["data_samples"].gt_sem_seg.data == 18).sum()) > 200 for g in rare)   # min_pixels * min_crop_ratio, full-size crop


def test_rare_class_sampling_matches_reference_golden(tmp_path, golden_dir):
    """tests/golden/rcs.npz was written by the REFERENCE's own DGDataset / get_rcs_class_probs (rein/datasets/uda_dataset.py:16-37, 44-103,
    imported by path: oracle/gen_golden.py --only rcs) on the toy source of tests/rcs_toy.py.  The sampling is host code whose np.random
    call order IS the contract: class order, probabilities, the per-class file lists, the first 32 draws under np.random.seed(0), every
    source access of the re-draw loop, and the position of the global RNG stream afterwards must all be reproduced exactly."""
    import rcs_toy
    G = np.load(os.path.join(golden_dir, "rcs.npz"))
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "labels"))
    rcs_toy.write_stats(root, rcs_toy.label_maps())
    classes, probs = D.get_rcs_class_probs(root, rcs_toy.RCS["class_temp"])
    assert classes == G["classes"].tolist()
    assert np.array_equal(np.asarray(probs, dtype=np.float64), G["classprob"]), "softmax((1 - freq) / T) in torch fp32, as the reference computes it"
    ds = D.DGDataset(rcs_toy.ToySource(data_root=root), rare_class_sampling=dict(rcs_toy.RCS))
    for c in classes:
        assert [int(f.split("_")[0]) for f in ds.samples_with_class[c]] == G[f"files_{c}"].tolist()
    np.random.seed(0)
    draws, ncalls = [], []
    for _ in range(32):
        n0 = len(ds.source.calls)
        smp = ds[0]
        draws.append((smp["index"],) + tuple(smp["offset"]))
        ncalls.append(len(ds.source.calls) - n0)
    assert np.array_equal(np.array(draws), G["draws"]) and ncalls == G["ncalls"].tolist() and max(ncalls) == 11   # 1 + ten re-draws
    assert np.array_equal(np.array(ds.source.calls), G["calls"])
    assert np.array_equal(np.random.randint(0, 1 << 30, size=4), G["rng_after"])


def test_infinite_sampler_shards_one_stream(tmp_path):
    a = D.InfiniteSampler(7, True, seed=3, rank=0, world=2)
    b = D.InfiniteSampler(7, True, seed=3, rank=1, world=2)
    one = D.InfiniteSampler(7, True, seed=3)
    ia, ib, io = iter(a), iter(b), iter(one)
    merged = [next(io) for _ in range(28)]
    assert [next(ia) for _ in range(14)] == merged[0::2] and [next(ib) for _ in range(14)] == merged[1::2]
    assert sorted(merged[:7]) == list(range(7)) and sorted(merged[7:14]) == list(range(7))
    root = str(tmp_path)
    _tree(root)
    it = D.DataLoaderIter(_cfg(root), batch_size=2, num_workers=0, seed=0)
    batch = next(it)
    assert len(batch["inputs"]) == 2 and batch["inputs"][0].dtype == torch.uint8 and len(batch["data_samples"]) == 2 and it.i == 1
    with pytest.raises(FileNotFoundError):
        DATASETS.build(dict(type="CityscapesDataset", data_root=os.path.join(root, "nope"), data_prefix=dict(img_path="x", seg_map_path="y")))


def test_load_annotations_reduce_zero_label_and_label_map(tmp_path):
    from PIL import Image
    lab = np.array([[0, 1, 2, 255], [3, 0, 254, 7]], np.uint8)
    fn = str(tmp_path / "l.png")
    Image.fromarray(lab).save(fn)
    r = D.LoadAnnotations(reduce_zero_label=True)(dict(seg_map_path=fn, seg_fields=[]))
    # mmseg: 0 -> 255 (ignored), every other id shifts down by one, 255 stays 255 (254 + 1 - 1 ... the 254 -> 255 rule)
    assert r["gt_seg_map"].tolist() == [[255, 0, 1, 255], [2, 255, 253, 6]] and r["seg_fields"] == ["gt_seg_map"]
    r = D.LoadAnnotations()(dict(seg_map_path=fn, seg_fields=[], label_map={7: 0, 3: 255}))
    assert r["gt_seg_map"].tolist() == [[0, 1, 2, 255], [255, 0, 254, 0]]
    packed = D.PackSegInputs()(dict(img=np.zeros((2, 4, 3), np.uint8), gt_seg_map=r["gt_seg_map"], img_path="x.png", ori_shape=(2, 4),
                                    img_shape=(2, 4), flip=False))
    assert tuple(packed["inputs"].shape) == (3, 2, 4) and packed["data_samples"].metainfo["img_path"] == "x.png"
    assert packed["data_samples"].gt_sem_seg.data.dtype == torch.int64


def test_worker_seeds_differ_between_ranks(tmp_path):
    """Round-2 advisor finding: DataLoader workers were seeded from torch.initial_seed() + worker_id, and the main-process torch
    generator is seeded alike on every rank, so worker k drew the SAME rare-class samples, crops and flips on every rank.  mmengine
    seeds worker k of rank r with num_workers * r + k + seed: the ranks' first batches must differ, a rank must reproduce itself."""
    root = str(tmp_path)
    _tree(root, n=12)

    def first_batches(rank, n=3):
        torch.manual_seed(0)            # Runner.from_cfg: identical on every rank
        it = D.DataLoaderIter(_cfg(root), batch_size=2, num_workers=2, shuffle=True, seed=0, rank=rank, world=2)
        out = [next(it) for _ in range(n)]
        del it
        return [np.stack([x.numpy() for x in b["inputs"]]) for b in out]

    r0, r1, r0b = first_batches(0), first_batches(1), first_batches(0)
    assert all(np.array_equal(a, b) for a, b in zip(r0, r0b)), "same rank, same seed: reproducible"
    assert not any(np.array_equal(a, b) for a, b in zip(r0, r1)), "ranks must not train on identical samples"


def test_sampler_fast_forward_continues_the_index_stream(tmp_path):
    s = D.InfiniteSampler(10, True, seed=3, rank=1, world=2)
    it = iter(s)
    head = [next(it) for _ in range(12)]
    s2 = D.InfiniteSampler(10, True, seed=3, rank=1, world=2)
    s2.skip = 8
    it2 = iter(s2)
    assert [next(it2) for _ in range(4)] == head[8:]
    # the skip is spent once: an iterator made later (a re-created DataLoader iterator, persistent workers) starts a stream that is not
    # shifted a second time (round-3 advisor finding)
    assert s2.skip == 0 and [next(iter(s2)) for _ in range(1)] == head[:1]


def test_transforms_match_independent_implementations(golden_dir):
    """Round-2 verdict (row f4): the transform chain had property tests only.  tests/golden/transforms.npz holds a small image / label and
    the outputs of INDEPENDENT implementations (oracle/gen_transform_fixture.py: a numpy bilinear / nearest resize written from OpenCV's
    sampling formula, Python's colorsys for the 8-bit HSV convention, numpy flips) - code that shares nothing with vfmseg_amd.datasets.
    Un-rounded float expectations are stored, so a value may differ only where rounding is a tie."""
    G = np.load(os.path.join(golden_dir, "transforms.npz"))
    img, lab = G["img"], G["lab"]
    for name in ("up", "down", "odd"):
        oh, ow = (int(v) for v in G[f"resize_{name}_size"])
        r = D.Resize(scale=(ow, oh), keep_ratio=False)(dict(img=img.copy(), gt_seg_map=lab.copy(), seg_fields=["gt_seg_map"]))
        want = G[f"resize_{name}_img"]
        got = r["img"].astype(np.float64)
        assert got.shape == want.shape and np.abs(got - want).max() <= 0.5 + 1e-6, (name, np.abs(got - want).max())   # = correctly rounded
        assert np.array_equal(r["gt_seg_map"], G[f"resize_{name}_lab"]), name
        assert r["img_shape"] == (oh, ow)
    hsv = D.bgr2hsv_u8(img).astype(np.float64)
    ref = G["hsv"]
    dh = np.abs(hsv[..., 0] - ref[..., 0] % 180)
    dh = np.minimum(dh, 180 - dh)                     # hue is circular
    assert dh.max() <= 0.5 + 1e-6 and np.abs(hsv[..., 1:] - ref[..., 1:]).max() <= 0.5 + 1e-6
    back = D.hsv2bgr_u8(G["hsv_u8"]).astype(np.float64)
    assert np.abs(back - G["bgr_back"]).max() <= 0.5 + 1e-6
    for direction, key in (("horizontal", "flip_h"), ("vertical", "flip_v")):
        r = D.RandomFlip(prob=1.0, direction=direction)(dict(img=img.copy(), gt_seg_map=lab.copy(), seg_fields=["gt_seg_map"]))
        assert np.array_equal(r["img"], G[key + "_img"]) and np.array_equal(r["gt_seg_map"], G[key + "_lab"]) and r["flip"] is True
