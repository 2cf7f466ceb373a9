"""Input pipeline of the reference's configs: `CityscapesDataset` folders, the mmseg / mmcv transform chain
of `configs/_base_/datasets/*.py` (LoadImageFromFile, LoadAnnotations, Resize, RandomCrop, RandomFlip, PhotoMetricDistortion,
PackSegInputs), the rare-class-sampling wrapper `DGDataset` (rein/datasets/uda_dataset.py:15-107) and an infinite, rank-sharded
loader (mmengine InfiniteSampler + pseudo_collate).  Host-side numpy / PIL code: it feeds `SegDataPreProcessor`'s uint8 path
(one HIP kernel per sample for channel swap + normalise + pad), nothing here touches the GPU.

The transform semantics are third-party (mmsegmentation 1.2.2 / mmcv 2.1.0, not under /root/reference): restated from their
published behaviour, with two deliberate, documented deviations forced by this image (no OpenCV): images are decoded with PIL
and flipped to BGR (mmcv's cv2 backend yields BGR, and the configs' `bgr_to_rgb=True` expects that), and bilinear resizing is
`F.interpolate(align_corners=False, antialias=False)` - cv2.INTER_LINEAR's sampling positions, up to its 11-bit fixed-point
rounding (+-1 grey level)."""
import json
import os
import os.path as osp

import numpy as np
import torch

from .registry import DATASETS, Registry
from .segmentors import SegDataSample

TRANSFORMS = Registry("transform")

CITYSCAPES_CLASSES = ("road", "sidewalk", "building", "wall", "fence", "pole", "traffic light", "traffic sign", "vegetation", "terrain",
                      "sky", "person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle")
CITYSCAPES_PALETTE = [[128, 64, 128], [244, 35, 232], [70, 70, 70], [102, 102, 156], [190, 153, 153], [153, 153, 153], [250, 170, 30],
                      [220, 220, 0], [107, 142, 35], [152, 251, 152], [70, 130, 180], [220, 20, 60], [255, 0, 0], [0, 0, 142], [0, 0, 70],
                      [0, 60, 100], [0, 80, 100], [0, 0, 230], [119, 11, 32]]


# ------------------------------------------------------------------------------------------------ colour space (OpenCV 8-bit convention)
def bgr2hsv_u8(img):
    """cv2.cvtColor(img, COLOR_BGR2HSV) for uint8: H in [0, 180) (degrees / 2), S and V in [0, 255]."""
    f = img.astype(np.float32)
    b, g, r = f[..., 0], f[..., 1], f[..., 2]
    v = np.maximum(np.maximum(r, g), b)
    mn = np.minimum(np.minimum(r, g), b)
    diff = v - mn
    s = np.where(v > 0, diff / np.maximum(v, 1e-12) * 255.0, 0.0)
    safe = np.where(diff > 0, diff, 1.0)
    h = np.where(v == r, 60.0 * (g - b) / safe, np.where(v == g, 120.0 + 60.0 * (b - r) / safe, 240.0 + 60.0 * (r - g) / safe))
    h = np.where(diff > 0, h, 0.0)
    h = np.where(h < 0, h + 360.0, h) / 2.0
    out = np.stack([np.rint(h) % 180, np.rint(s), v], -1)
    return np.clip(out, 0, 255).astype(np.uint8)


def hsv2bgr_u8(img):
    """cv2.cvtColor(img, COLOR_HSV2BGR) for uint8 (H in [0, 180))."""
    f = img.astype(np.float32)
    h, s, v = f[..., 0] * 2.0, f[..., 1] / 255.0, f[..., 2]
    hi = np.floor(h / 60.0)
    fr = h / 60.0 - hi
    hi = hi.astype(np.int32) % 6
    p, q, t = v * (1 - s), v * (1 - s * fr), v * (1 - s * (1 - fr))
    r = np.choose(hi, [v, q, p, p, t, v])
    g = np.choose(hi, [t, v, v, q, p, p])
    b = np.choose(hi, [p, p, t, v, v, q])
    return np.clip(np.rint(np.stack([b, g, r], -1)), 0, 255).astype(np.uint8)


def _resize(arr, size_hw, mode):
    """arr HxW or HxWxC uint8 -> size_hw; bilinear at cv2.INTER_LINEAR's sampling positions, or nearest (cv2.INTER_NEAREST:
    floor(dst * scale), which is what F.interpolate(mode='nearest') computes)."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    t = t[None, None] if t.dim() == 2 else t.permute(2, 0, 1)[None]
    if mode == "nearest":
        o = torch.nn.functional.interpolate(t.float(), size=size_hw, mode="nearest")
    else:
        o = torch.nn.functional.interpolate(t.float(), size=size_hw, mode="bilinear", align_corners=False).round_().clamp_(0, 255)
    o = o[0].to(torch.uint8)
    return (o[0] if arr.ndim == 2 else o.permute(1, 2, 0)).contiguous().numpy()


# ------------------------------------------------------------------------------------------------ transforms
@TRANSFORMS.register_module()
class LoadImageFromFile:
    """mmcv LoadImageFromFile: decoded image HxWx3 uint8 in BGR order (cv2 backend), img_shape / ori_shape."""

    def __init__(self, to_float32=False, color_type="color", **_):
        self.to_float32 = to_float32

    def __call__(self, results):
        from PIL import Image
        with Image.open(results["img_path"]) as im:
            img = np.asarray(im.convert("RGB"))[..., ::-1]
        img = np.ascontiguousarray(img)
        if self.to_float32:
            img = img.astype(np.float32)
        results["img"] = img
        results["img_shape"] = img.shape[:2]
        results["ori_shape"] = img.shape[:2]
        return results


@TRANSFORMS.register_module()
class LoadAnnotations:
    """mmseg LoadAnnotations: label map HxW uint8 read unchanged; reduce_zero_label / label_map as in BaseSegDataset."""

    def __init__(self, reduce_zero_label=None, **_):
        self.reduce_zero_label = reduce_zero_label

    def __call__(self, results):
        from PIL import Image
        with Image.open(results["seg_map_path"]) as im:
            seg = np.asarray(im).astype(np.uint8)
        if seg.ndim == 3:
            seg = seg[..., 0]
        seg = np.ascontiguousarray(seg)
        rz = self.reduce_zero_label if self.reduce_zero_label is not None else results.get("reduce_zero_label", False)
        if rz:
            seg = seg.copy()
            seg[seg == 0] = 255
            seg = seg - 1
            seg[seg == 254] = 255
        lm = results.get("label_map")
        if lm:
            cp = seg.copy()
            for old, new in lm.items():
                seg[cp == old] = new
        results["gt_seg_map"] = seg
        results.setdefault("seg_fields", []).append("gt_seg_map")
        return results


@TRANSFORMS.register_module()
class Resize:
    """mmcv Resize: scale = (w, h); keep_ratio rescales so that the image fits inside the scale box (mmcv.rescale_size).
    Image bilinear, label maps nearest."""

    def __init__(self, scale=None, scale_factor=None, keep_ratio=False, **_):
        self.scale, self.scale_factor, self.keep_ratio = scale, scale_factor, keep_ratio

    def __call__(self, results):
        h, w = results["img"].shape[:2]
        if self.scale is not None:
            sw, sh = self.scale
        else:
            f = self.scale_factor if isinstance(self.scale_factor, (tuple, list)) else (self.scale_factor, self.scale_factor)
            sw, sh = int(w * f[0] + 0.5), int(h * f[1] + 0.5)
        if self.keep_ratio:
            k = min(max(sw, sh) / max(h, w), min(sw, sh) / min(h, w))
            nw, nh = int(w * float(k) + 0.5), int(h * float(k) + 0.5)
        else:
            nw, nh = int(sw), int(sh)
        results["img"] = _resize(results["img"], (nh, nw), "bilinear")
        results["img_shape"] = (nh, nw)
        results["scale_factor"] = (nw / w, nh / h)
        results["keep_ratio"] = self.keep_ratio
        for key in results.get("seg_fields", []):
            results[key] = _resize(results[key], (nh, nw), "nearest")
        return results


@TRANSFORMS.register_module()
class RandomCrop:
    """mmseg RandomCrop: a random crop_size window; with cat_max_ratio < 1 up to ten draws until no single category (ignore_index
    excluded) covers more than that share of the window and at least two categories are present."""

    def __init__(self, crop_size, cat_max_ratio=1.0, ignore_index=255):
        self.crop_size = (crop_size, crop_size) if isinstance(crop_size, int) else tuple(crop_size)
        self.cat_max_ratio, self.ignore_index = cat_max_ratio, ignore_index

    def _bbox(self, img):
        mh, mw = max(img.shape[0] - self.crop_size[0], 0), max(img.shape[1] - self.crop_size[1], 0)
        oh, ow = np.random.randint(0, mh + 1), np.random.randint(0, mw + 1)
        return oh, oh + self.crop_size[0], ow, ow + self.crop_size[1]

    def __call__(self, results):
        img = results["img"]
        box = self._bbox(img)
        if self.cat_max_ratio < 1.0 and "gt_seg_map" in results:
            for _ in range(10):
                seg = results["gt_seg_map"][box[0]:box[1], box[2]:box[3]]
                labels, cnt = np.unique(seg, return_counts=True)
                cnt = cnt[labels != self.ignore_index]
                if len(cnt) > 1 and np.max(cnt) / np.sum(cnt) < self.cat_max_ratio:
                    break
                box = self._bbox(img)
        results["img"] = np.ascontiguousarray(img[box[0]:box[1], box[2]:box[3]])
        results["img_shape"] = results["img"].shape[:2]
        for key in results.get("seg_fields", []):
            results[key] = np.ascontiguousarray(results[key][box[0]:box[1], box[2]:box[3]])
        return results


@TRANSFORMS.register_module()
class RandomFlip:
    """mmcv RandomFlip (prob, direction='horizontal'): image and label maps flipped together."""

    def __init__(self, prob=None, direction="horizontal", **_):
        self.prob, self.direction = prob, direction

    def __call__(self, results):
        flip = self.prob is not None and np.random.rand() < self.prob
        results["flip"] = bool(flip)
        results["flip_direction"] = self.direction if flip else None
        if flip:
            ax = 1 if self.direction == "horizontal" else 0
            results["img"] = np.ascontiguousarray(np.flip(results["img"], ax))
            for key in results.get("seg_fields", []):
                results[key] = np.ascontiguousarray(np.flip(results[key], ax))
        return results


@TRANSFORMS.register_module()
class PhotoMetricDistortion:
    """mmseg PhotoMetricDistortion on the BGR uint8 image: random brightness (+-32), contrast (x0.5..1.5, before or after the
    colour ops with equal chance), saturation (x0.5..1.5 in HSV) and hue (+-18 of 180), each applied with probability 1/2."""

    def __init__(self, brightness_delta=32, contrast_range=(0.5, 1.5), saturation_range=(0.5, 1.5), hue_delta=18):
        self.bd, self.cl, self.cu = brightness_delta, contrast_range[0], contrast_range[1]
        self.sl, self.su, self.hd = saturation_range[0], saturation_range[1], hue_delta

    @staticmethod
    def _convert(img, alpha=1.0, beta=0.0):
        return np.clip(img.astype(np.float32) * alpha + beta, 0, 255).astype(np.uint8)

    def __call__(self, results):
        img = results["img"]
        if np.random.randint(2):
            img = self._convert(img, beta=np.random.uniform(-self.bd, self.bd))
        mode = np.random.randint(2)
        if mode == 1 and np.random.randint(2):
            img = self._convert(img, alpha=np.random.uniform(self.cl, self.cu))
        if np.random.randint(2):
            hsv = bgr2hsv_u8(img)
            hsv[..., 1] = self._convert(hsv[..., 1], alpha=np.random.uniform(self.sl, self.su))
            img = hsv2bgr_u8(hsv)
        if np.random.randint(2):
            hsv = bgr2hsv_u8(img)
            hsv[..., 0] = (hsv[..., 0].astype(np.int32) + np.random.randint(-self.hd, self.hd)) % 180
            img = hsv2bgr_u8(hsv)
        if mode == 0 and np.random.randint(2):
            img = self._convert(img, alpha=np.random.uniform(self.cl, self.cu))
        results["img"] = img
        return results


@TRANSFORMS.register_module()
class PackSegInputs:
    """mmseg PackSegInputs: inputs = uint8 CHW tensor, data_samples = SegDataSample(gt_sem_seg [1, H, W] int64, metainfo)."""

    META = ("img_path", "seg_map_path", "ori_shape", "img_shape", "pad_shape", "scale_factor", "flip", "flip_direction", "reduce_zero_label")

    def __init__(self, meta_keys=None):
        self.meta_keys = tuple(meta_keys) if meta_keys is not None else self.META

    def __call__(self, results):
        img = results["img"]
        inputs = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))
        gt = None
        if "gt_seg_map" in results:
            gt = torch.from_numpy(results["gt_seg_map"][None].astype(np.int64))
        meta = {k: results[k] for k in self.meta_keys if k in results}
        return dict(inputs=inputs, data_samples=SegDataSample(gt_sem_seg=gt, metainfo=meta))


class Compose:
    def __init__(self, pipeline):
        self.transforms = [TRANSFORMS.build(t) if isinstance(t, dict) else t for t in (pipeline or [])]

    def __call__(self, results):
        for t in self.transforms:
            results = t(results)
            if results is None:
                return None
        return results


# ------------------------------------------------------------------------------------------------ datasets
@DATASETS.register_module()
class BaseSegDataset:
    """mmseg BaseSegDataset, folder form: every file under data_root/data_prefix.img_path (recursively) ending in img_suffix is a
    sample; its label map is the same relative name with seg_map_suffix under data_prefix.seg_map_path."""

    METAINFO = dict(classes=(), palette=[])

    def __init__(self, data_root="", data_prefix=None, img_suffix=".jpg", seg_map_suffix=".png", pipeline=None, ignore_index=255,
                 reduce_zero_label=False, test_mode=False, serialize_data=True, lazy_init=False, metainfo=None, **_):
        self.data_root, self.ignore_index, self.reduce_zero_label, self.test_mode = data_root, ignore_index, reduce_zero_label, test_mode
        self.data_prefix = dict(data_prefix or dict(img_path="", seg_map_path=""))
        self.img_suffix, self.seg_map_suffix = img_suffix, seg_map_suffix
        self.metainfo = dict(self.METAINFO, **(metainfo or {}))
        self.pipeline = Compose(pipeline)
        self.data_list = self.load_data_list()

    def load_data_list(self):
        img_dir = osp.join(self.data_root, self.data_prefix.get("img_path", ""))
        ann_dir = self.data_prefix.get("seg_map_path")
        ann_dir = osp.join(self.data_root, ann_dir) if ann_dir is not None else None
        if not osp.isdir(img_dir):
            raise FileNotFoundError(f"{type(self).__name__}: image directory {img_dir!r} does not exist")
        out = []
        for root, _, files in sorted(os.walk(img_dir)):
            for fn in sorted(files):
                if not fn.endswith(self.img_suffix):
                    continue
                rel = osp.relpath(osp.join(root, fn), img_dir)
                item = dict(img_path=osp.join(img_dir, rel), label_map=None, reduce_zero_label=self.reduce_zero_label, seg_fields=[])
                if ann_dir is not None:
                    item["seg_map_path"] = osp.join(ann_dir, rel[:-len(self.img_suffix)] + self.seg_map_suffix)
                out.append(item)
        return out

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, idx):
        item = dict(self.data_list[idx])
        item["seg_fields"] = []
        item["sample_idx"] = idx
        return self.pipeline(item)


@DATASETS.register_module()
class CityscapesDataset(BaseSegDataset):
    """mmseg CityscapesDataset: the 19 train-id classes; default suffixes of the leftImg8bit / gtFine trees."""

    METAINFO = dict(classes=CITYSCAPES_CLASSES, palette=CITYSCAPES_PALETTE)

    def __init__(self, img_suffix="_leftImg8bit.png", seg_map_suffix="_gtFine_labelTrainIds.png", **kw):
        super().__init__(img_suffix=img_suffix, seg_map_suffix=seg_map_suffix, **kw)


def get_rcs_class_probs(data_root, temperature):
    """rein/datasets/uda_dataset.py:15-37: class sampling probabilities softmax((1 - freq) / T) over the classes of
    sample_class_stats.json, classes ordered by ascending pixel count."""
    with open(osp.join(data_root, "sample_class_stats.json")) as f:
        stats = json.load(f)
    overall = {}
    for s in stats:
        for c, n in s.items():
            if c == "file":
                continue
            overall[int(c)] = overall.get(int(c), 0) + n
    overall = dict(sorted(overall.items(), key=lambda kv: kv[1]))
    freq = torch.tensor(list(overall.values()), dtype=torch.float32)
    freq = 1 - freq / freq.sum()
    return list(overall.keys()), torch.softmax(freq / temperature, dim=-1).numpy()


@DATASETS.register_module()
class DGDataset:
    """rein/datasets/uda_dataset.py:40-107: the source dataset, optionally drawn by rare class sampling - a class c ~ p(c), a
    file known to hold more than min_pixels of c, and up to ten re-draws of the random crop until the crop holds more than
    min_pixels * min_crop_ratio pixels of c."""

    def __init__(self, source, **cfg):
        self.source = DATASETS.build(source) if isinstance(source, dict) else source
        self.ignore_index = self.source.ignore_index
        self.CLASSES, self.PALETTE = self.source.metainfo["classes"], self.source.metainfo["palette"]
        rcs = cfg.get("rare_class_sampling")
        self.rcs_enabled = rcs is not None
        if self.rcs_enabled:
            self.rcs_class_temp, self.rcs_min_crop_ratio, self.rcs_min_pixels = rcs["class_temp"], rcs["min_crop_ratio"], rcs["min_pixels"]
            root = source["data_root"] if isinstance(source, dict) else self.source.data_root
            self.rcs_classes, self.rcs_classprob = get_rcs_class_probs(root, self.rcs_class_temp)
            with open(osp.join(root, "samples_with_class.json")) as f:
                swc = json.load(f)
            swc = {int(k): v for k, v in swc.items() if int(k) in self.rcs_classes}
            self.samples_with_class = {}
            for c in self.rcs_classes:
                self.samples_with_class[c] = [file.split("/")[-1] for file, pixels in swc[c] if pixels > self.rcs_min_pixels]
                assert len(self.samples_with_class[c]) > 0, f"no sample holds more than {self.rcs_min_pixels} pixels of class {c}"
            self.file_to_idx = {item["seg_map_path"].split("/")[-1]: i for i, item in enumerate(self.source.data_list)}

    def get_rare_class_sample(self):
        c = np.random.choice(self.rcs_classes, p=self.rcs_classprob)
        f1 = np.random.choice(self.samples_with_class[c])
        i1 = self.file_to_idx[f1]
        s1 = self.source[i1]
        if self.rcs_min_crop_ratio > 0:
            for _ in range(10):
                if int((s1["data_samples"].gt_sem_seg.data == c).sum()) > self.rcs_min_pixels * self.rcs_min_crop_ratio:
                    break
                s1 = self.source[i1]
        return s1

    def __getitem__(self, idx):
        return self.get_rare_class_sample() if self.rcs_enabled else self.source[idx]

    def __len__(self):
        return len(self.source)


# ------------------------------------------------------------------------------------------------ sampling / loading
class InfiniteSampler:
    """mmengine InfiniteSampler: an endless stream of shuffled permutations of range(size) from ONE generator seeded alike on
    every rank; rank r takes every world-th index of it."""

    def __init__(self, size, shuffle=True, seed=0, rank=0, world=1):
        self.size, self.shuffle, self.seed, self.rank, self.world = size, shuffle, seed, rank, world
        self.skip = 0

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed)
        k = 0
        # resume: this rank's first `skip` indices were consumed before the checkpoint.  The skip is spent by the FIRST iterator made
        # after fast_forward (DataLoaderIter re-creates its iterator; persistent workers iterate again): a later iterator continues
        # a stream that is already in step with the uninterrupted run and must not skip again.
        skip, self.skip = self.skip, 0
        while True:
            idx = torch.randperm(self.size, generator=g).tolist() if self.shuffle else list(range(self.size))
            for i in idx:
                if k % self.world == self.rank:
                    if skip > 0:
                        skip -= 1
                    else:
                        yield i
                k += 1


def pseudo_collate(batch):
    """mmengine pseudo_collate for this sample layout: lists, no stacking (SegDataPreProcessor pads and stacks on the GPU)."""
    return dict(inputs=[b["inputs"] for b in batch], data_samples=[b["data_samples"] for b in batch])


class DataLoaderIter:
    """Iterator of training batches dict(inputs=[uint8 CHW] * B, data_samples=[SegDataSample] * B) from a dataset config
    (`train_dataloader` of the reference's configs).  num_workers > 0 uses torch's DataLoader worker processes; every worker
    is seeded with num_workers * rank + worker_id + seed (mmengine.dataset.worker_init_fn): the DataLoader's own base seed comes from
    the main process's torch generator, which Runner.from_cfg seeds identically on every rank, so it must not be the only source -
    with rare class sampling the dataset ignores idx and draws class, file and crop from np.random, and equal worker seeds would make
    every rank train on the same samples."""

    def __init__(self, dataset, batch_size=2, num_workers=0, shuffle=True, seed=0, rank=0, world=1, infinite=True):
        self.dataset = DATASETS.build(dataset) if isinstance(dataset, dict) else dataset
        self.bs, self.i = batch_size, 0
        self.sampler = InfiniteSampler(len(self.dataset), shuffle, seed, rank, world) if infinite else range(rank, len(self.dataset), world)
        self.seed, self.rank, self.num_workers, self.epoch_offset = seed, rank, num_workers, 0
        loader_self = self

        def init_fn(worker_id):
            import random
            ws = (loader_self.num_workers * loader_self.rank + worker_id + loader_self.seed + 7919 * loader_self.epoch_offset) % 2 ** 32
            np.random.seed(ws)
            random.seed(ws)
            torch.manual_seed(ws)

        self.loader = torch.utils.data.DataLoader(self.dataset, batch_size=batch_size, sampler=self.sampler, num_workers=num_workers,
                                                  collate_fn=pseudo_collate, worker_init_fn=init_fn, drop_last=infinite,
                                                  persistent_workers=num_workers > 0)
        self._it = None

    def __iter__(self):
        self._it = iter(self.loader)
        return self

    def __next__(self):
        if self._it is None:
            self._it = iter(self.loader)
        self.i += 1
        return next(self._it)

    def fast_forward(self, batches):
        """Resume: continue the sample order where the checkpoint stopped (this rank consumed `batches` batches) and give the
        workers seeds they have not used yet."""
        self.i = batches
        if isinstance(self.sampler, InfiniteSampler):
            self.sampler.skip = batches * self.bs
        self.epoch_offset = batches
        self._it = None

    next = __next__
