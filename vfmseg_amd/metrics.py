"""mIoU evaluation on the HIP path: mmseg IoUMetric (1.2.2 semantics, restated: SURVEY App. D) and the reference's
DGIoUMetric (rein/dg_metrics.py:24-102: results grouped per dataset by a substring of `seg_map_path`, `<dataset>_<metric>`
keys plus `mean_<metric>` over `mean_used_keys`).

Per sample one launch of vfm_confusion_hist accumulates the (label, prediction) confusion counts into a device int64 matrix;
the three areas mmseg's intersect_and_union returns are its diagonal, column sums and row sums.  The host only sees
(num_classes+1) x num_classes integers per sample group, once, in compute_metrics()."""
from collections import OrderedDict, defaultdict

import numpy as np
import torch

from . import ops
from .registry import METRICS

CITYSCAPES_CLASSES = ("road", "sidewalk", "building", "wall", "fence", "pole", "traffic light", "traffic sign", "vegetation",
                      "terrain", "sky", "person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle")


def _field(sample, name):
    """`sample[name]['data']` for the dict form mmengine evaluators pass, `.name.data` for SegDataSample objects."""
    v = sample[name] if isinstance(sample, dict) else getattr(sample, name)
    if isinstance(v, dict):
        return v["data"]
    return v.data if hasattr(v, "data") and not torch.is_tensor(v) else v


def _meta(sample, name, default=None):
    if isinstance(sample, dict):
        return sample.get(name, default)
    m = getattr(sample, "metainfo", None) or {}
    return m.get(name, getattr(sample, name, default))


def confusion(pred, label, num_classes, ignore_index=255, out=None):
    """int64 [(nc+1), nc] confusion counts (rows: label, last row = labels outside [0, nc) that are not ignore_index; columns:
    prediction) accumulated into `out` on the GPU."""
    dev = pred.device
    p = pred.reshape(-1)
    if p.dtype != torch.uint8:
        p = p.to(torch.uint8)
    lab = label.reshape(-1).to(dev)
    if lab.dtype not in (torch.uint8, torch.int64):
        lab = lab.long()
    if out is None:
        out = torch.zeros((num_classes + 1) * num_classes, dtype=torch.int64, device=dev)
    ops.confusion_hist(p.contiguous(), lab.contiguous(), out, num_classes, ignore_index)
    return out


def areas_from_confusion(cm, num_classes):
    """(area_intersect, area_union, area_pred_label, area_label) of mmseg IoUMetric.intersect_and_union from the confusion
    counts of one sample (or a sum of samples)."""
    cm = np.asarray(cm, dtype=np.float64).reshape(num_classes + 1, num_classes)
    inter = np.diag(cm[:num_classes]).copy()
    pred = cm.sum(0)
    lab = cm[:num_classes].sum(1)
    return inter, pred + lab - inter, pred, lab


def total_area_to_metrics(inter, union, pred, lab, metrics=("mIoU",), nan_to_num=None, beta=1):
    """mmseg IoUMetric.total_area_to_metrics."""
    ret = OrderedDict(aAcc=inter.sum() / lab.sum())
    with np.errstate(divide="ignore", invalid="ignore"):
        for m in metrics:
            if m == "mIoU":
                ret["IoU"], ret["Acc"] = inter / union, inter / lab
            elif m == "mDice":
                ret["Dice"], ret["Acc"] = 2 * inter / (pred + lab), inter / lab
            elif m == "mFscore":
                precision, recall = inter / pred, inter / lab
                ret["Fscore"] = (1 + beta ** 2) * precision * recall / (beta ** 2 * precision + recall)
                ret["Precision"], ret["Recall"] = precision, recall
            else:
                raise KeyError(f"metrics {m} is not supported")
    if nan_to_num is not None:
        ret = OrderedDict((k, np.nan_to_num(v, nan=nan_to_num)) for k, v in ret.items())
    return ret


@METRICS.register_module()
class IoUMetric:
    """mmseg.evaluation.IoUMetric surface: process(data_batch, data_samples) / compute_metrics(results) / evaluate(size),
    `dataset_meta['classes']`, `results`."""

    def __init__(self, ignore_index=255, iou_metrics=("mIoU",), nan_to_num=None, beta=1, collect_device="cpu", output_dir=None,
                 format_only=False, prefix=None, num_classes=None, **kw):
        self.ignore_index, self.metrics, self.nan_to_num, self.beta = ignore_index, list(iou_metrics), nan_to_num, beta
        self.output_dir, self.format_only, self.prefix = output_dir, format_only, prefix
        self.dataset_meta = dict(classes=CITYSCAPES_CLASSES[:num_classes] if num_classes else CITYSCAPES_CLASSES)
        self.results = []

    @property
    def num_classes(self):
        return len(self.dataset_meta["classes"])

    def _sample_confusion(self, sample):
        pred = _field(sample, "pred_sem_seg").squeeze()
        label = _field(sample, "gt_sem_seg").squeeze()
        return confusion(pred, label, self.num_classes, self.ignore_index)

    def process(self, data_batch, data_samples):
        for s in data_samples:
            if not self.format_only:
                self.results.append(self._sample_confusion(s))

    def _summarise(self, cms):
        nc = self.num_classes
        total = torch.stack(list(cms)).sum(0).cpu().numpy() if cms else np.zeros((nc + 1) * nc)
        ret = total_area_to_metrics(*areas_from_confusion(total, nc), self.metrics, self.nan_to_num, self.beta)
        out = OrderedDict()
        for k, v in ret.items():
            val = float(np.round(np.nanmean(v) * 100, 2))
            out[k if k == "aAcc" else "m" + k] = val
        self.last_per_class = OrderedDict((k, np.round(np.asarray(v) * 100, 2)) for k, v in ret.items() if k != "aAcc")
        return out

    def compute_metrics(self, results):
        if self.format_only:
            return OrderedDict()
        return self._summarise(results)

    def evaluate(self, size=None):
        m = self.compute_metrics(self.results)
        self.results = []
        return {f"{self.prefix}/{k}": v for k, v in m.items()} if self.prefix else m


@METRICS.register_module()
class DGIoUMetric(IoUMetric):
    """rein/dg_metrics.py:24-102: each sample is filed under the first `dataset_keys` entry contained in the batch's
    FIRST sample's `seg_map_path` (:53-58 - the reference indexes data_samples[0], reproduced), else "unknown"; metrics are
    computed per dataset and averaged over `mean_used_keys` (default: all dataset_keys) into `mean_<metric>`."""

    def __init__(self, dataset_keys=(), mean_used_keys=(), **kw):
        super().__init__(**kw)
        self.dataset_keys = list(dataset_keys)
        self.mean_used_keys = list(mean_used_keys) if mean_used_keys else list(dataset_keys)

    def process(self, data_batch, data_samples):
        for s in data_samples:
            if self.format_only:
                continue
            key = "unknown"
            path0 = _meta(data_samples[0], "seg_map_path", "") or ""
            for k in self.dataset_keys:
                if k in path0:
                    key = k
                    break
            self.results.append([key, self._sample_confusion(s)])

    def compute_metrics(self, results):
        per = defaultdict(list)
        for key, cm in results:
            per[key].append(cm)
        metrics, to_mean = OrderedDict(), defaultdict(list)
        self.samples_per_dataset = {k: len(v) for k, v in per.items()}
        for key, cms in per.items():
            for k, v in self._summarise(cms).items():
                metrics[f"{key}_{k}"] = v
                if key in self.mean_used_keys:
                    to_mean[k].append(v)
        for k, v in to_mean.items():
            metrics[f"mean_{k}"] = sum(v) / len(v)
        return metrics
