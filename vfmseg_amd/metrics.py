"""mIoU evaluation (mmseg IoUMetric / rein/dg_metrics.py:24-102 DGIoUMetric without the per-dataset grouping):
per-class intersection / union from a confusion histogram accumulated on the GPU."""
import numpy as np
import torch

from .registry import METRICS


@METRICS.register_module()
class IoUMetric:
    def __init__(self, num_classes=19, ignore_index=255, **kw):
        self.nc, self.ignore = num_classes, ignore_index
        self.reset()

    def reset(self):
        self.inter = torch.zeros(self.nc, dtype=torch.float64)
        self.pred = torch.zeros(self.nc, dtype=torch.float64)
        self.lab = torch.zeros(self.nc, dtype=torch.float64)

    def process(self, pred, label):
        """pred uint8/int [H,W], label int64 [H,W] (same device)."""
        pred, label = pred.reshape(-1).long(), label.reshape(-1).long()
        valid = label != self.ignore
        p, l = pred[valid], label[valid]
        self.inter += torch.bincount(p[p == l], minlength=self.nc).double().cpu()
        self.pred += torch.bincount(p, minlength=self.nc).double().cpu()
        self.lab += torch.bincount(l, minlength=self.nc).double().cpu()

    def compute(self):
        union = self.pred + self.lab - self.inter
        iou = (self.inter / union).numpy()
        acc = (self.inter / self.lab).numpy()
        return dict(aAcc=float(100 * self.inter.sum() / self.lab.sum()), mIoU=float(np.round(np.nanmean(iou) * 100, 2)),
                    mAcc=float(np.round(np.nanmean(acc) * 100, 2)))


DGIoUMetric = IoUMetric
METRICS.register_module(name="DGIoUMetric", module=IoUMetric, force=True)
