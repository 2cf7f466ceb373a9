"""Toy source dataset for the rare-class-sampling fixture (tests/golden/rcs.npz): shared by the generator (oracle/gen_golden.py --only rcs,
which drives the REFERENCE's DGDataset with it) and by tests/test_datasets_cpu.py (which drives vfmseg_amd.datasets.DGDataset with it).
Twelve 16 x 16 label maps over six classes; __getitem__ draws a random 8 x 8 crop with np.random (as the real pipeline's RandomCrop
does), so that the ten re-draws of get_rare_class_sample consume the global RNG stream exactly where the reference consumes it."""
import json
import os
import types

import numpy as np
import torch

N_FILES, N_CLASSES, SIZE, CROP = 12, 6, 16, 8
RCS = dict(class_temp=0.01, min_crop_ratio=0.5, min_pixels=20)


def label_maps():
    g = np.random.RandomState(1234)
    maps = []
    for i in range(N_FILES):
        m = g.randint(0, 3, size=(SIZE, SIZE))                     # common classes 0..2 everywhere
        if i % 2 == 0:                                            # rarer classes in blobs, so that crops often miss them
            m[:6, :6] = 3
        if i % 3 == 0:
            m[10:, 9:] = 4
        if i % 4 == 1:
            m[2:8, 10:15] = 5
        maps.append(m.astype(np.uint8))
    return maps


def write_stats(root, maps):
    """sample_class_stats.json / samples_with_class.json in the format the reference's converters write (tools/convert_datasets/gta.py)."""
    stats, swc = [], {}
    for i, m in enumerate(maps):
        name = f"labels/{i:05d}_labelTrainIds.png"
        row = {"file": name}
        for c in range(N_CLASSES):
            n = int((m == c).sum())
            if n > 0:
                row[str(c)] = n
                swc.setdefault(str(c), []).append([name, n])
        stats.append(row)
    with open(os.path.join(root, "sample_class_stats.json"), "w") as f:
        json.dump(stats, f)
    with open(os.path.join(root, "samples_with_class.json"), "w") as f:
        json.dump(swc, f)


class ToySource:
    """What DGDataset needs of its source: ignore_index, METAINFO / metainfo, data_list, __getitem__ (random crop), __len__."""
    ignore_index = 255
    METAINFO = metainfo = dict(classes=tuple(f"c{i}" for i in range(N_CLASSES)), palette=[[i, i, i] for i in range(N_CLASSES)])

    def __init__(self, data_root=None, **kw):
        self.data_root = data_root
        self.maps = label_maps()
        self.data_list = [dict(seg_map_path=f"{data_root}/labels/{i:05d}_labelTrainIds.png") for i in range(N_FILES)]
        self.calls = []

    def __len__(self):
        return N_FILES

    def __getitem__(self, i):
        oy, ox = (int(v) for v in np.random.randint(0, SIZE - CROP + 1, size=2))
        self.calls.append((int(i), oy, ox))
        crop = torch.from_numpy(self.maps[i][oy:oy + CROP, ox:ox + CROP].astype(np.int64))
        return dict(inputs=None, data_samples=types.SimpleNamespace(gt_sem_seg=types.SimpleNamespace(data=crop)), index=int(i), offset=(oy, ox))
