"""The thin slice of mmengine.Runner the hot path needs (tools/train.py:64-121, tools/test.py:96-145): build model /
optimiser from a config, iteration-based train loop with LoggerHook-style scalars, checkpoint save / resume, DP."""
import json
import os
import time

import numpy as np
import torch

from . import parallel
from .optim import PEFTOptimWrapperConstructor
from .registry import MODELS, OPTIM_WRAPPER_CONSTRUCTORS
from .segmentors import SegDataSample
from .synth import synth_image, synth_label, synth_like


class SyntheticLoader:
    """Infinite stream of synthetic 19-class samples (no datasets offline); rank r takes every world-th index
    (mmengine InfiniteSampler semantics)."""

    def __init__(self, batch_size, size, rank=0, world=1, seed=0, pool=8):
        self.bs, self.size, self.rank, self.world, self.seed, self.pool = batch_size, size, rank, world, seed, pool
        self.i = 0

    def __iter__(self):
        return self

    def __next__(self):
        idx = [(self.i + k) * self.world + self.rank for k in range(self.bs)]
        self.i += self.bs
        imgs = torch.cat([synth_image(1, self.size, seed=self.seed + j % self.pool) for j in idx])
        labs = torch.cat([synth_label(1, self.size, seed=self.seed + j % self.pool) for j in idx])
        return dict(inputs=imgs, data_samples=[SegDataSample(gt_sem_seg=labs[k]) for k in range(self.bs)])


class Runner:
    def __init__(self, cfg, model, optim_wrapper, loader, work_dir, rank=0, world=1):
        self.cfg, self.model, self.ow, self.loader, self.work_dir = cfg, model, optim_wrapper, loader, work_dir
        self.rank, self.world = rank, world
        self.iter = 0

    @classmethod
    def from_cfg(cfg_cls, cfg, synthetic=True, device=None):
        rank, world, local = parallel.init_from_env()
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            from . import lib
            lib.set_device_index(local)
        seed = cfg.get("randomness", {}).get("seed", 0)
        np.random.seed(seed + rank)
        torch.manual_seed(seed)
        model = MODELS.build(cfg["model"])
        if cfg.get("synthetic_init", True):
            model.load_state_dict(synth_like(model.state_dict()))
        model = model.cuda().train()
        ow_cfg = dict(cfg.get("optim_wrapper", {}))
        ctor = OPTIM_WRAPPER_CONSTRUCTORS.get(ow_cfg.get("constructor", "PEFTOptimWrapperConstructor")) or PEFTOptimWrapperConstructor
        ow = ctor(ow_cfg, ow_cfg.get("paramwise_cfg"))(model, cfg.get("param_scheduler"))
        parallel.attach(model, ow)
        bs = cfg.get("train_dataloader", {}).get("batch_size", 2)
        size = tuple(cfg.get("crop_size", cfg["model"].get("data_preprocessor", {}).get("size", (1024, 1024))))
        loader = SyntheticLoader(bs, size, rank, world, seed)
        work_dir = cfg.get("work_dir", "./work_dirs/run")
        return cfg_cls(cfg, model, ow, loader, work_dir, rank, world)

    def save_checkpoint(self, path):
        sd = {k: v.detach().cpu() for k, v in self.model.state_dict().items()}
        torch.save(dict(state_dict=sd, meta=dict(iter=self.iter), optimizer=self.ow.optimizer.state_dict()), path)

    def resume(self, path):
        ck = torch.load(path, map_location="cpu")
        self.model.load_state_dict(ck["state_dict"], strict=False)
        self.ow.optimizer.load_state_dict({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ck["optimizer"].items()})
        self.iter = self.ow.iter = ck["meta"]["iter"]

    def train(self, max_iters=None, log_interval=50, ckpt_interval=4000):
        tc = self.cfg.get("train_cfg", {})
        max_iters = max_iters or tc.get("max_iters", 40000)
        os.makedirs(self.work_dir, exist_ok=True)
        log = open(os.path.join(self.work_dir, f"scalars_rank{self.rank}.jsonl"), "a") if self.rank == 0 else None
        t0 = time.time()
        while self.iter < max_iters:
            data = next(self.loader)
            out = self.model.train_step(data, self.ow)
            self.iter += 1
            if self.iter % log_interval == 0 or self.iter == max_iters:
                rec = {k: float(v) for k, v in out.items() if v is not None}  # one sync per log interval
                rec.update(iter=self.iter, lr=self.ow.get_lr(), time=(time.time() - t0) / log_interval,
                           memory=torch.cuda.max_memory_allocated() // (1 << 20))
                t0 = time.time()
                if log:
                    log.write(json.dumps(rec) + "\n")
                    log.flush()
                    print(rec, flush=True)
            if self.rank == 0 and ckpt_interval and self.iter % ckpt_interval == 0:
                self.save_checkpoint(os.path.join(self.work_dir, f"iter_{self.iter}.pth"))
        return self.model
