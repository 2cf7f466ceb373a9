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


def find_latest_checkpoint(work_dir):
    """mmengine's resume rule: the `last_checkpoint` pointer file if it names an existing file, else the iter_N.pth with the
    largest N (numeric, not lexicographic: iter_8000 < iter_12000)."""
    import re
    if not os.path.isdir(work_dir):
        return None
    ptr = os.path.join(work_dir, "last_checkpoint")
    if os.path.exists(ptr):
        with open(ptr) as f:
            path = f.read().strip()
        if os.path.exists(path):
            return path
    best = None
    for fn in os.listdir(work_dir):
        m = re.fullmatch(r"iter_(\d+)\.pth", fn)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), os.path.join(work_dir, fn))
    return best[1] if best else None


class Runner:
    def __init__(self, cfg, model, optim_wrapper, loader, work_dir, rank=0, world=1):
        self.cfg, self.model, self.ow, self.loader, self.work_dir = cfg, model, optim_wrapper, loader, work_dir
        self.rank, self.world = rank, world
        self.iter = 0
        self.seed = 0

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
        if getattr(ow, "mode", None) is not None:   # AmpOptimWrapper: the autocast dtype is the engine's precision mode (fp16 unless the config says bfloat16)
            from .precision import set_compute_dtype
            set_compute_dtype(ow.mode)
        parallel.attach(model, ow)
        dl = dict(cfg.get("train_dataloader", {}) or {})
        ds = dl.get("dataset")
        bs = dl.get("batch_size", 2)
        loader = None
        if ds is not None and not synthetic:
            # the reference's input pipeline (configs/dg/datasets/*.py -> DGDataset + rare class sampling over a CityscapesDataset
            # folder tree, mmseg transform chain, InfiniteSampler): vfmseg_amd.datasets.  Fails loudly when the data are not there.
            from .datasets import DataLoaderIter
            shuffle = dict(dl.get("sampler", {}) or {}).get("shuffle", True)
            loader = DataLoaderIter(ds, bs, dl.get("num_workers", 0), shuffle, seed, rank, world, infinite=True)
        elif ds is not None and rank == 0:
            import warnings
            warnings.warn(f"train_dataloader.dataset ({ds.get('type', '?') if isinstance(ds, dict) else type(ds).__name__}) is NOT read: "
                          "synthetic=True trains on the synthetic 19-class stream (pass synthetic=False / tools/train.py --data real "
                          "to read it through vfmseg_amd.datasets)")
        if loader is None:
            size = tuple(cfg.get("crop_size", cfg["model"].get("data_preprocessor", {}).get("size", (1024, 1024))))
            loader = SyntheticLoader(bs, size, rank, world, seed)
        work_dir = cfg.get("work_dir", "./work_dirs/run")
        r = cfg_cls(cfg, model, ow, loader, work_dir, rank, world)
        r.seed = seed
        return r

    @staticmethod
    def _rng_path(path, rank):
        return f"{path}.rng_rank{rank}"

    def _rng_state(self):
        from . import functional as Fh
        return dict(iter=self.iter, loader_pos=getattr(self.loader, "i", None), np_random=np.random.get_state(),
                    mask_rng=dict(Fh._seed_state), torch_rng=torch.get_rng_state())

    def save_rank_state(self, path):
        """Every rank > 0 keeps ITS random streams (np.random = crop boxes, and rare class sampling / transforms when num_workers = 0;
        the dropout / query-mask counter; the loader position) in a side file next to rank 0's checkpoint: the streams were seeded
        seed + rank and must stay different after a resume."""
        if self.rank != 0:
            torch.save(self._rng_state(), self._rng_path(path, self.rank))

    def save_checkpoint(self, path):
        """mmengine CheckpointHook: weights, optimiser state, iteration - plus rank 0's RNG consumers of the hot path (the crop-box
        np.random stream, the dropout / query-mask counter, the loader position) so that a resumed run continues the same
        sample and mask sequence - and a `last_checkpoint` pointer file next to it.  Ranks > 0: save_rank_state."""
        sd = {k: v.detach().cpu() for k, v in self.model.state_dict().items()}
        getattr(self.ow, "sync", lambda: None)()   # AmpOptimWrapper: the last step's overflow verdict (AdamW step count, loss scale)
        torch.save(dict(state_dict=sd, meta=self._rng_state(), optimizer=self.ow.optimizer.state_dict(), optim_wrapper=self.ow.state_dict()), path)
        with open(os.path.join(os.path.dirname(path) or ".", "last_checkpoint"), "w") as f:
            f.write(os.path.abspath(path))

    def resume(self, path):
        from . import functional as Fh
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(ck["state_dict"], strict=False)
        self.ow.optimizer.load_state_dict({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ck["optimizer"].items()})
        meta = ck["meta"]
        if "optim_wrapper" in ck:   # AmpOptimWrapper: the loss scaler's state
            self.ow.load_state_dict(ck["optim_wrapper"])
        self.iter = self.ow.iter = meta["iter"]
        if self.rank != 0:
            # rank 0's streams are NOT this rank's: take this rank's side file, else re-derive streams that differ per rank
            side = self._rng_path(path, self.rank)
            if os.path.exists(side):
                own = torch.load(side, map_location="cpu", weights_only=False)
                meta = dict(meta, **{k: own[k] for k in ("np_random", "mask_rng", "torch_rng", "loader_pos")})
            else:
                np.random.seed((self.seed + self.rank + 1000003 * meta["iter"]) % 2 ** 32)
                meta = dict(meta, np_random=None, torch_rng=None,
                            mask_rng=None if meta.get("mask_rng") is None else
                            dict(seed=(int(meta["mask_rng"]["seed"]) + 0x9E3779B1 * self.rank) % 2 ** 62, offset=meta["mask_rng"]["offset"]))
        if meta.get("loader_pos") is not None:
            if hasattr(self.loader, "fast_forward"):
                self.loader.fast_forward(meta["loader_pos"])   # sampler continues where it stopped; workers get fresh seeds
            elif hasattr(self.loader, "i"):
                self.loader.i = meta["loader_pos"]
        if meta.get("np_random") is not None:
            np.random.set_state(meta["np_random"])
        if meta.get("mask_rng") is not None:
            Fh._seed_state.update(meta["mask_rng"])
        if meta.get("torch_rng") is not None:
            torch.set_rng_state(meta["torch_rng"])

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
                rec = {k: float(v.detach() if torch.is_tensor(v) else v) for k, v in out.items() if v is not None}  # one sync per log interval
                rec.update(iter=self.iter, lr=self.ow.get_lr(), time=(time.time() - t0) / log_interval,
                           memory=torch.cuda.max_memory_allocated() // (1 << 20))
                t0 = time.time()
                if log:
                    log.write(json.dumps(rec) + "\n")
                    log.flush()
                    print(rec, flush=True)
            if ckpt_interval and (self.iter % ckpt_interval == 0 or self.iter == max_iters):   # always one at the last iteration
                ck = os.path.join(self.work_dir, f"iter_{self.iter}.pth")
                self.save_checkpoint(ck) if self.rank == 0 else self.save_rank_state(ck)
        return self.model
