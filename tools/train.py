#!/usr/bin/env python
"""Training entry point with the reference's flags (tools/train.py:19-61): CONFIG [--work-dir --resume --amp
--cfg-options k=v ... --launcher {none,pytorch,slurm,mpi}].  Launch multi-GPU runs with
`python -m torch.distributed.run --nproc-per-node N tools/train.py CONFIG --launcher pytorch`."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser(description="Train a segmentor on the MI355X HIP path")
    ap.add_argument("config")
    ap.add_argument("--work-dir")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--amp", action="store_true", help="AmpOptimWrapper: fp16 autocast (the fp16 twin library) with a dynamic loss scale; optim_wrapper.dtype=bfloat16 for bf16")
    ap.add_argument("--cfg-options", nargs="+")
    ap.add_argument("--launcher", choices=["none", "pytorch", "slurm", "mpi"], default="none")
    ap.add_argument("--local_rank", "--local-rank", type=int, default=0)
    ap.add_argument("--max-iters", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "bf16x3", "fp16"])
    ap.add_argument("--data", choices=["auto", "synthetic", "real"], default="auto",
                    help="auto: read train_dataloader.dataset when its data_root exists, else the synthetic 19-class stream")
    a = ap.parse_args()
    if "LOCAL_RANK" not in os.environ:
        os.environ["LOCAL_RANK"] = str(a.local_rank)
    import vfmseg_amd  # noqa: F401
    from vfmseg_amd.config import Config, parse_cfg_options
    from vfmseg_amd.precision import set_compute_dtype
    from vfmseg_amd.runner import Runner
    set_compute_dtype(a.dtype)
    cfg = Config.fromfile(a.config)
    cfg.merge_from_dict(parse_cfg_options(a.cfg_options))
    if a.amp:   # tools/train.py:87-102: OptimWrapper -> AmpOptimWrapper with a dynamic loss scale
        ow_type = cfg["optim_wrapper"].get("type", "OptimWrapper")
        if ow_type == "AmpOptimWrapper":
            print("AMP training is already enabled in your config.")
        else:
            assert ow_type == "OptimWrapper", f"`--amp` is only supported when the optimizer wrapper type is `OptimWrapper` but got {ow_type}."
            cfg["optim_wrapper"]["type"] = "AmpOptimWrapper"
            cfg["optim_wrapper"]["loss_scale"] = "dynamic"
    if cfg["optim_wrapper"].get("type") == "AmpOptimWrapper":
        # mmengine's AmpOptimWrapper runs the model under torch.autocast(dtype or fp16): here, the engine's precision mode
        amp_dt = cfg["optim_wrapper"].get("dtype")
        mode = "bf16" if amp_dt in ("bfloat16", "bf16") else "fp16"
        print(f"AmpOptimWrapper: autocast dtype {amp_dt or 'float16'} -> precision mode {mode} (dynamic loss scale)")
        set_compute_dtype(mode)
    cfg["work_dir"] = a.work_dir or cfg.get("work_dir") or os.path.join("./work_dirs", os.path.splitext(os.path.basename(a.config))[0])
    if "train_cfg" not in cfg["model"] or cfg["model"]["train_cfg"] is None:
        cfg["model"]["train_cfg"] = {}
    cfg["model"]["train_cfg"]["work_dir"] = cfg["work_dir"]           # tools/train.py:108-109
    cfg["model"]["train_cfg"]["log_config"] = cfg.get("log_config", dict(interval=50, img_interval=500))
    synthetic = a.data == "synthetic"
    if a.data == "auto":
        ds = dict(cfg.get("train_dataloader", {}) or {}).get("dataset") or {}
        src = ds.get("source", ds) if isinstance(ds, dict) else {}
        root = src.get("data_root") if isinstance(src, dict) else None
        synthetic = not (root and os.path.isdir(root))
        if synthetic:
            print(f"[train] data_root {root!r} not found: training on the synthetic stream (--data real to insist)")
    runner = Runner.from_cfg(cfg, synthetic=synthetic)
    if a.resume:
        from vfmseg_amd.runner import find_latest_checkpoint
        ck = find_latest_checkpoint(cfg["work_dir"])
        if ck:
            runner.resume(ck)
    runner.train(a.max_iters, log_interval=cfg.get("default_hooks", {}).get("logger", {}).get("interval", 50),
                 ckpt_interval=cfg.get("default_hooks", {}).get("checkpoint", {}).get("interval", 4000))


if __name__ == "__main__":
    main()
