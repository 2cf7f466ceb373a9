"""One-process A/B of kernel tuning knobs on the two 1024^2 prediction paths of the bench line (DINOv2-L ms_slide_inference with all nine
windows refined, SAM-H slide), settings interleaved in rounds (gpurun boxes differ by up to 8 % on the same build).

    python tools/ab_eval.py base gemm_use_192=0 [--rounds 5 --iters 4 --models dinov2,sam]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vfmseg_amd  # noqa: E402,F401
from vfmseg_amd import ops, presets  # noqa: E402
from vfmseg_amd.registry import MODELS  # noqa: E402
from vfmseg_amd.synth import synth_image, synth_like  # noqa: E402

DEFAULTS = {"gemm_w44_k": 1024, "gemm_use_192": 3, "gemm_use_pp": 184, "gemm_cfg": -1, "attn_xcd": 1, "attn_short_grid": 256}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings", nargs="+")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--models", default="dinov2,sam")
    a = ap.parse_args()
    parsed = [(s, [] if s == "base" else [(x.split("=")[0], int(x.split("=")[1])) for x in s.split(",")]) for s in a.settings]
    # a key "env:NAME" sets the environment variable NAME (read per prediction, e.g. env:VFMSEG_EVAL_OVERLAP=0) instead of a vfm_tune knob
    envs = {k[4:]: os.environ.get(k[4:]) for _, kv in parsed for k, _ in kv if k.startswith("env:")}
    keys = {k for _, kv in parsed for k, _ in kv if not k.startswith("env:")}
    img = synth_image(1, 1024, seed=77).cuda()
    for name in a.models.split(","):
        cfg = presets.dinov2_ms_masked() if name == "dinov2" else presets.sam_linear()
        if name == "dinov2":
            cfg["test_cfg"]["conf"] = 2.0
        model = MODELS.build(cfg)
        model.load_state_dict(synth_like(model.state_dict()), strict=False)
        model = model.cuda().eval()
        times = {s: [] for s, _ in parsed}
        with torch.no_grad():
            model.predict(img)
            for r in range(a.rounds):
                for s, kv in parsed:
                    for k in keys:
                        ops.tune(k, DEFAULTS[k])
                    for e, v0 in envs.items():
                        os.environ.pop(e, None) if v0 is None else os.environ.__setitem__(e, v0)
                    for k, v in kv:
                        if k.startswith("env:"):
                            os.environ[k[4:]] = str(v)
                        else:
                            ops.tune(k, v)
                    model.predict(img)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(a.iters):
                        model.predict(img)
                    torch.cuda.synchronize()
                    times[s].append(1e3 * (time.perf_counter() - t0) / a.iters)
        for k in keys:
            ops.tune(k, DEFAULTS[k])
        base = sorted(times[parsed[0][0]])[a.rounds // 2]
        for s, _ in parsed:
            t = sorted(times[s])
            print(f"{name:7s} {s:32s} median {t[len(t) // 2]:7.3f} ms/img  min {t[0]:7.3f}  max {t[-1]:7.3f}  vs first {100 * (base / t[len(t) // 2] - 1):+5.2f} %", flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
