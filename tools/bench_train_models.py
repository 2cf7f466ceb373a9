"""Train-step throughput of the other backbone configurations (bs 2, bf16, synthetic 1024^2 samples -> LR+HR 512^2 passes),
same step as bench.py: EVA02-L (BASELINE config 4), CLIP ViT-L/16, SAM-ViT-H under MsVFMEncoderDecoder."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def run(name, preset, steps=6, warmup=2, batch=2):
    import vfmseg_amd  # noqa: F401
    from vfmseg_amd import functional as Fh, presets
    from vfmseg_amd.optim import PEFTOptimWrapperConstructor
    from vfmseg_amd.precision import set_compute_dtype
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.synth import synth_like
    set_compute_dtype("bf16")
    dev = torch.device("cuda", 0)
    model = MODELS.build(getattr(presets, preset)())
    sd = {k: v for k, v in synth_like(model.state_dict()).items() if "rope." not in k}
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).train()
    oc = presets.optim_cfg()
    ow = PEFTOptimWrapperConstructor(oc["optim_wrapper"])(model, oc["param_scheduler"])
    Fh.manual_seed(7)
    data = bench.make_batch(batch, 0, 0, dev)
    for _ in range(warmup):
        model.train_step(data, ow)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.train_step(data, ow)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"metric": "train images/sec", "model": name, "value": round(batch / dt, 2), "unit": "images/s", "ms_per_step": round(dt * 1e3, 2),
                      "batch": batch, "dtype": "bf16", "data": "synthetic", "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}), flush=True)
    del model, ow
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    only = os.environ.get("ONLY")
    if only == "sam":
        run("SAM-ViT-H + LoRA(qkv) + LinearHead + VFMHead (lora_sam_ms_masked)", "sam_ms_masked", steps=3, warmup=1)
        sys.exit(0)
    if only == "clip":
        run("CLIP ViT-L/16 + LoRA(mlp.c_fc, mlp.c_proj) + LinearHead + VFMHead (lora_clip_ms_masked)", "clip_ms_masked", steps=4, warmup=2)
        sys.exit(0)
    if only == "eva":
        run("EVA02-L + LoRA(attn.proj) + LinearHead + VFMHead (lora_eva02_ms_masked)", "eva02_ms_masked", steps=4, warmup=2)
        sys.exit(0)
    run("EVA02-L + LoRA(attn.proj) + LinearHead + VFMHead (lora_eva02_ms_masked)", "eva02_ms_masked")
    run("CLIP ViT-L/16 + LoRA(mlp.c_fc, mlp.c_proj) + LinearHead + VFMHead (lora_clip_ms_masked)", "clip_ms_masked")
    run("SAM-ViT-H + LoRA(qkv) + LinearHead + VFMHead (lora_sam_ms_masked)", "sam_ms_masked")
