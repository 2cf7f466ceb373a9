"""The wide, store-heavy GEMMs of the train step as the step sees them: the fc1 forward (GELU + gelu' outputs: 67 MB of stores), the fc2
input gradient (multiply by the saved gelu': 34 MB read + 34 MB written) and the qkv projection, each launch on a DIFFERENT set of
buffers (24 sets, as 24 layers: weights and activations are not L2-warm from the previous launch), configs interleaved in one
process on random data (cdna_hip_programming.md 5.4 rules 24 / 25).  CFGS=34,33,37 python tools/bench_gemm_step.py
A CFGS entry "torch" is the vendor yardstick under the same conditions: torch.matmul (hipBLASLt) on the same 24 cold buffer sets, plain
bf16 product with NO epilogue (no bias / GELU / second output) - a measuring stick only, never a dependency of the product."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vfmseg_amd import ops

NSET = int(os.environ.get("NSET", "24"))
CASES = [  # (label, M, N, K, kind)
    ("fc1 fwd gelu+dgelu", 4100, 4096, 1024, "gelu_dgelu"),
    ("fc2 dgrad mul", 4100, 4096, 1024, "mul"),
    ("qkv fwd bias", 4100, 3072, 1088, "bias"),
    ("fc1 fwd M4096", 4096, 4096, 1024, "gelu_dgelu"),
    ("eval fc1 gelu", 9225, 4096, 1024, "gelu"),
    ("N1024 K4096 bias", 4096, 1024, 4096, "bias"),
    ("N1024 K1024 bias", 4096, 1024, 1024, "bias"),
    ("qkv dgrad", 4100, 1088, 3072, "bias"),
    ("proj fwd res", 4100, 1024, 1024, "res"),
    ("fc2 fwd res", 4100, 1024, 4096, "res"),
    ("plain 4096x4096x1024", 4096, 4096, 1024, "plain"),
    ("plain 4096^3", 4096, 4096, 4096, "plain"),
]
EVAL_CASES = [  # SHAPES=eval: the GEMMs of one 1024^2 prediction (DINOv2-L ms_slide: nine 512^2 windows in one batch, then the 512 x 1024 LR pass; SAM-H slide)
    ("HR qkv", 9225, 3072, 1024, "bias"), ("HR proj", 9225, 1024, 1024, "bias"), ("HR fc1 gelu", 9225, 4096, 1024, "gelu"), ("HR fc2", 9225, 1024, 4096, "bias"),
    ("LR qkv", 2049, 3072, 1024, "bias"), ("LR proj", 2049, 1024, 1024, "bias"), ("LR fc1 gelu", 2049, 4096, 1024, "gelu"), ("LR fc2", 2049, 1024, 4096, "bias"),
    ("SAM qkv", 9216, 3840, 1280, "bias"), ("SAM proj", 9216, 1280, 1280, "bias"), ("SAM fc1 gelu", 9216, 5120, 1280, "gelu"), ("SAM fc2", 9216, 1280, 5120, "bias"),
]
if os.environ.get("SHAPES") == "eval":
    CASES = EVAL_CASES
if os.environ.get("CASES"):
    CASES = [CASES[int(i)] for i in os.environ["CASES"].split(",")]


def main():
    dev = "cuda"
    # CFGS entries: CFG or CFG:DBG (vfm_tune pp_dbg diagnostic build of that config; garbage results, timing only)
    cfgs = [x for x in os.environ.get("CFGS", "34,33,37").split(",")]
    ncase = int(os.environ.get("NCASE", str(len(CASES))))
    rounds = int(os.environ.get("ROUNDS", "5"))
    for label, M, N, K, kind in CASES[:ncase]:
        sets = []
        for i in range(NSET):
            a = torch.randn(M, K, device=dev).bfloat16()
            b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
            sets.append(dict(a=a, b=b, c=torch.empty(M, N, dtype=torch.bfloat16, device=dev), c2=torch.empty(M, N, dtype=torch.bfloat16, device=dev),
                             aux=torch.randn(M, N, device=dev).bfloat16(), bias=torch.randn(N, device=dev), bt=b.t(),
                             cf=torch.randn(M, N, device=dev) if kind == "res" else None))

        def run(s):
            if kind == "gelu_dgelu":
                ops.gemm(s["a"], s["b"], s["c"], bias=s["bias"], ep_mode=ops.EP_GELU_DGELU, c2=s["c2"])
            elif kind == "mul":
                ops.gemm(s["a"], s["b"], s["c"], ep_mode=ops.EP_MUL, aux=s["aux"])
            elif kind == "gelu":
                ops.gemm(s["a"], s["b"], s["c"], bias=s["bias"], ep_mode=ops.EP_GELU)
            elif kind == "res":   # fp32 residual stream: out = res + ls * (a b^T + bias), as proj / fc2 forward
                ops.gemm(s["a"], s["b"], s["cf"], bias=s["bias"], residual=s["cf"])
            elif kind == "plain":
                ops.gemm(s["a"], s["b"], s["c"])
            else:
                ops.gemm(s["a"], s["b"], s["c"], bias=s["bias"])

        def run_vendor(s):
            torch.matmul(s["a"], s["bt"], out=s["c"])

        fl = 2.0 * M * N * K
        times = {c: [] for c in cfgs}
        for r in range(rounds + 1):
            for cfg in cfgs:
                if cfg == "torch":
                    run_vendor(sets[0])
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for s in sets:
                        run_vendor(s)
                    e1.record()
                    torch.cuda.synchronize()
                    if r > 0:
                        times[cfg].append(e0.elapsed_time(e1) / NSET * 1e3)
                    continue
                ops.tune("gemm_cfg", int(cfg.split(":")[0]))
                dbg = cfg.split(":")[1] if ":" in cfg else "0"
                ops.tune("pp_dbg", int(dbg.rstrip("b")))
                ops.tune("ps_burst", 1 if dbg.endswith("b") else 0)
                try:
                    run(sets[0])
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for s in sets:
                        run(s)
                    e1.record()
                    torch.cuda.synchronize()
                    if r > 0:
                        times[cfg].append(e0.elapsed_time(e1) / NSET * 1e3)
                except Exception as ex:
                    times[cfg].append(float("nan"))
                    if r == 0:
                        print(f"  cfg {cfg}: {str(ex)[:100]}")
        ops.tune("gemm_cfg", -1)
        ops.tune("pp_dbg", 0)
        ops.tune("ps_burst", 0)
        line = " | ".join(f"c{c}: {sorted(t)[len(t) // 2]:6.1f} us (min {min(t):6.1f}) {fl / sorted(t)[len(t) // 2] / 1e6:5.0f} TF" for c, t in times.items())
        print(f"{label:20s} [{M} x {N} x {K}] {line}", flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
