"""Micro-benchmark of vfm_gemm (bf16) on the hot-path shapes, next to torch.matmul (hipBLASLt) as a yardstick.
Interleaved rounds in one process, random data (cdna_hip_programming.md 5.4 rules 24/25)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vfmseg_amd import ops

SHAPES = [  # (M, N, K, out_dtype, label)
    (4096, 4096, 1024, torch.bfloat16, "fc1 fwd M4096"),
    (4096, 1024, 1024, torch.float32, "proj M4096"),
    (4, 4096, 1024, torch.bfloat16, "skinny fc1"),
    (4, 1024, 4096, torch.float32, "skinny fc2"),
    (4100, 3072, 1088, torch.bfloat16, "qkv+lora fwd"),
    (4100, 1024, 1024, torch.float32, "proj fwd (+res)"),
    (4100, 4096, 1024, torch.bfloat16, "fc1 fwd"),
    (4100, 1024, 4096, torch.float32, "fc2 fwd (+res)"),
    (4100, 1088, 3072, torch.bfloat16, "qkv dgrad"),
    (4096, 4096, 4096, torch.bfloat16, "4096^3"),
    (2048, 1024, 4096, torch.float32, "fusion_conv"),
    (8192, 1024, 512, torch.bfloat16, "convT2"),
    (32768, 19, 256, torch.float32, "conv_seg"),
]


if os.environ.get("SHAPES") == "small":
    SHAPES = [
        (4096, 64, 1024, torch.bfloat16, "lora T fwd"),
        (4100, 64, 4096, torch.bfloat16, "lora T K4096"),
        (4096, 1024, 64, torch.bfloat16, "lora dgrad"),
        (4096, 256, 1024, torch.bfloat16, "head 256"),
        (2048, 256, 1024, torch.bfloat16, "head 2048x256"),
        (8192, 256, 256, torch.bfloat16, "head 8192x256"),
    ]


if os.environ.get("SHAPES") == "eval":  # DINOv2-L ms_slide_inference: nine 1025-token windows at once
    SHAPES = [
        (9225, 1024, 1024, torch.float32, "eval proj"),
        (9225, 3072, 1024, torch.bfloat16, "eval qkv"),
        (9225, 4096, 1024, torch.bfloat16, "eval fc1"),
        (9225, 1024, 4096, torch.float32, "eval fc2"),
    ]


if os.environ.get("SHAPES") == "sam":   # SAM-H train step at bs 2 (4 images of 32 x 32 tokens), D = 1280
    SHAPES = [
        (4096, 1280, 1280, torch.float32, "sam proj"),
        (4096, 1280, 5120, torch.float32, "sam fc2"),
        (4096, 1280, 3840, torch.bfloat16, "sam qkv dgrad"),
        (4096, 1280, 5120, torch.bfloat16, "sam fc1 dgrad"),
        (4096, 5120, 1280, torch.bfloat16, "sam fc1"),
        (4096, 3840, 1344, torch.bfloat16, "sam qkv+lora"),
    ]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = "cuda"
    for (M, N, K, odt, label) in SHAPES:
        a = torch.randn(M, K, device=dev).bfloat16()
        b = torch.randn(N, K, device=dev).bfloat16()
        c = torch.empty(M, N, dtype=odt, device=dev)
        res = torch.randn(M, N, device=dev) if odt == torch.float32 else None
        bias = torch.randn(N, device=dev)
        fl = 2.0 * M * N * K
        ref = (a.float() @ b.float().t())
        res_line = []
        cfgs = [int(x) for x in os.environ.get("CFGS", "-1,34,33,17").split(",")]
        for cfg in cfgs:
            ops.tune("gemm_cfg", cfg)
            try:
                ops.gemm(a, b, c)
                err = ((c.float() - ref).abs().max() / ref.abs().max()).item()
                t = timeit(lambda: ops.gemm(a, b, c, bias=bias, residual=res))
                res_line.append(f"c{cfg}:{fl/t/1e9:6.0f}" + ("" if err < 2e-2 else f"(ERR {err:.1e})"))
            except Exception as ex:
                res_line.append(f"c{cfg}:fail")
        ops.tune("gemm_cfg", -1)
        t_ref = timeit(lambda: torch.matmul(a, b.t()))
        print(f"{label:16s} M={M:5d} N={N:5d} K={K:5d} torch {fl/t_ref/1e9:6.0f} TF | " + " ".join(res_line), flush=True)


if __name__ == "__main__":
    main()
