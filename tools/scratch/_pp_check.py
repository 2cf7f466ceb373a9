"""Correctness screen of the ping-pong GEMM (cfg 30) over odd shapes, several repetitions (race screen)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vfmseg_amd import ops

cfg = int(os.environ.get("CFG", "30"))
torch.manual_seed(0)
bad = 0
for (M, N, K) in [(256, 256, 256), (256, 256, 320), (300, 520, 320), (4096, 4096, 1024), (4100, 3072, 1088), (1000, 1000, 4096),
                  (4096, 1024, 4096), (515, 2048, 64 * 7)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = torch.randn(N, K, device="cuda").bfloat16()
    ref = a.float() @ b.float().t()
    for rep in range(6):
        c = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        ops.tune("gemm_cfg", cfg)
        ops.tune("gemm_split_tail", 0)
        ops.gemm(a, b, c)
        ops.tune("gemm_cfg", -1)
        ops.tune("gemm_split_tail", 1)
        err = ((c - ref).abs().max() / ref.abs().max()).item()
        if not (err < 1e-5):
            bad += 1
            print(f"BAD M={M} N={N} K={K} rep={rep} err={err:.3e}", flush=True)
            break
    else:
        print(f"ok  M={M} N={N} K={K} err={err:.2e}", flush=True)
print("FAILED" if bad else "ALL OK")
