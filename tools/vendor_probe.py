"""Yardstick only (never a dependency of the product): which vendor (hipBLASLt) kernels torch.matmul picks on the GEMM shapes of the
train step, run under `rocprofv3 --kernel-trace` so that the trace carries their names (Tensile encodes tile / MFMA shape / staging in
the name), grid, LDS and register counts.  python tools/vendor_probe.py"""
import torch

SHAPES = [("qkv fwd", 4100, 3072, 1088), ("proj fwd", 4100, 1024, 1024), ("fc1 fwd", 4100, 4096, 1024), ("fc2 fwd", 4100, 1024, 4096),
          ("qkv dgrad", 4100, 1088, 3072), ("4096x4096x1024", 4096, 4096, 1024), ("4096^3", 4096, 4096, 4096), ("M4096 N1024 K4096", 4096, 1024, 4096)]


def main():
    dev = "cuda"
    for label, M, N, K in SHAPES:
        sets = [(torch.randn(M, K, device=dev).bfloat16(), (torch.randn(N, K, device=dev) * 0.05).bfloat16().t(), torch.empty(M, N, dtype=torch.bfloat16, device=dev)) for _ in range(12)]
        for a, bt, c in sets[:2]:
            torch.matmul(a, bt, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for a, bt, c in sets:
            torch.matmul(a, bt, out=c)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / len(sets) * 1e3
        print(f"{label:20s} [{M} x {N} x {K}] {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF", flush=True)
        del sets


if __name__ == "__main__":
    main()
