"""Back-to-back timing of the bf16 flash attention at the backbone's shape (bs 2: 4 image-passes x 16 heads x 1024 + 1 tokens, d = 64) and
the decoder's (2 x 8 heads x 1024): forward and backward (dQ + dK/dV + [cls] finish) launches, HIP events over 50 calls, random data."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vfmseg_amd import ops


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = "cuda"
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        ops.tune(k, int(v))
        print("tune", k, v)
    for label, B, H, n, ex in (("backbone 4x16x(1024+1)", 4, 16, 1024, 1), ("decoder 2x8x1024", 2, 8, 1024, 0), ("eval 9x16x(1024+1)", 9, 16, 1024, 1), ("eval coarse 1x16x(2048+1)", 1, 16, 2048, 1)):
        d = 64
        rows = B * n + B * ex
        qkv = torch.randn(rows, 3 * H * d, device=dev).bfloat16()
        q, k, v = qkv[:, :H * d], qkv[:, H * d:2 * H * d], qkv[:, 2 * H * d:]
        o = torch.empty(rows, H * d, dtype=torch.bfloat16, device=dev)
        lse = torch.empty(B, H, n + ex, device=dev)
        do = torch.randn(rows, H * d, device=dev).bfloat16()
        dqkv = torch.empty_like(qkv)
        fl = 4.0 * B * H * (n + ex) ** 2 * d
        tf = timeit(lambda: ops.attn_fwd(q, k, v, o, lse, B, H, d, n, ex, n, ex, d ** -0.5))
        tb = timeit(lambda: ops.attn_bwd(q, k, v, o, lse, do, dqkv[:, :H * d], dqkv[:, H * d:2 * H * d], dqkv[:, 2 * H * d:], B, H, d, n, ex, n, ex, d ** -0.5))
        print(f"{label:26s} fwd {tf:7.1f} us {fl / tf / 1e6:6.0f} TF | bwd {tb:7.1f} us {2.5 * fl / tb / 1e6:6.0f} TF", flush=True)


if __name__ == "__main__":
    main()
