"""Yardstick only: torch's scaled_dot_product_attention (the vendor's flash kernels: aotriton / CK) on the backbone's attention shape
(4 image-passes x 16 heads x 1025 tokens x 64), forward and forward+backward, next to vfm_attn_fwd / vfm_attn_bwd.  12 buffer sets, interleaved."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from vfmseg_amd import ops

B, H, N, d = 4, 16, 1025, 64
NSET = 12


def timed(fn, sets, rounds=4):
    ts = []
    for r in range(rounds):
        fn(sets[0])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in sets:
            fn(s)
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / len(sets) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    dev = "cuda"
    sets = []
    for i in range(NSET):
        qkv = torch.randn(B * (N - 1) + B, 3 * H * d, device=dev).bfloat16()
        s = dict(qkv=qkv, o=torch.empty(qkv.shape[0], H * d, dtype=torch.bfloat16, device=dev), lse=torch.empty(B, H, N, device=dev),
                 do=torch.randn(qkv.shape[0], H * d, device=dev).bfloat16(), dqkv=torch.empty_like(qkv))
        t = torch.randn(B, H, N, d, device=dev).bfloat16()
        s.update(q4=t.clone().requires_grad_(True), k4=t.clone().requires_grad_(True), v4=t.clone().requires_grad_(True), g4=torch.randn_like(t))
        sets.append(s)
    D = H * d

    def ours_fwd(s):
        ops.attn_fwd(s["qkv"][:, :D], s["qkv"][:, D:2 * D], s["qkv"][:, 2 * D:], s["o"], s["lse"], B, H, d, N - 1, 1, N - 1, 1, d ** -0.5)

    def ours_bwd(s):
        q = s["qkv"]
        ops.attn_bwd(q[:, :D], q[:, D:2 * D], q[:, 2 * D:], s["o"], s["lse"], s["do"], s["dqkv"][:, :D], s["dqkv"][:, D:2 * D], s["dqkv"][:, 2 * D:],
                     B, H, d, N - 1, 1, N - 1, 1, d ** -0.5)

    def vend_fwd(s):
        with torch.no_grad():
            F.scaled_dot_product_attention(s["q4"], s["k4"], s["v4"])

    def vend_fwd_bwd(s):
        o = F.scaled_dot_product_attention(s["q4"], s["k4"], s["v4"])
        o.backward(s["g4"])
        s["q4"].grad = s["k4"].grad = s["v4"].grad = None

    for s in sets[:2]:
        ours_fwd(s)
    fl = 4.0 * B * H * N * N * d
    tf = timed(ours_fwd, sets)
    tb = timed(ours_bwd, sets)
    vf = timed(vend_fwd, sets)
    vfb = timed(vend_fwd_bwd, sets)
    print(f"shape B {B} H {H} N {N} d {d}: forward FLOPs {fl / 1e9:.1f} G, backward 2.5x")
    print(f"ours   forward {tf:6.1f} us ({fl / tf / 1e6:5.0f} TF/s)   backward {tb:6.1f} us ({2.5 * fl / tb / 1e6:5.0f} TF/s)")
    print(f"vendor forward {vf:6.1f} us ({fl / vf / 1e6:5.0f} TF/s)   forward+backward {vfb:6.1f} us -> backward ~ {vfb - vf:6.1f} us ({2.5 * fl / max(vfb - vf, 1e-3) / 1e6:5.0f} TF/s)")


if __name__ == "__main__":
    main()
