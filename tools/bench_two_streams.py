"""Does the train step's backbone forward gain from running as two half-batch chains on two streams (one chain's store-bound epilogues and
launch tails under the other's K loops)?  One ViT-L block forward (LN, qkv, attention, proj + residual, LN, fc1 + GELU + GELU', fc2 +
residual) x 24 on M = 4 x 1025 rows on one stream, against two chains of M = 2 x 1025 rows issued alternately on two streams.
Yardstick for a design decision only."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vfmseg_amd import ops

D, H, NP, L = 1024, 16, 1024, 24
dev = "cuda"


class Chain:
    def __init__(self, nimg):
        self.nimg, self.M = nimg, nimg * NP + nimg
        M = self.M
        bf = torch.bfloat16
        self.x = torch.randn(M, D, device=dev)
        self.a1, self.a2 = torch.empty(M, D, dtype=bf, device=dev), torch.empty(M, D, dtype=bf, device=dev)
        self.st = torch.empty(M, 2, device=dev)
        self.qkv, self.ao = torch.empty(M, 3 * D, dtype=bf, device=dev), torch.empty(M, D, dtype=bf, device=dev)
        self.lse = torch.empty(nimg, H, NP + 1, device=dev)
        self.xm, self.xo = torch.empty(M, D, device=dev), torch.empty(M, D, device=dev)
        self.g, self.hp = ops.empty_ld(M, 4 * D, bf, dev), ops.empty_ld(M, 4 * D, bf, dev)

    def layer(self, W):
        ops.layernorm_fwd(self.x, W["n1w"], W["n1b"], 1e-6, self.a1, self.st)
        ops.gemm(self.a1, W["qkv"], self.qkv, bias=W["qkv_b"])
        ops.attn_fwd(self.qkv[:, :D], self.qkv[:, D:2 * D], self.qkv[:, 2 * D:], self.ao, self.lse, self.nimg, H, 64, NP, 1, NP, 1, 0.125)
        ops.gemm(self.ao, W["proj"], self.xm, bias=W["b1"], colscale=W["g"], residual=self.x)
        ops.layernorm_fwd(self.xm, W["n1w"], W["n1b"], 1e-6, self.a2, self.st)
        ops.gemm(self.a2, W["fc1"], self.g, bias=W["b4"], ep_mode=ops.EP_GELU_DGELU, c2=self.hp)
        ops.gemm(self.g, W["fc2"], self.xo, bias=W["b1"], colscale=W["g"], residual=self.xm)


def weights():
    bf = torch.bfloat16
    r = lambda *s: (torch.randn(*s, device=dev) * 0.02).to(bf)
    return dict(qkv=r(3 * D, D), proj=r(D, D), fc1=r(4 * D, D), fc2=r(D, 4 * D), qkv_b=torch.zeros(3 * D, device=dev), b1=torch.zeros(D, device=dev),
                b4=torch.zeros(4 * D, device=dev), g=torch.ones(D, device=dev), n1w=torch.ones(D, device=dev), n1b=torch.zeros(D, device=dev))


def main():
    Ws = [weights() for _ in range(L)]
    one, h0, h1 = Chain(4), Chain(2), Chain(2)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def run_one():
        for W in Ws:
            one.layer(W)

    def run_two():
        main_s = torch.cuda.current_stream()
        s0.wait_stream(main_s), s1.wait_stream(main_s)
        for W in Ws:
            with torch.cuda.stream(s0):
                h0.layer(W)
            with torch.cuda.stream(s1):
                h1.layer(W)
        main_s.wait_stream(s0), main_s.wait_stream(s1)

    res = {"one": [], "two": []}
    for r in range(6):
        for name, fn in (("one", run_one), ("two", run_two)):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            if r:
                res[name].append((time.perf_counter() - t0) / 3 * 1e3)
    for k, v in res.items():
        v.sort()
        print(f"{k}: median {v[len(v) // 2]:.3f} ms per 24-block forward (min {v[0]:.3f}, max {v[-1]:.3f})")


if __name__ == "__main__":
    main()
