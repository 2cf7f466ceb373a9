"""Leading-dimension sensitivity of the wide GEMMs: the same [M x N x K] product with operand rows K elements apart (2 KiB at K = 1024,
8 KiB at K = 4096: the rows of an operand panel then fall on few L2 channels) against rows K + PAD apart.  24 cold buffer sets,
configs interleaved.  CFGS=33,50 PADS=0,64,128 python tools/bench_gemm_ld.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vfmseg_amd import ops

NSET = 24
CASES = [("K1024", 4096, 4096, 1024), ("K4096 N1024", 4096, 1024, 4096), ("4096^3", 4096, 4096, 4096)]


def main():
    dev = "cuda"
    cfgs = os.environ.get("CFGS", "33,50").split(",")
    pads = [int(p) for p in os.environ.get("PADS", "0,64,128").split(",")]
    for label, M, N, K in CASES:
        for pad in pads:
            sets = []
            for i in range(NSET):
                a = torch.randn(M, K + pad, device=dev).bfloat16()[:, :K]
                b = (torch.randn(N, K + pad, device=dev) * 0.05).bfloat16()[:, :K]
                sets.append((a, b, torch.empty(M, N, dtype=torch.bfloat16, device=dev)))
            res = []
            for cfg in cfgs:
                ts = []
                for r in range(4):
                    if cfg == "torch":
                        fn = lambda s: torch.matmul(s[0], s[1].t(), out=s[2])
                    else:
                        ops.tune("gemm_cfg", int(cfg.split(":")[0]))
                        ops.tune("pp_dbg", int(cfg.split(":")[1]) if ":" in cfg else 0)
                        fn = lambda s: ops.gemm(s[0], s[1], s[2])
                    fn(sets[0])
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for s in sets:
                        fn(s)
                    e1.record()
                    torch.cuda.synchronize()
                    if r:
                        ts.append(e0.elapsed_time(e1) / NSET * 1e3)
                ops.tune("gemm_cfg", -1)
                ops.tune("pp_dbg", 0)
                t = sorted(ts)[len(ts) // 2]
                res.append(f"c{cfg}: {t:6.1f} us {2.0 * M * N * K / t / 1e6:5.0f} TF")
            print(f"{label:12s} ld = K + {pad:3d}: " + " | ".join(res), flush=True)
            del sets
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
