"""Host-side cost of the op wrappers (us per call, GPU work negligible / asynchronous)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vfmseg_amd import ops, lib as L
dev = "cuda"
a = torch.randn(128, 64, device=dev).bfloat16(); b = torch.randn(128, 64, device=dev).bfloat16(); c = torch.empty(128, 128, device=dev, dtype=torch.bfloat16)
x = torch.randn(256, 1024, device=dev); y = torch.empty(256, 1024, device=dev, dtype=torch.bfloat16)
w = torch.ones(1024, device=dev); bb = torch.zeros(1024, device=dev); st = torch.empty(256, 2, device=dev)
def t(name, f, n=3000):
    for _ in range(50): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name:28s} {1e6 * (t1 - t0) / n:6.2f} us/call")
t("ops.gemm", lambda: ops.gemm(a, b, c))
t("ops.cast", lambda: ops.cast(x, y))
t("ops.layernorm_fwd", lambda: ops.layernorm_fwd(x, w, bb, 1e-6, y, st))
t("torch.empty", lambda: torch.empty(256, 1024, device=dev))
t("L.stream()", lambda: L.stream())
t("L.ptr", lambda: L.ptr(x))
lib = L.load()
t("raw ctypes vfm_tune", lambda: lib.vfm_tune(b"attn_xcd", 1))
