"""How much do two INDEPENDENT kernel chains gain from running on two HIP streams?  (LR pass / HR pass of the backbone.)
Each 'chain' here is one op repeated; pairs of different ops are timed alone and concurrently."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vfmseg_amd import ops

dev = "cuda"
M, D, H = 2050, 1024, 16
def rb(*s): return torch.randn(*s, device=dev).bfloat16()
def mk_gemm(m, n, k, gelu=False):
    a, b = rb(m, k), rb(n, k)
    c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    c2 = torch.empty(m, n, dtype=torch.bfloat16, device=dev) if gelu else None
    bias = torch.randn(n, device=dev)
    if gelu:
        return lambda: ops.gemm(a, b, c, bias=bias, ep_mode=ops.EP_GELU_DGELU, c2=c2)
    return lambda: ops.gemm(a, b, c, bias=bias)
def mk_gemm_res(m, n, k):
    a, b = rb(m, k), rb(n, k)
    c = torch.empty(m, n, device=dev); r = torch.randn(m, n, device=dev); cs = torch.randn(n, device=dev); bias = torch.randn(n, device=dev)
    return lambda: ops.gemm(a, b, c, bias=bias, colscale=cs, residual=r)
def mk_ln_bwd(m):
    dy = rb(m, D); x = torch.randn(m, D, device=dev); w = torch.randn(D, device=dev); st = torch.rand(m, 2, device=dev) + 0.5
    dx = torch.zeros(m, D, device=dev); t = torch.empty(m, D, dtype=torch.bfloat16, device=dev); ts = torch.randn(D, device=dev)
    return lambda: ops.layernorm_bwd_scaled(dy, x, w, st, dx, t, ts, accumulate_dx=True)
def mk_ln_fwd(m):
    x = torch.randn(m, D, device=dev); w = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
    y = torch.empty(m, D, dtype=torch.bfloat16, device=dev); st = torch.empty(m, 2, device=dev)
    return lambda: ops.layernorm_fwd(x, w, b, 1e-6, y, st)
def mk_attn(nimg, bwd=False):
    Np = 1024; m = nimg * Np + nimg
    qkv = rb(m, 3 * D); o = torch.empty(m, D, dtype=torch.bfloat16, device=dev); lse = torch.empty(nimg, H, Np + 1, device=dev)
    ops.attn_fwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], o, lse, nimg, H, 64, Np, 1, Np, 1, 0.125)
    if not bwd:
        return lambda: ops.attn_fwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], o, lse, nimg, H, 64, Np, 1, Np, 1, 0.125)
    do = rb(m, D); dqkv = torch.empty(m, 3 * D, dtype=torch.bfloat16, device=dev)
    return lambda: ops.attn_bwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], o, lse, do, dqkv[:, :D], dqkv[:, D:2*D], dqkv[:, 2*D:], nimg, H, 64, Np, 1, Np, 1, 0.125)

def time_alone(f, n=40):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def time_pair(f, g, n=40):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        for _ in range(n): f()
    with torch.cuda.stream(s2):
        for _ in range(n): g()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n * 1e6
    # interleaved issue (launches alternate, as a host loop over two chains would)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(s1): f()
        with torch.cuda.stream(s2): g()
    torch.cuda.synchronize()
    t2 = (time.perf_counter() - t0) / n * 1e6
    return t, t2

ops_ = {
    "fc1 (M2050,N4096,K1024,gelu)": mk_gemm(M, 4096, 1024, True),
    "fc2 (M2050,N1024,K4096,res)": mk_gemm_res(M, 1024, 4096),
    "qkv (M2050,N3072,K1088)": mk_gemm(M, 3072, 1088),
    "ln_bwd": mk_ln_bwd(M), "ln_fwd": mk_ln_fwd(M),
    "attn_fwd(2img)": mk_attn(2), "attn_bwd(2img)": mk_attn(2, True),
}
full = {
    "fc1 (M4100)": mk_gemm(4100, 4096, 1024, True), "fc2 (M4100)": mk_gemm_res(4100, 1024, 4096), "qkv (M4100)": mk_gemm(4100, 3072, 1088),
    "attn_fwd(4img)": mk_attn(4), "attn_bwd(4img)": mk_attn(4, True), "ln_bwd(4100)": mk_ln_bwd(4100),
}
alone = {k: time_alone(f) for k, f in ops_.items()}
for k, v in alone.items(): print(f"alone  {k:34s} {v:7.1f} us")
for k, f in full.items(): print(f"alone  {k:34s} {time_alone(f):7.1f} us   (the batched form used today)")
pairs = [("fc1 (M2050,N4096,K1024,gelu)", "fc1 (M2050,N4096,K1024,gelu)"), ("fc1 (M2050,N4096,K1024,gelu)", "fc2 (M2050,N1024,K4096,res)"),
         ("fc1 (M2050,N4096,K1024,gelu)", "ln_bwd"), ("fc1 (M2050,N4096,K1024,gelu)", "attn_fwd(2img)"), ("fc2 (M2050,N1024,K4096,res)", "attn_bwd(2img)"),
         ("qkv (M2050,N3072,K1088)", "ln_fwd"), ("attn_bwd(2img)", "ln_bwd"), ("attn_fwd(2img)", "attn_bwd(2img)"), ("fc2 (M2050,N1024,K4096,res)", "fc2 (M2050,N1024,K4096,res)")]
for a, b in pairs:
    t, t2 = time_pair(ops_[a], ops_[b])
    print(f"pair   {a[:28]:28s} + {b[:28]:28s}: sum alone {alone[a] + alone[b]:7.1f}  two streams {t:7.1f} / interleaved issue {t2:7.1f} us  -> {t / (alone[a] + alone[b]):.2f}")
