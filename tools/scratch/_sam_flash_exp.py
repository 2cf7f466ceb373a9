"""Times the SF_EXP variants of sam_flash.hip (built by hand into tools/scratch/sfexp/sfK.so) on the slide-eval shapes."""
import ctypes, glob, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
vp, cl, ci, cf = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float
torch.manual_seed(0)
H, d, nimg, G = 16, 80, 9, 32
qkv = (torch.randn(nimg * G * G, 3 * H * d, device="cuda") * 0.5).bfloat16()
bias = torch.randn(3 * H * d, device="cuda") * 0.1
out = torch.empty(nimg * G * G, H * d, device="cuda", dtype=torch.bfloat16)
for path in sorted(glob.glob(os.path.join(here, "sfexp", "sf*.so"))):
    lib = ctypes.CDLL(path)
    f = lib.vfm_sam_attn_flash_fwd
    f.argtypes = [vp, cl, vp, vp, vp, vp, cl, ci, ci, ci, ci, ci, cf, vp]
    for S in (14, 32):
        JP = 32 if S == 14 else 64
        tbl = torch.zeros(2, JP, d, device="cuda", dtype=torch.bfloat16)
        tbl[:, :2 * S - 1] = (torch.randn(2, 2 * S - 1, d, device="cuda") * 0.1).bfloat16()
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: f(qkv.data_ptr(), qkv.stride(0), bias.data_ptr(), tbl[0].data_ptr(), tbl[1].data_ptr(), out.data_ptr(), out.stride(0),
                         nimg, G, S, H, d, d ** -0.5, st)
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        print(os.path.basename(path), "S", S, "%.1f us" % (e0.elapsed_time(e1) * 50), flush=True)
