"""Times the SF_EXP variants of sam_flash.hip (built by hand into tools/scratch/sfexp/sfK.so) on the slide-eval shapes; for the
SF_EXP=9 build prints wave 0's phase clocks (s_memtime ticks, 100 MHz) averaged over the blocks."""
import ctypes, glob, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
vp, cl, ci, cf = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float
torch.manual_seed(0)
H, d, nimg, G = 16, 80, int(os.environ.get("NIMG", 9)), 32
qkv = (torch.randn(nimg * G * G, 3 * H * d, device="cuda") * 0.5).bfloat16()
bias = torch.randn(3 * H * d, device="cuda") * 0.1
out = torch.empty(nimg * G * G, H * d, device="cuda", dtype=torch.bfloat16)
names = ["load+bar1", "Tprod+gather", "bar2", "commit0(wait fetch0)", "bar3", "fetch issue + QK", "softmax", "PV", "commit(wait fetch)", "barrier", "loop top", "tail"]
for path in sorted(glob.glob(os.path.join(here, "sfexp", "sf*.so"))):
    lib = ctypes.CDLL(path)
    f = lib.vfm_sam_attn_flash_fwd_train
    f.argtypes = [vp, cl, vp, vp, vp, vp, cl, ci, ci, ci, ci, ci, cf, vp, vp, vp]
    for S in (14, 32):
        JP = 32 if S == 14 else 64
        nws = 1 if S == 32 else 3
        rows = nimg * nws * nws * H * (256 if S == 14 else 1024)
        lse = torch.zeros(rows, device="cuda")
        qext = torch.empty(rows, JP, device="cuda", dtype=torch.bfloat16)
        tbl = torch.zeros(2, JP, d, device="cuda", dtype=torch.bfloat16)
        tbl[:, :2 * S - 1] = (torch.randn(2, 2 * S - 1, d, device="cuda") * 0.1).bfloat16()
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: f(qkv.data_ptr(), qkv.stride(0), bias.data_ptr(), tbl[0].data_ptr(), tbl[1].data_ptr(), out.data_ptr(), out.stride(0),
                         nimg, G, S, H, d, d ** -0.5, lse.data_ptr(), qext.data_ptr(), st)
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
        if path.endswith("sf9.so"):
            lse.zero_()
            call()
            torch.cuda.synchronize()
            nb = nimg * nws * nws * H * (2 if S == 14 else 8)
            v = lse[:nb * 16].view(nb, 16).double().sum(0).cpu()
            n = float(nb)
            tot = float(v[:12].sum()) / n
            print("   blocks %d, ticks per block %.0f (%.2f us at 100 MHz)" % (n, tot, tot / 100))
            for k in range(12):
                print("   %-24s %7.1f ticks  %5.1f %%" % (names[k], float(v[k]) / n, 100 * float(v[k]) / n / tot))
