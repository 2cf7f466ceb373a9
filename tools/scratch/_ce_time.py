"""Times the fused upsample+CE kernels at the bs-2 training shapes (x4 LinearHead, x16 VFMHead)."""
import torch
from vfmseg_amd import ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
for h, H in ((128, 512), (32, 512)):
    lg = (torch.randn(2, h, h, 19, generator=g) * 2).to(dev)
    lab = torch.randint(0, 19, (2, H, H), generator=g)
    lab[:, :40] = 255
    lab = lab.to(dev)
    for _ in range(3):
        ops.upsample_ce_loss_acc(lg, lab)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.upsample_ce_loss_acc(lg, lab)
    e1.record()
    torch.cuda.synchronize()
    print(h, H, "%.1f us per call (incl. finish + allocs)" % (e0.elapsed_time(e1) * 1000 / 50))
