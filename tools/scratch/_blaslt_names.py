import torch
for (M, N, K) in [(4096, 4096, 1024), (4100, 4096, 1024), (4100, 1024, 4096), (4100, 3072, 1088), (4100, 1088, 3072), (4096, 4096, 4096), (4100, 1024, 1024)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    for _ in range(5): torch.matmul(a, b.t())
torch.cuda.synchronize()
