import torch
dev = torch.device("cuda:0")
for M, N, K in [(86016, 6144, 1536), (86016, 1536, 1536), (86016, 1536, 6144), (86016, 12288, 1536)]:
    a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(3): torch.matmul(a, w.t())
torch.cuda.synchronize()
