import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
for M, N, K in [(16384, 640, 2560), (4096, 1280, 5120), (4096, 1280, 1280), (4096, 3840, 1280), (16384, 5120, 640), (16384, 640, 640), (16384, 640, 5760), (65536, 320, 320)]:
    a = (torch.randn(M, K, device=dev)).half(); w = (torch.randn(N, K, device=dev)).half(); b = torch.randn(N, device=dev).half()
    for _ in range(3): F.linear(a, w, b)
    torch.cuda.synchronize()
