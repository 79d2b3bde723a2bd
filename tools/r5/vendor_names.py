"""Which hipBLASLt solutions the vendor picks on the long-K shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
import torch.nn.functional as F

dev = torch.device("cuda")
for (M, N, K) in [(25088, 384, 1536), (25088, 384, 1152), (50432, 384, 1536), (6272, 768, 3072), (6272, 2304, 768), (100352, 192, 768)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(5):
        F.linear(a, w)
    torch.cuda.synchronize()
    print(M, N, K)
