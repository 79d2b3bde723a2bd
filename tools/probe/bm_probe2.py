"""64- vs 128-row tiles on launches of 256..512 128-row tiles outside the compaction case (Swin stage 4, PVT, small batches)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops, options
dev = torch.device("cuda")


def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, M, N, K, kw in (("swin4 proj", 6272, 768, 768, "resid"), ("swin4 fc2", 6272, 768, 3072, "resid"), ("swin4 fc1dg", 6272, 768, 3072, ""),
                          ("swin4 qkvdg", 6272, 768, 2304, ""), ("swin4 qkv", 6272, 2304, 768, "bias"), ("swin4 fc1", 6272, 3072, 768, "silu"),
                          ("pvt3 fc2", 25088, 320, 1280, "resid"), ("pvt4 fc2", 6272, 512, 2048, "resid"), ("pvt4 fc1", 6272, 2048, 512, "silu"),
                          ("vit b128 proj", 25216, 384, 384, "resid"), ("dino local fc2", 18944, 384, 1536, "resid"), ("dino local qkv", 18944, 1152, 384, "bias")):
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    b = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).bfloat16()
    args = dict(resid=dict(bias=b, resid=res), bias=dict(bias=b), silu=dict(bias=b, act=ops.ACT_SILU, want_aux=True)).get(kw, {})
    ts = []
    for bm in (64, 128):
        with options.override(GLDS_BM=bm):
            ts.append(timeit(lambda: ops.gemm(x, w, 0, **args)))
    t128 = ((N + 127) // 128) * ((M + 127) // 128)
    print(f"{name:16s} M={M:6d} N={N:5d} K={K:5d} t128={t128:5d}  64-row {ts[0]:6.1f} us   128-row {ts[1]:6.1f} us   picked {ops.gemm_kernel_name(torch.bfloat16, N, 0, K=K, M=M)}")
