"""64- vs 128-row tiles of the LDS-DMA GEMM on the N = 384 stage-3 shapes at the row counts stochastic-depth compaction
produces (B_k kept samples x 196 tokens): does the 128-row tile win once its tiles fit ONE round of 512 resident workgroups?"""
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


for Bk in (128, 118, 111, 108, 104, 96, 90):
    M = Bk * 196
    line = f"Bk={Bk:3d} M={M:6d} t64={((M+63)//64)*3:5d} t128={((M+127)//128)*3:4d} |"
    for name, N, K, kw in (("proj", 384, 384, "resid"), ("fc2", 384, 1536, "resid"), ("fc1dg", 384, 1536, ""), ("qkvdg", 384, 1152, "")):
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        b = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).bfloat16()
        args = dict(bias=b, resid=res) if kw else {}
        ts = []
        for bm in (64, 128):
            with options.override(GLDS_BM=bm):
                ts.append(timeit(lambda: ops.gemm(x, w, 0, **args)))
        line += f" {name} {ts[0]:5.1f}/{ts[1]:5.1f}"
    print(line)
