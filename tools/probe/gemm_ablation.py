import os, sys
sys.path.insert(0, "/root/repo/vision-transformers-pytorch_amd")
import torch
from vtx import ops, options
dev = torch.device("cuda")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, M, N, K, kind in [("s3 fc1 fwd", 25088, 1536, 384, "silu"), ("s3 qkv fwd", 25088, 1152, 384, "bias"), ("s3 fc2 fwd", 25088, 384, 1536, "resid"),
                            ("vit fc1 fwd", 50432, 1536, 384, "silu"), ("vit fc2dgrad", 50432, 1536, 384, "dsilu"), ("vit proj", 50432, 384, 384, "resid")]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    b = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).bfloat16(); z = torch.randn(M, N, device=dev).bfloat16()
    kw = dict(silu=dict(bias=b, act=ops.ACT_SILU, want_aux=True), bias=dict(bias=b), resid=dict(bias=b, resid=res),
              dsilu=dict(act=ops.ACT_DSILU, aux_in=z))[kind]
    line = f"{name:13s} {ops.gemm_kernel_name(torch.bfloat16, N, 0, K=K, M=M)[16:34]:18s}"
    for bits in (0, 1, 2, 4, 8, 3, 7, 15, 12):
        with options.override(GLDS_ABLATE_BITS=bits):   # needs a library built with the ablation switch (tools/probe/build_ablate.sh)
            line += f" {bits:2d}:{timeit(lambda: ops.gemm(x, w, 0, **kw)):6.1f}"
    print(line)
print("bits: 1 no main-loop DMA | 2 no frag reads + MFMA | 4 no stores | 8 no epilogue operand loads")
