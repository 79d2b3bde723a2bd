"""What does the fc1-forward epilogue cost in the A-stationary kernel?  N = 1536, K = 384: bias only | SiLU, one output | SiLU + z (the training launch) | dsilu."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops, options
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
for M in (25088, 50432):
    N, K, NSET = 1536, 384, 6
    sets = [dict(x=rn(M, K).bfloat16(), w=(rn(N, K) * 0.05).bfloat16(), b=rn(N), z=rn(M, N).bfloat16()) for _ in range(NSET)]
    kinds = {"bias": lambda d: ops.gemm(d["x"], d["w"], 0, bias=d["b"]),
             "silu": lambda d: ops.gemm(d["x"], d["w"], 0, bias=d["b"], act=ops.ACT_SILU),
             "silu+z": lambda d: ops.gemm(d["x"], d["w"], 0, bias=d["b"], act=ops.ACT_SILU, want_aux=True),
             "gelu+z": lambda d: ops.gemm(d["x"], d["w"], 0, bias=d["b"], act=ops.ACT_GELU, want_aux=True),
             "dsilu": lambda d: ops.gemm(d["x"], d["w"], 0, act=ops.ACT_DSILU, aux_in=d["z"])}
    for name, fn in kinds.items():
        for d in sets: fn(d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(24): fn(sets[i % NSET])
        e1.record(); torch.cuda.synchronize()
        print(f"M {M} N {N} K {K} {name:8s}: {e0.elapsed_time(e1) / 24 * 1e3:6.1f} us", flush=True)
