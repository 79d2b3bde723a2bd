"""Small reproducer attempt: PatchMerge backward (weight gradient, dgrad GEMM, merge LayerNorm backward on the main stream) next to
the grouped weight gradient of a Swin stage-4 layer on a second stream -- is dx bit-reproducible?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops
dev = torch.device("cuda")
B, C = 128, 384
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
xm = rn(B, 14, 14, C).bfloat16()
gm = (1 + 0.1 * rn(4 * C)).float()
ln, mm, rm = ops.layernorm_fwd(xm, gm, torch.zeros(4 * C, device=dev), 1e-5, merge_hw=(14, 14))
M = B * 49
dy = rn(M, 2 * C).bfloat16()
wt = (rn(4 * C, 2 * C) * 0.05).bfloat16()
C4, ff4 = 768, 3072
jobs = [(rn(M, C4).bfloat16(), rn(M, ff4).bfloat16(), True, None), (rn(M, ff4).bfloat16(), rn(M, C4).bfloat16(), True, None),
        (rn(M, C4).bfloat16(), rn(M, C4).bfloat16(), True, None), (rn(M, 3 * C4).bfloat16(), rn(M, C4).bfloat16(), True, None)]
side = torch.cuda.Stream()

def merge_bwd():
    dW, _ = ops.wgrad(dy, ln.view(M, 4 * C), want_bias=False)
    dln = ops.gemm(dy, wt, 0)
    dx, dg, db = ops.layernorm_bwd(dln.view(B, 7, 7, 4 * C), xm, mm, rm, gm, merge_hw=(14, 14))
    return dx, dg, dW

ref = merge_bwd()
torch.cuda.synchronize()
for mode in ("alone", "with the stage-4 grouped weight gradient on a second stream"):
    bad = [0, 0, 0]
    for it in range(1500):
        if mode != "alone":
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ops.wgrad_group(jobs)
        out = merge_bwd()
        for i in range(3):
            bad[i] += not torch.equal(out[i], ref[i])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print(f"{mode}: of 1500: dx differs {bad[0]}, dgamma {bad[1]}, dW {bad[2]}", flush=True)
