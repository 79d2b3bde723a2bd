"""Producer -> consumer on ONE stream (GEMM writes dln, the merge LayerNorm backward reads it) with ALTERNATING inputs, so a
consumer that sees stale lines of the previous iteration's dln gives a wrong (not just repeated) result -- alone and with an
unrelated grouped weight gradient running on a second stream."""
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
gm = torch.ones(4 * C, device=dev)
ym, mm, rm = ops.layernorm_fwd(xm, gm, torch.zeros(4 * C, device=dev), 1e-5, merge_hw=(14, 14))
M = B * 49
dys = [rn(M, 2 * C).bfloat16() for _ in range(2)]                 # PatchMerge: Linear(4C -> 2C, no bias); dgrad: dy [M, 2C] @ W [2C, 4C]
wt = (rn(4 * C, 2 * C) * 0.05).bfloat16()                          # transposed weight copy [in = 4C][out = 2C]
Ms, ff = 25088, 1536
sx = [rn(Ms, C).bfloat16(), rn(Ms, ff).bfloat16()]
jobs = [(rn(Ms, ff).bfloat16(), sx[0], True, None), (rn(Ms, C).bfloat16(), sx[1], True, None),
        (rn(Ms, C).bfloat16(), sx[0], True, None), (rn(Ms, 3 * C).bfloat16(), sx[0], True, None)]
side = torch.cuda.Stream()

def chain(i):
    dln = ops.gemm(dys[i], wt, 0)                                   # [M, 4C]
    dx, dg, db = ops.layernorm_bwd(dln.view(B, 7, 7, 4 * C), xm, mm, rm, gm, merge_hw=(14, 14))
    return dx, dg

refs = [chain(0), chain(1)]
torch.cuda.synchronize()
for mode in ("alone", "with a weight gradient on a second stream"):
    bad = [0, 0]
    for it in range(600):
        if mode != "alone" and it % 2 == 0:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ops.wgrad_group(jobs)
        dx, dg = chain(it & 1)
        bad[0] += not torch.equal(dx, refs[it & 1][0])
        bad[1] += not torch.equal(dg, refs[it & 1][1])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print(f"{mode}: of 600 chains, dx differs {bad[0]}, dgamma differs {bad[1]}", flush=True)
