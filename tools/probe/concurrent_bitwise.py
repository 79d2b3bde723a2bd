"""Are the main-path kernels bit-reproducible while an unrelated weight-gradient launch runs on a second stream?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops
from vtx.tables import make_pos_mask, mask_regions
dev = torch.device("cuda")
M, C, ff, T = 25088, 384, 1536, 196
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
x, dy, dln = rn(M, C).bfloat16(), rn(M, C).bfloat16(), rn(M, C).bfloat16()
gamma = torch.ones(C, device=dev)
_, mean, rstd = ops.layernorm_fwd(x, gamma, torch.zeros(C, device=dev), 1e-6)
w = (rn(C, C) * 0.05).bfloat16()
s1 = ((torch.rand(M // T, device=dev) < 0.8).float() / 0.8)
z = rn(M, ff).bfloat16(); w2 = (rn(ff, C) * 0.05).bfloat16()
# window attention problem (stage 3)
B, H, nH = 128, 14, 12
pos, mask = make_pos_mask((H, H), 7, True); pos = pos.to(dev); region = mask_regions(mask.to(dev))[0]
rel = rn(169, nH) * 0.5
qkv = rn(M, 3 * C).bfloat16(); dout = rn(M, C).bfloat16()
o, lse = ops.wattn_fwd(qkv, rel, pos, region, B, 49, nH, (H, H, 7, True))
# side-stream load: an unrelated grouped weight gradient
sx = [rn(M, C).bfloat16(), rn(M, ff).bfloat16()]
jobs = [(rn(M, ff).bfloat16(), sx[0], True, None), (rn(M, C).bfloat16(), sx[1], True, None),
        (rn(M, C).bfloat16(), sx[0], True, None), (rn(M, 3 * C).bfloat16(), sx[0], True, None)]
side = torch.cuda.Stream()
# PatchMerge LayerNorm backward (stage 3 -> 4): x (B, 14, 14, 384), merged rows of 1536
xm = rn(B, 14, 14, C).bfloat16()
gm = torch.ones(4 * C, device=dev)
ym, mm, rm = ops.layernorm_fwd(xm, gm, torch.zeros(4 * C, device=dev), 1e-5, merge_hw=(14, 14))
dlm = rn(B, 7, 7, 4 * C).bfloat16()

def main_work():
    dx1, dg, db = ops.layernorm_bwd(dln, x, mean, rstd, gamma, dres=dy)
    do = ops.gemm(dx1, w, 0, rowscale=s1, rows_per_scale=T)
    dz = ops.gemm(dy, w2, 0, act=ops.ACT_DSILU, aux_in=z, rowscale=s1, rows_per_scale=T)
    dqkv, drel = ops.wattn_bwd(qkv, o, dout, lse, rel, pos, region, B, 49, nH, (H, H, 7, True), 169)
    dxm, dgm, dbm = ops.layernorm_bwd(dlm, xm, mm, rm, gm, merge_hw=(14, 14))
    return dx1, dg, db, do, dz, dqkv, drel, dxm, dgm

ref = main_work()
torch.cuda.synchronize()
names = ("ln_bwd dx", "ln dgamma", "ln dbeta", "gemm proj-dgrad", "gemm fc2-dgrad dsilu", "wattn dqkv", "wattn drel", "merge ln_bwd dx", "merge ln dgamma")
for mode in ("alone", "with a weight gradient on a second stream"):
    bad = {n: 0 for n in names}
    for it in range(300):
        if mode != "alone":
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    ops.wgrad_group(jobs)
        out = main_work()
        for n, a, b in zip(names, ref, out):
            if not torch.equal(a, b):
                bad[n] += 1
        if mode != "alone":
            torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print(f"{mode}: mismatching launches of 300: {bad}", flush=True)
