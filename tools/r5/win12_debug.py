import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops, tables
from oracle import ref_ops as R
d = torch.device("cuda")
for (H, W, win, shift, nH) in [(24, 24, 12, False, 3), (24, 24, 12, True, 3), (12, 12, 12, True, 2), (48, 48, 12, True, 2), (14, 14, 7, True, 3), (16, 16, 8, True, 2)]:
    B, D = 2, 32
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B, H, W, 3 * nH * D, generator=g)
    ntab = (2 * win - 1) ** 2
    rel = 0.5 * torch.randn(ntab, nH, generator=g)
    do = torch.randn(B, H, W, nH * D, generator=g)
    pos, mask = tables.make_pos_mask((H, W), win, shift)
    order, offsets = tables.pos_csr(pos, ntab)
    bias = ops.relpos_bias(rel.to(d), pos.to(d), nH)
    L = win * win
    m = mask.to(d) if shift else None
    o, lse = ops.attention_fwd(qkv.reshape(-1, 3 * nH * D).to(d), B, L, nH, D, swin=(H, W, win, shift), bias=bias, mask=m)
    dqkv, drel = ops.attention_bwd(qkv.reshape(-1, 3 * nH * D).to(d), o, do.reshape(-1, nH * D).to(d), lse, B, L, nH, D, swin=(H, W, win, shift),
                                   bias=bias, mask=m, csr=(order.to(d), offsets.to(d)), ntab=ntab)
    qr, rr = qkv.double().requires_grad_(True), rel.double().requires_grad_(True)
    oref = R.window_attention_core(qr, rr, nH, D, win, shift)
    gq, gr = torch.autograd.grad(oref, [qr, rr], do.double())
    rel_err = lambda a, b: ((a.double().cpu().reshape(-1) - b.reshape(-1)).norm() / b.norm()).item()
    e = (o.double().cpu().reshape(B, H, W, -1) - oref).abs()
    print(f"H{H} win{win} shift{int(shift)}: o {rel_err(o, oref):.2e} (max abs {e.max().item():.2e}, bad tokens {(e.amax(-1) > 1e-4).sum().item()} of {B*H*W}) dqkv {rel_err(dqkv, gq):.2e} drel {rel_err(drel, gr):.2e}")
    if (e.amax(-1) > 1e-4).any():
        idx = (e.amax(-1) > 1e-4).nonzero()[:10]
        print("   first bad (b, y, x):", idx.tolist())
print("--- global attention D=32 / 64, no bias")
for (L, nH, D) in [(144, 3, 32), (100, 2, 32), (70, 2, 32), (144, 2, 64), (64, 2, 32)]:
    B = 2
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(B, L, 3 * nH * D, generator=g)
    o, lse = ops.attention_fwd(qkv.reshape(-1, 3 * nH * D).to(d), B, L, nH, D)
    oref = R.global_attention_core(qkv.double(), nH)
    print(f"L{L} D{D}: o {rel_err(o, oref):.2e}")
print("--- window 12, zero bias table")
H = W = 24; win = 12; nH = 3; D = 32; B = 2
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B, H, W, 3 * nH * D, generator=g)
o, lse = ops.attention_fwd(qkv.reshape(-1, 3 * nH * D).to(d), B, win * win, nH, D, swin=(H, W, win, False))
oref = R.window_attention_core(qkv.double(), torch.zeros(529, nH, dtype=torch.float64), nH, D, win, False)
print(f"no bias: o {rel_err(o, oref):.2e}")
print("--- pattern over L, D=32, fp32 and bf16")
for dt in (torch.float32, torch.bfloat16):
    row = []
    for L in (65, 70, 80, 81, 96, 100, 112, 113, 128, 129, 144, 150, 160):
        B, nH, D = 2, 2, 32
        g = torch.Generator().manual_seed(2)
        qkv = torch.randn(B, L, 3 * nH * D, generator=g).to(dt)
        o, lse = ops.attention_fwd(qkv.reshape(-1, 3 * nH * D).to(d), B, L, nH, D)
        oref = R.global_attention_core(qkv.double(), nH)
        row.append(f"L{L}:{rel_err(o, oref):.1e}")
    print(dt, " ".join(row))
