import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops
d = torch.device("cuda")
for L in (144, 80):
    B, nH, D = 1, 1, 32
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(B, L, 3 * nH * D, generator=g)
    o, lse = ops.attention_fwd(qkv.reshape(-1, 3 * nH * D).to(d), B, L, nH, D)
    q, k = qkv.double()[0, :, :D], qkv.double()[0, :, D:2 * D]
    S = q @ k.t() / D ** 0.5
    lref = torch.logsumexp(S, -1)
    delta = torch.exp(lse.double().cpu()) - torch.exp(lref)            # per query
    E = torch.exp(S)
    for qi in (0, 1, 17, 50, L - 1):
        dm = (E[qi] + delta[qi]).abs()      # missing key k: delta = -E[k]
        dp = (E[qi] - delta[qi]).abs()      # doubled key k: delta = +E[k]
        print(f"L{L} q{qi}: delta {delta[qi].item():+.4e}; best 'missing' key {dm.argmin().item()} (res {dm.min().item():.2e}); best 'doubled' key {dp.argmin().item()} (res {dp.min().item():.2e}); sumE {E[qi].sum().item():.3e}")
    # recompute with the last key's score replaced etc: also check whether S uses q.k over only part of D
    for dd in (8, 16, 24):
        Sp = q[:, :dd] @ k[:, :dd].t() / D ** 0.5
        print(f"   if only {dd} channels: lse err {((torch.logsumexp(Sp, -1) - lse.double().cpu()).abs().max().item()):.2e}")
