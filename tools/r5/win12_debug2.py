import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops
from oracle import ref_ops as R
d = torch.device("cuda")
rel_err = lambda a, b: ((a.double().cpu().reshape(-1) - b.reshape(-1)).norm() / b.norm()).item()
for L in (144, 112):
    B, nH, D = 2, 2, 32
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(B, L, 3 * nH * D, generator=g)
    oref = R.global_attention_core(qkv.double(), nH)
    outs = []
    for i in range(4):
        o, lse = ops.attention_fwd(qkv.reshape(-1, 3 * nH * D).to(d), B, L, nH, D)
        outs.append(o.clone())
    print(L, [f"{rel_err(o, oref):.2e}" for o in outs], "identical runs:", all(torch.equal(outs[0], o) for o in outs))
    e = (outs[0].double().cpu().reshape(B, L, nH, D) - oref.reshape(B, L, nH, D)).abs().amax(-1)   # (B, L, nH)
    bad = (e > 1e-4)
    print("  bad (b, token, head) count", bad.sum().item(), "of", bad.numel(), "; bad tokens of b0 h0:", bad[0, :, 0].nonzero().flatten().tolist()[:40])
    lref = torch.logsumexp(torch.einsum("bihd,bjhd->bhij", qkv.double()[..., :nH*D].reshape(B, L, nH, D), qkv.double()[..., nH*D:2*nH*D].reshape(B, L, nH, D)) / D ** 0.5, -1)
    print("  lse err", rel_err(lse.reshape(B, nH, L), lref))
