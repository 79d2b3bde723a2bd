"""Which layer's output first differs between GEMM_ASTAT = 0 and 1 (and between two runs with 1) in a full-size forward?"""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
import bench
from vtx import options

ap = argparse.ArgumentParser(); ap.add_argument("--model", default="swin_s"); ap.add_argument("--dp", type=float, default=0.3)
ap.add_argument("--batch", type=int, default=0)
a = ap.parse_args()
d = torch.device("cuda")
B = a.batch or bench.default_batch(a.model)
g = torch.Generator(device=d).manual_seed(77)
x = torch.randn(B, 3, 224, 224, device=d, generator=g)


def run(astat):
    torch.manual_seed(5)
    model = bench.build_model(a.model, a.dp).to(d).train()
    outs = {}
    hs = []
    for n, m in model.named_modules():
        if type(m).__name__ in ("TransformerLayer", "SwinTransformerLayer", "PatchMerge", "PatchEmbedding") or n.count(".") == 1:
            hs.append(m.register_forward_hook(lambda mod, i, o, n=n: outs.__setitem__(n, (o[0] if isinstance(o, tuple) else o).detach().float().clone()) if torch.is_tensor(o) or isinstance(o, tuple) else None))
    torch.manual_seed(6)
    with options.override(GEMM_ASTAT=astat), torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = model(x).float()
    for h in hs: h.remove()
    outs["logits"] = y
    return outs

o0, o1, o1b = run(0), run(1), run(1)
for n in o0:
    if n not in o1: continue
    e01 = (o0[n] - o1[n]).abs().max().item(); e11 = (o1[n] - o1b[n]).abs().max().item()
    nbad = int(((o0[n] - o1[n]).abs() > 0).sum().item())
    print(f"{n:40s} shape {tuple(o0[n].shape)}  max|astat0 - astat1| {e01:.3e} ({nbad} elements)   max|run - run| {e11:.3e}", flush=True)
