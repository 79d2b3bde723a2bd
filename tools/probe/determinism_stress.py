"""Stress: is a full-size bf16 train step bit-reproducible?  N trials of (same weights, same seed, 2 steps) in one process;
prints the trials whose parameters differ from trial 0 and WHICH parameters.
    python tools/probe/determinism_stress.py [swin_s|vit_s16|pvt_small] [N]      (env: VTX_LAYER_CALL=0, VTX_SIDE_WGRAD=0 ...)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
import bench
from vtx.optim import FusedAdamW
from vtx.train_step import MixLoss, make_param_groups, train_step

name = sys.argv[1] if len(sys.argv) > 1 else "swin_s"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B = {"swin_s": 128, "vit_s16": 256, "pvt_small": 128, "twins_svt_s": 128}[name]
dev = torch.device("cuda")
torch.manual_seed(0)
model = bench.build_model(name, 0.3 if name == "swin_s" else 0.1).to(dev).train()
init = {k: v.clone() for k, v in model.state_dict().items()}
g = torch.Generator().manual_seed(3)
x = torch.randn(B, 3, 224, 224, generator=g).to(dev)
l1 = torch.randint(0, 1000, (B,), generator=g).to(dev)
data = (x, l1, l1.roll(1), torch.rand(B, generator=g).to(dev))
crit = MixLoss(0.1)
names = [n for n, _ in model.named_parameters()]
ref = None
bad = 0
for t in range(N):
    model.load_state_dict(init)
    opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
    torch.manual_seed(7)
    losses = [train_step(model, crit, opt, data) for _ in range(2)]
    torch.cuda.synchronize()
    cur = [p.detach().clone() for p in model.parameters()] + [torch.stack([l.detach().float() for l in losses])]
    if ref is None:
        ref = cur
        continue
    diff = [i for i, (a, b) in enumerate(zip(ref, cur)) if not torch.equal(a, b)]
    if diff:
        bad += 1
        if bad <= 3:
            what = [(names[i] if i < len(names) else "losses", float((ref[i].double() - cur[i].double()).abs().max())) for i in diff[:6]]
            same = [names[i] for i in range(len(names)) if i not in set(diff)]
            print(f"trial {t}: {len(diff)} tensors differ, first: {what}; unchanged ({len(same)}): {same[:3]} ...", flush=True)
print(f"{name}: {bad} of {N - 1} trials differ from trial 0  (env: " +
      " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VTX_")) + ")")
