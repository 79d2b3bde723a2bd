"""Which host-side ops launch the tiny torch kernels (FillFunctor, copies, RNG ...) inside one Swin-S train step?
torch.profiler with stacks; prints every aten op that is NOT one of ours, with its count per step and python site."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
import bench
from vtx.optim import FusedAdamW
from vtx.train_step import MixLoss, make_param_groups, train_step
dev = torch.device("cuda")
model = bench.build_model("swin_s", 0.3).to(dev).train()
opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
B = 32
x = torch.randn(B, 3, 224, 224, device=dev); l1 = torch.randint(0, 1000, (B,), device=dev)
data = (x, l1, l1.roll(1), torch.rand(B, device=dev))
for _ in range(2):
    train_step(model, MixLoss(0.1), opt, data)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_step(model, MixLoss(0.1), opt, data)
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    n = e.name
    if ("fill" in n.lower() or "zero" in n.lower() or "copy" in n.lower() or "Memcpy" in n or "Memset" in n) and not n.startswith("void") :
        st = [f for f in (e.stack or [])][:6]
        key = (n[:40], " <- ".join(s.split("/")[-1][:60] for s in st))
        rows[key] = rows.get(key, 0) + 1
for (n, st), c in sorted(rows.items(), key=lambda kv: -kv[1])[:30]:
    print(f"{c:5d} {n:40s} {st[:400]}")
print(prof.key_averages().table(sort_by="count", row_limit=45, max_name_column_width=70))
