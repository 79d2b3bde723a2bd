"""Which torch glue ops (fill / copy / elementwise) does one Swin-S train step launch, and from where?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from vtx.train_step import MixLoss, make_param_groups, train_step
from vtx.optim import FusedAdamW
dev = torch.device("cuda")
torch.manual_seed(0)
name = sys.argv[1] if len(sys.argv) > 1 else "swin_s"
B = 256 if name == "vit_s16" else 128
model = bench.build_model(name, 0.3 if name == "swin_s" else 0.1).to(dev).train()
crit = MixLoss(eps=0.1)
opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
x = torch.randn(B, 3, 224, 224, device=dev); l1 = torch.randint(0, 1000, (B,), device=dev); l2 = torch.randint(0, 1000, (B,), device=dev)
data = (x, l1, l2, torch.rand(B, device=dev))
step = lambda: train_step(model, crit, opt, data, clip_grad_norm=5.0, autocast_dtype=torch.bfloat16, ddp=None)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.key.startswith("aten::") and e.device_time_total > 5]
for k, c, t in sorted(rows, key=lambda r: -r[2])[:14]:
    print(f"{name}: {t:9.0f} us  x{c:4d}  {k}")
