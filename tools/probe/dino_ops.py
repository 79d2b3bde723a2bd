"""Which torch ops (not vtx kernels) cost GPU time in a DINO step?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from vtx.dino import DINOLoss, dino_train_step
from vtx.optim import FusedAdamW
from vtx.train_step import make_param_groups
dev = torch.device("cuda"); torch.manual_seed(0)
B = 64
model = bench.build_model("dino", 0.1).to(dev).train()
teacher = bench.build_model("dino", 0.0).to(dev).train(); teacher.load_state_dict(model.state_dict())
for p in teacher.parameters(): p.requires_grad = False
crit = DINOLoss(65536, 10, 0.04, 0.07, 30, 300).to(dev)
opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.04, "dino"), lr=5e-4)
crops = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(2)] + [torch.randn(B, 3, 96, 96, device=dev) for _ in range(8)]
step = lambda: dino_train_step(model, teacher, crit, opt, crops, epoch=1, momentum=0.996, clip_grad_norm=3.0, freeze_last_layer=1, autocast_dtype=torch.bfloat16, ddp=None)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.key.startswith("aten::") and e.device_time_total > 20]
for k, c, t in sorted(rows, key=lambda r: -r[2])[:25]:
    print(f"{t:9.0f} us  x{c:4d}  {k}")
