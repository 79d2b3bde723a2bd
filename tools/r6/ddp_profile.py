"""Which device operations the data-parallel machinery adds to a Swin-S step on ONE GPU (forced one-rank RCCL group): torch.profiler
over 3 steps of each arm, every device op that is more frequent or longer in the ON arm."""
import os, socket, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch, torch.distributed as dist
from torch.profiler import profile, ProfilerActivity
import bench
from vtx.ddp import GradAllReduce
from vtx.optim import FusedAdamW
from vtx.train_step import MixLoss, make_param_groups, train_step

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B = 128
x = torch.randn(B, 3, 224, 224, device=dev); l1 = torch.randint(0, 1000, (B,), device=dev)
data = (x, l1, l1.roll(1), torch.rand(B, device=dev))
crit = MixLoss(0.1)
tabs = {}
for use in (False, True):
    torch.manual_seed(0)
    model = bench.build_model("swin_s", 0.3).to(dev).train()
    opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
    ddp = GradAllReduce(model, force=use)
    for _ in range(6):
        train_step(model, crit, opt, data, clip_grad_norm=5.0, autocast_dtype=torch.bfloat16, ddp=ddp)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            train_step(model, crit, opt, data, clip_grad_norm=5.0, autocast_dtype=torch.bfloat16, ddp=ddp)
        torch.cuda.synchronize()
    tab = {}
    for e in prof.key_averages():
        dt = getattr(e, "device_time_total", None)
        if dt is None:
            dt = getattr(e, "cuda_time_total", 0.0)
        tab[e.key] = (e.count / 3.0, dt / 3.0, e.cpu_time_total / 3.0)
    tabs[use] = tab
    if use:
        ddp.remove()
    del model, opt, ddp
    torch.cuda.empty_cache()
print("op | calls/step off -> ON | device us/step off -> ON | host us/step off -> ON")
rows = []
for k in set(tabs[False]) | set(tabs[True]):
    a = tabs[False].get(k, (0, 0, 0)); b = tabs[True].get(k, (0, 0, 0))
    if abs(b[0] - a[0]) >= 0.5 or abs(b[1] - a[1]) > 5 or abs(b[2] - a[2]) > 20:
        rows.append((b[1] - a[1] + (b[2] - a[2]) * 0.0, k, a, b))
for _, k, a, b in sorted(rows, key=lambda r: -abs(r[0]))[:45]:
    print(f"{k[:90]:90s} | {a[0]:7.1f} -> {b[0]:7.1f} | {a[1]:9.1f} -> {b[1]:9.1f} | {a[2]:9.1f} -> {b[2]:9.1f}")
dist.destroy_process_group()
