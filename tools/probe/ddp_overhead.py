"""What the gradient all-reduce machinery costs per Swin-S step when there is nothing to reduce: a 1-rank RCCL group with
GradAllReduce(force=True) (hooks, bucket sinks, side-stream joins, one all_reduce per bucket, finish) vs no DDP object."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch, torch.distributed as dist
import bench
from vtx.ddp import GradAllReduce
from vtx.optim import FusedAdamW
from vtx.train_step import MixLoss, make_param_groups, train_step
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B = 128
x = torch.randn(B, 3, 224, 224, device=dev); l1 = torch.randint(0, 1000, (B,), device=dev)
data = (x, l1, l1.roll(1), torch.rand(B, device=dev))
crit = MixLoss(0.1)
for use in (False, True, False, True):
    torch.manual_seed(0)
    model = bench.build_model("swin_s", 0.3).to(dev).train()
    opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
    ddp = GradAllReduce(model, force=True) if use else None
    for _ in range(5):
        train_step(model, crit, opt, data, clip_grad_norm=5.0, ddp=ddp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        train_step(model, crit, opt, data, clip_grad_norm=5.0, ddp=ddp)
    torch.cuda.synchronize()
    print(f"ddp machinery {'ON ' if use else 'off'}: {1e3 * (time.perf_counter() - t0) / 20:.2f} ms/step"
          + (f"  ({len(ddp.buckets)} buckets)" if use else ""))
    if ddp: ddp.remove()
    del model, opt, ddp
dist.destroy_process_group()
