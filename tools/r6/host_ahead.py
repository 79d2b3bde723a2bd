"""How far ahead of the GPU the host runs: K train steps enqueued back to back -- host wall time until the last step is ENQUEUED vs until the GPU is done.
host_ms / step well below gpu_ms / step = the GPU never waits for the host at a step boundary."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
import bench
from vtx.ddp import GradAllReduce
from vtx.optim import FusedAdamW
from vtx.train_step import MixLoss, make_param_groups, train_step
dev = torch.device("cuda:0")
for name, B, dp in (("swin_s", 128, 0.3), ("vit_s16", 256, 0.1), ("pvt_small", 128, 0.1), ("twins_svt_s", 128, 0.1)):
    torch.manual_seed(0)
    model = bench.build_model(name, dp).to(dev).train()
    opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
    ddp = GradAllReduce(model)
    x = torch.randn(B, 3, 224, 224, device=dev); l1 = torch.randint(0, 1000, (B,), device=dev)
    data = (x, l1, l1.roll(1), torch.rand(B, device=dev)); crit = MixLoss(0.1)
    for _ in range(8):
        train_step(model, crit, opt, data, clip_grad_norm=5.0, autocast_dtype=torch.bfloat16, ddp=ddp)
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for _ in range(K):
        train_step(model, crit, opt, data, clip_grad_norm=5.0, autocast_dtype=torch.bfloat16, ddp=ddp)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:12s} host enqueue {1e3 * (t1 - t0) / K:7.3f} ms/step | GPU {1e3 * (t2 - t0) / K:7.3f} ms/step | host finished {1e3 * (t2 - t1):7.1f} ms before the GPU")
    del model, opt, ddp
    torch.cuda.empty_cache()
