"""Host (Python + launch) time per train step vs GPU time per step: is the stream ever starved?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
import bench
from vtx.optim import FusedAdamW
from vtx.train_step import MixLoss, make_param_groups, train_step
dev = torch.device("cuda")
CASES = (("swin_s", 128, 0.3), ("vit_s16", 256, 0.1), ("pvt_small", 128, 0.1), ("twins_svt_s", 128, 0.1))
ONLY = [m for m in os.environ.get("VTX_HOST_MODELS", "").split(",") if m]           # e.g. VTX_HOST_MODELS=pvt_small,twins_svt_s
for name, B, dp in [c for c in CASES if not ONLY or c[0] in ONLY]:
    model = bench.build_model(name, dp).to(dev).train()
    opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
    x = torch.randn(B, 3, 224, 224, device=dev); l1 = torch.randint(0, 1000, (B,), device=dev)
    data = (x, l1, l1.roll(1), torch.rand(B, device=dev))
    crit = MixLoss(0.1)
    for _ in range(5):
        train_step(model, crit, opt, data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        train_step(model, crit, opt, data)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e3 * (t1 - t0) / 20:.2f} ms/step, wall {1e3 * (t2 - t0) / 20:.2f} ms/step")
    # pure host cost: the first steps after a synchronise run ahead of the GPU (nothing to wait for yet)
    ahead = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        train_step(model, crit, opt, data)
        train_step(model, crit, opt, data)
        ahead.append((time.perf_counter() - t0) / 2)
    torch.cuda.synchronize()
    print(f"{name}: run-ahead enqueue (2 steps after a synchronise, GPU not yet the limiter) {1e3 * min(ahead):.2f} ms/step")
    # host-only cost: how long does the python side take when the GPU is not the limiter?  (tiny batch)
    xs = torch.randn(2, 3, 224, 224, device=dev); ls = torch.randint(0, 1000, (2,), device=dev)
    ds = (xs, ls, ls.roll(1), torch.rand(2, device=dev))
    for _ in range(3):
        train_step(model, crit, opt, ds)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        train_step(model, crit, opt, ds)
    torch.cuda.synchronize()
    print(f"{name}: batch-2 step (host-bound) {1e3 * (time.perf_counter() - t0) / 20:.2f} ms/step")
    del model, opt
