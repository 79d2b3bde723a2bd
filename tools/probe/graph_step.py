"""Experiment: how much of a train step is inter-kernel launch gap?  Capture fwd + bwd + optimizer of one step in a
HIP graph (torch.cuda.CUDAGraph) and compare its replay time with the eager step.  Timing probe only: the optimizer's
step count / learning rate travel as kernel arguments and are frozen inside the graph."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
import bench
from vtx.train_step import MixLoss, make_param_groups, train_step
from vtx.optim import FusedAdamW
dev = torch.device("cuda")
name = sys.argv[1] if len(sys.argv) > 1 else "swin_s"
B = 256 if name == "vit_s16" else 128
torch.manual_seed(0)
model = bench.build_model(name, 0.3 if name == "swin_s" else 0.1).to(dev).train()
crit = MixLoss(eps=0.1)
opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
data = (torch.randn(B, 3, 224, 224, device=dev), torch.randint(0, 1000, (B,), device=dev),
        torch.randint(0, 1000, (B,), device=dev), torch.rand(B, device=dev))
step = lambda: train_step(model, crit, opt, data, clip_grad_norm=5.0, autocast_dtype=torch.bfloat16, ddp=None)


def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
print(f"{name}: eager {timeit(step):.3f} ms/step")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
print(f"{name}: graph replay {timeit(g.replay):.3f} ms/step   loss {loss.item():.4f}")
print(f"{name}: eager again {timeit(step):.3f} ms/step")
