"""Does a HIGH-priority main stream (the side stream of the weight gradients stays at default priority) let the dgrad chain run
ahead of the grouped weight-gradient launches?  ms per Swin-S / ViT-S/16 step, default stream vs a priority -1 stream."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
import bench
from vtx.optim import FusedAdamW
from vtx.train_step import MixLoss, make_param_groups, train_step
dev = torch.device("cuda")
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
for name in ("swin_s", "vit_s16"):
    B = bench.default_batch(name)
    torch.manual_seed(0)
    model = bench.build_model(name, 0.3 if name == "swin_s" else 0.1).to(dev).train()
    opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
    data = (torch.randn(B, 3, 224, 224, device=dev), torch.randint(0, 1000, (B,), device=dev),
            torch.randint(0, 1000, (B,), device=dev), torch.rand(B, device=dev))
    crit = MixLoss(0.1)

    def run(stream, n=30):
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for _ in range(5):
                train_step(model, crit, opt, data)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                train_step(model, crit, opt, data)
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    hp = torch.cuda.Stream(device=dev, priority=-1)
    for rep in range(2):
        print(f"{name}: default stream {run(None):.3f} ms/step   high-priority main stream {run(hp):.3f} ms/step")
    del model, opt
