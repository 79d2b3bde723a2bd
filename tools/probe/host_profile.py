"""cProfile of the host side of a train step (tiny batch: the GPU is never the limiter).  VTX_HOST_MODELS=pvt_small,..."""
import cProfile, os, pstats, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
import bench
from vtx.optim import FusedAdamW
from vtx.train_step import MixLoss, make_param_groups, train_step
dev = torch.device("cuda")
for name in [m for m in os.environ.get("VTX_HOST_MODELS", "pvt_small").split(",") if m]:
    model = bench.build_model(name, 0.1).to(dev).train()
    opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
    x = torch.randn(2, 3, 224, 224, device=dev); l1 = torch.randint(0, 1000, (2,), device=dev)
    data = (x, l1, l1.roll(1), torch.rand(2, device=dev))
    crit = MixLoss(0.1)
    for _ in range(5):
        train_step(model, crit, opt, data)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    with torch.autograd.set_multithreading_enabled(False):       # the backward's Python runs on this thread: visible to cProfile
        for _ in range(20):
            train_step(model, crit, opt, data)
    torch.cuda.synchronize()
    pr.disable()
    print(f"==== {name}: 20 steps")
    st = pstats.Stats(pr)
    st.sort_stats(os.environ.get("VTX_PROF_SORT", "tottime")).print_stats(int(os.environ.get("VTX_PROF_TOP", "28")))
