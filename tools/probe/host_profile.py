"""cProfile of the host side of one train step (batch 2: the GPU is never the limiter): where do the ~15 ms of Python + launch time
per Swin-S step go?   python tools/probe/host_profile.py [swin_s|vit_s16|pvt_small] [N lines]"""
import cProfile, os, pstats, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
import bench
from vtx.optim import FusedAdamW
from vtx.train_step import MixLoss, make_param_groups, train_step
name = sys.argv[1] if len(sys.argv) > 1 else "swin_s"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 45
dev = torch.device("cuda")
model = bench.build_model(name, 0.1).to(dev).train()
opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
xs = torch.randn(2, 3, 224, 224, device=dev); ls = torch.randint(0, 1000, (2,), device=dev)
ds = (xs, ls, ls.roll(1), torch.rand(2, device=dev))
crit = MixLoss(0.1)
for _ in range(5):
    train_step(model, crit, opt, ds)
torch.cuda.synchronize()
# the autograd engine runs backward functions on its own (C++) thread, which cProfile does not see: keep it on this thread
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    train_step(model, crit, opt, ds)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(nl)
