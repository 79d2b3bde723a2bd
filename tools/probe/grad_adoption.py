"""Does autograd ADOPT the gradients the side stream produced (param.grad is the tensor backward returned), or clone them
(a copy enqueued on the main stream, not ordered after the side-stream kernel that writes the source)?"""
import os, sys
os.environ["VTX_DEBUG_SIDE"] = "p"
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
import bench
from vtx import functional as VF
from vtx.train_step import MixLoss
dev = torch.device("cuda")
for name, B in (("swin_s", 32), ("vit_s16", 32)):
    model = bench.build_model(name, 0.1).to(dev).train()
    x = torch.randn(B, 3, 224, 224, device=dev); l1 = torch.randint(0, 1000, (B,), device=dev)
    crit = MixLoss(0.1)
    for it in range(3):
        VF._debug_sums.clear()
        for p in model.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = crit(model(x), l1, l1.roll(1), torch.rand(B, device=dev))
        with VF.deferred_wgrad(True):
            loss.backward()
        torch.cuda.synchronize()
        ptrs = set(VF._debug_sums)
        layer = [(n, p) for n, p in model.named_parameters() if ("block" in n or "layers" in n) and p.grad is not None]
        adopted = [n for n, p in layer if p.grad.data_ptr() in ptrs]
        cloned = [n for n, p in layer if p.grad.data_ptr() not in ptrs]
        print(f"{name} it {it} (VTX_LAYER_CALL={os.environ.get('VTX_LAYER_CALL', '1')}): recorded {len(ptrs)} side-stream gradient tensors; layer parameters "
              f"adopted {len(adopted)}, NOT adopted {len(cloned)}: {cloned[:8]}")
