"""Two seeded runs of one full-size Swin-S train step: are the losses / logits bitwise equal?  (round 4: the A-stationary GEMM)
   python tools/r4/det_check.py [--model swin_s] [--trials 3]"""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
import bench
from vtx.optim import FusedAdamW
from vtx.train_step import MixLoss, make_param_groups, train_step

ap = argparse.ArgumentParser(); ap.add_argument("--model", default="swin_s"); ap.add_argument("--trials", type=int, default=3)
ap.add_argument("--fwd-only", action="store_true")
a = ap.parse_args()
d = torch.device("cuda")
B = bench.default_batch(a.model)
dp = 0.3 if a.model == "swin_s" else 0.1
g = torch.Generator(device=d).manual_seed(77)
x = torch.randn(B, 3, 224, 224, device=d, generator=g)
l1 = torch.randint(0, 1000, (B,), device=d, generator=g)
data = (x, l1, l1.roll(1), torch.rand(B, device=d, generator=g))


def run():
    torch.manual_seed(5)
    model = bench.build_model(a.model, dp).to(d).train()
    opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
    torch.manual_seed(6)
    if a.fwd_only:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return model, [model(x).float().sum().item()]
    return model, [train_step(model, MixLoss(0.1), opt, data, clip_grad_norm=5.0).item() for _ in range(2)]


ref_m, ref_l = run()
bad = 0
for t in range(a.trials):
    m, l = run()
    same = l == ref_l and all(torch.equal(p, q) for p, q in zip(m.parameters(), ref_m.parameters()))
    bad += not same
    print(f"trial {t}: losses {l} vs {ref_l}: {'same' if same else 'DIFFERENT'}", flush=True)
print(f"{bad} of {a.trials} trials differ")
