"""Do the kernels address tensors with more than 2^31 elements correctly?  (6 M rows x 384 bf16 = 2.3 G elements)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops
dev = torch.device("cuda")
rows, C = 6_000_000, 384
x = torch.randn(rows, C, device=dev, dtype=torch.bfloat16)
g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
for r in (0, 3_000_000, rows - 1):
    ref = torch.nn.functional.layer_norm(x[r].float(), (C,), eps=1e-6)
    print("ln row", r, (y[r].float() - ref).abs().max().item())
w = (torch.randn(C, 64, device=dev) * 0.1).bfloat16()
a = torch.randn(rows, 64, device=dev, dtype=torch.bfloat16)
c = ops.gemm(a, w, 0)
for r in (0, 2_999_999, 5_592_406, rows - 1):          # 5 592 406 * 384 > 2^31
    ref = a[r].float() @ w.float().t()
    print("gemm row", r, ((c[r].float() - ref).abs().max() / ref.abs().max()).item())
dW, db = ops.wgrad(c, a)
refW = torch.zeros(C, 64, device=dev, dtype=torch.float32)
for i in range(0, rows, 500_000):
    refW += c[i:i + 500_000].float().t() @ a[i:i + 500_000].float()
print("wgrad rel", ((dW - refW).norm() / refW.norm()).item())
