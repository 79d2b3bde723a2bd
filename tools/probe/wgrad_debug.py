import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "vision-transformers-pytorch_amd"))
import torch
from vtx import ops
d = torch.device("cuda")
M, N, K = 64, 128, 128
for t in (0, 1, 2, 4, 5, 17, 63):
    dy0 = torch.zeros(M, N).bfloat16(); x1 = torch.ones(M, K).bfloat16()
    dy0[t, 3] = 1
    dW, db = ops.wgrad(dy0.to(d), x1.to(d))
    print("A one-hot token", t, ": row3 sum", dW[3].sum().item(), "(expect 128) total", dW.sum().item(), "db[3]", db[3].item(), "db sum", db.sum().item())
    dy1 = torch.ones(M, N).bfloat16(); x0 = torch.zeros(M, K).bfloat16()
    x0[t, 5] = 1
    dW, db = ops.wgrad(dy1.to(d), x0.to(d))
    print("B one-hot token", t, ": col5 sum", dW[:, 5].sum().item(), "(expect 128) total", dW.sum().item())
