"""Per-workgroup phase stamps of the weight-gradient kernel (needs the stamp patch; not a product path)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops
dev = torch.device("cuda")
F = 2100.0   # s_memtime ticks per us (shader clock, approximate)
for name, M, N, K in (("s3 qkv", 25088, 1152, 384), ("s3 proj", 25088, 384, 384), ("s3 fc1", 25088, 1536, 384),
                      ("s4 fc2", 6272, 768, 3072), ("vit fc1", 50432, 1536, 384)):
    x = torch.randn(M, K, device=dev).bfloat16(); dy = torch.randn(M, N, device=dev).bfloat16()
    nblk = 1024
    st = torch.zeros(nblk * 6, dtype=torch.int64, device=dev)
    for _ in range(3): ops.wgrad(dy, x)
    torch.cuda.synchronize()
    os.environ["VTX_STAMP_PTR"] = str(st.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.wgrad(dy, x)
    e1.record()
    torch.cuda.synchronize()
    print(f"   wall per call {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    del os.environ["VTX_STAMP_PTR"]
    s = st.view(nblk, 6).cpu().double()
    s = s[s[:, 0] > 0]
    ph = [(s[:, i + 1] - s[:, i]).mean().item() / F for i in range(4)]
    nkt = s[:, 5].mean().item()
    print(f"{name}: blocks {s.shape[0]} k-tiles/block {nkt:.1f} | first tile {ph[0]:.2f} us  main loop {ph[1]:.2f} ({ph[1] / max(nkt, 1) * F:.0f} cyc/k-tile)"
          f"  epilogue {ph[2]:.2f}  store drain {ph[3]:.2f}  lifetime {(s[:, 4] - s[:, 0]).mean().item() / F:.2f} us (max {(s[:, 4] - s[:, 0]).max().item() / F:.2f})")
