"""Per-workgroup phase stamps of the LDS-DMA GEMM (needs the stamp patch of gemm_glds.hip; not a product path)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops
dev = torch.device("cuda")
for name, M, N, K, act, res in (("s2 fc1 fwd", 100352, 768, 192, 1, 0), ("s2 qkv fwd", 100352, 576, 192, 0, 0),
                                ("s3 qkv fwd", 25088, 1152, 384, 0, 0), ("s3 fc1 fwd", 25088, 1536, 384, 1, 0),
                                ("s3 fc2 fwd", 25088, 384, 1536, 0, 1), ("s4 fc2 fwd", 6272, 768, 3072, 0, 1)):
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev).bfloat16() if res else None
    bn = 128 if N % 128 == 0 else 96
    nblk = ((M + 63) // 64) * ((N + bn - 1) // bn)
    st = torch.zeros(nblk * 6, dtype=torch.int64, device=dev)
    kw = dict(bias=bias, resid=r, act=act, want_aux=bool(act))
    for _ in range(3): ops.gemm(x, w, **kw)
    torch.cuda.synchronize()
    os.environ["VTX_STAMP_PTR"] = str(st.data_ptr())
    ops.gemm(x, w, **kw)
    torch.cuda.synchronize()
    del os.environ["VTX_STAMP_PTR"]
    s = st.view(nblk, 6).cpu().double()
    t0 = s[:, 0].min()
    f = 100.0  # s_memtime ticks at 100 MHz -> 10 ns units
    span = (s[:, 4].max() - t0) / f
    ph = [(s[:, i + 1] - s[:, i]).mean().item() / f for i in range(4)]
    life = (s[:, 4] - s[:, 0]).mean().item() / f
    # concurrency: sum of lifetimes / span
    conc = (s[:, 4] - s[:, 0]).sum().item() / f / span
    start_sorted = torch.sort(s[:, 0] - t0)[0] / f
    print(f"{name}: blocks {nblk} span {span:.1f} us | mean: first-tile wait {ph[0]:.2f}  main loop {ph[1]:.2f}  epilogue {ph[2]:.2f}  store drain {ph[3]:.2f}  lifetime {life:.2f} us | avg resident {conc:.0f}"
          f" | start times: 10% {start_sorted[nblk // 10]:.1f} 50% {start_sorted[nblk // 2]:.1f} 90% {start_sorted[nblk * 9 // 10]:.1f}")
