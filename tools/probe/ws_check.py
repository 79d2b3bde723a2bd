"""A GEMM variant selected by an option (default GEMM_WS=1; e.g. `ws_check.py GLDS_EPI=1`) vs the shipped LDS-DMA GEMM: bitwise
equality of all fused epilogues."""
import sys
sys.path.insert(0, "/root/repo/vision-transformers-pytorch_amd")
import torch
from vtx import ops, options
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
OPT = dict([(sys.argv[1].split("=")[0], int(sys.argv[1].split("=")[1]))]) if len(sys.argv) > 1 else {"GEMM_WS": 1}
ok = True
for M, N, K, T in ((25088, 1536, 384, 196), (50432, 1536, 384, 197), (50432, 384, 1536, 197), (100352, 768, 192, 784), (25088, 1152, 384, 196), (16500, 1280, 64, 100), (6272, 768, 768, 49), (3000, 384, 1536, 100), (130, 128, 64, 10)):
    x = torch.randn(M, K, device=dev, generator=g).bfloat16(); w = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    b = torch.randn(N, device=dev, generator=g); res = torch.randn(M, N, device=dev, generator=g).bfloat16()
    keep = (torch.rand(M // T, device=dev, generator=g) < 0.8).float() / 0.8
    def run():
        h, z = ops.gemm(x, w, 0, bias=b, act=ops.ACT_SILU, want_aux=True)
        dz = ops.gemm(x, w, 0, act=ops.ACT_DSILU, aux_in=z, rowscale=keep, rows_per_scale=T)
        y = ops.gemm(x, w, 0, bias=b, resid=res, rowscale=keep, rows_per_scale=T)
        p = ops.gemm(x, w, 0)
        return h, z, dz, y, p
    base = run()
    with options.override(**OPT):
        got = run()
        got2 = run()
    torch.cuda.synchronize()
    same = all(torch.equal(a, c) for a, c in zip(base, got)) and all(torch.equal(a, c) for a, c in zip(base, got2))
    print(M, N, K, "tiles", (N // 128) * ((M + 127) // 128), "bitwise equal:", same)
    ok = ok and same
print("ALL OK" if ok else "MISMATCH")
