"""Does splitting a 2.3-round launch of 128 x 128 tiles into two full rounds + a tail on 64-row tiles pay?  (ViT-S/16 long-K N = 384 GEMMs)
   python tools/r4/tail_split_check.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops, options
dev = torch.device("cuda"); BF = torch.bfloat16

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

T = 197
for name, B, K, N, epi in (("fc2 fwd", 256, 1536, 384, True), ("fc1 dgrad", 256, 1536, 384, False), ("qkv dgrad", 256, 1152, 384, False)):
    g = torch.Generator(device="cpu").manual_seed(3)
    M = B * T
    # rotate over 3 operand sets (> 256 MB: the Infinity Cache does not hold them)
    sets = []
    for _ in range(3):
        a = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
        bias = torch.randn(N, generator=g).to(dev) if epi else None
        resid = torch.randn(M, N, generator=g).to(BF).to(dev) if epi else None
        rs = ((torch.rand(B, generator=g) < 0.9).float() / 0.9).to(dev) if epi else None
        sets.append((a, w, bias, resid, rs, torch.empty(M, N, dtype=BF, device=dev)))
    it = [0]
    def full():
        a, w, bias, resid, rs, out = sets[it[0] % 3]; it[0] += 1
        ops.gemm(a, w, 0, bias=bias, resid=resid, rowscale=rs, rows_per_scale=T, out=out)
        return out
    res = {}
    t_full = timeit(full)
    ref = full().clone()
    for bs in (221, 216, 208, 192, 170):
        r0 = bs * T
        def split():
            a, w, bias, resid, rs, out = sets[it[0] % 3]; it[0] += 1
            ops.gemm(a[:r0], w, 0, bias=bias, resid=None if resid is None else resid[:r0], rowscale=None if rs is None else rs[:bs], rows_per_scale=T, out=out[:r0])
            ops.gemm(a[r0:], w, 0, bias=bias, resid=None if resid is None else resid[r0:], rowscale=None if rs is None else rs[bs:], rows_per_scale=T, out=out[r0:])
            return out
        t = timeit(split)
        it[0] = 3 * ((it[0] + 2) // 3) + (it[0] - 1) % 3 if False else it[0]
        res[bs] = t
    # value check on set 0
    it[0] = 0; ref = full().clone(); it[0] = 0
    r0 = 221 * T
    a, w, bias, resid, rs, out = sets[0]
    o2 = torch.empty_like(out)
    ops.gemm(a[:r0], w, 0, bias=bias, resid=None if resid is None else resid[:r0], rowscale=None if rs is None else rs[:221], rows_per_scale=T, out=o2[:r0])
    ops.gemm(a[r0:], w, 0, bias=bias, resid=None if resid is None else resid[r0:], rowscale=None if rs is None else rs[221:], rows_per_scale=T, out=o2[r0:])
    print(f"{name:10s} K {K}: one launch {t_full:6.1f} us | split at sample " + "  ".join(f"{b}: {t:6.1f}" for b, t in res.items()) + f" | bitwise {torch.equal(ref, o2)}", flush=True)
