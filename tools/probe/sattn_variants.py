"""ViT attention fast path (csrc/attention_seq.hip) under option SATTN_WAVES = 4 | 8: bitwise equality and time per launch on the
ViT-S/16 (B = 256, L = 197), DINO global (B = 128) and DINO local-crop (L = 37, B = 512) shapes."""
import sys
sys.path.insert(0, "/root/repo/vision-transformers-pytorch_amd")
import torch
from vtx import ops, options
dev = torch.device("cuda")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for B, L in ((256, 197), (128, 197), (512, 37), (64, 100)):
    nH, D = 6, 64
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.randn(B * L, 3 * nH * D, device=dev, generator=g).bfloat16()
    do = torch.randn(B * L, nH * D, device=dev, generator=g).bfloat16()
    res = {}
    for wv in (4, 8):
        with options.override(SATTN_WAVES=wv):
            o, lse = ops.attention_fwd(qkv, B, L, nH, D)
            dq = ops.attention_bwd(qkv, o, do, lse, B, L, nH, D)
            dq = dq[0] if isinstance(dq, tuple) else dq
            tf = timeit(lambda: ops.attention_fwd(qkv, B, L, nH, D))
            tb = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, L, nH, D))
        res[wv] = (o, lse, dq, tf, tb)
    same = all(torch.equal(a, b) for a, b in zip(res[4][:3], res[8][:3]))
    print(f"B={B} L={L}: fwd {res[4][3]:.1f} -> {res[8][3]:.1f} us, bwd {res[4][4]:.1f} -> {res[8][4]:.1f} us, bitwise equal: {same}")
