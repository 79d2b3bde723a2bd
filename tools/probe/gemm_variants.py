"""Time the LDS-DMA GEMM on the Swin stage-3 / ViT-S/16 shapes under dispatch-switch variants given as NAME=VALUE[,NAME=VALUE]
arguments (each set vs the default); per-step totals with the layer multiplicities of the models."""
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
CASES = []
for tag, M, C, layers in (("s3", 25088, 384, 18), ("vit", 50432, 384, 12), ("s2", 100352, 192, 2), ("s4", 6272, 768, 2)):
    CASES += [(f"{tag} qkv fwd", M, 3 * C, C, "bias", layers), (f"{tag} proj fwd", M, C, C, "resid", layers),
              (f"{tag} fc1 fwd", M, 4 * C, C, "silu", layers), (f"{tag} fc2 fwd", M, C, 4 * C, "resid", layers),
              (f"{tag} fc2 dgrad", M, 4 * C, C, "dsilu", layers), (f"{tag} fc1 dgrad", M, C, 4 * C, "plain", layers),
              (f"{tag} proj dgrad", M, C, C, "plain", layers), (f"{tag} qkv dgrad", M, C, 3 * C, "plain", layers)]
variants = [{}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]
tot = {}
for name, M, N, K, kind, layers in CASES:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    b = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).bfloat16(); z = torch.randn(M, N, device=dev).bfloat16()
    kw = dict(silu=dict(bias=b, act=ops.ACT_SILU, want_aux=True), bias=dict(bias=b), resid=dict(bias=b, resid=res),
              dsilu=dict(act=ops.ACT_DSILU, aux_in=z), plain={})[kind]
    line = f"{name:15s} {M:6d}x{N:4d}x{K:4d}"
    for i, v in enumerate(variants):
        with options.override(**{k: int(val) for k, val in v.items()}):
            t = timeit(lambda: ops.gemm(x, w, 0, **kw))
        tot[(name.split()[0], i)] = tot.get((name.split()[0], i), 0.0) + t * layers / 1e3
        line += f" {t:7.1f}"
    print(line)
print("variants:", variants)
for tag in ("s2", "s3", "s4", "vit"):
    print(tag, "per-step ms:", [round(tot[(tag, i)], 3) for i in range(len(variants))])
