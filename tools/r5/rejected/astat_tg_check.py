"""Two-group variant of the A-stationary GEMM (csrc/gemm_astat.hip, option ASTAT_TG) against the un-grouped kernel: bitwise + time, on the
epilogue kinds it exists for (plain / bias, SiLU forward with z) at the Swin-S stage-2 / 3 and ViT-S/16 shapes, plain and row-mapped."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops, options

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
NSET = 6


def run(d, kind):
    if kind == "bias":
        return (ops.gemm(d["x"], d["w"], 0, bias=d["bias"]),)
    if kind == "silu":
        return ops.gemm(d["x"], d["w"], 0, bias=d["bias"], act=ops.ACT_SILU, want_aux=True)
    return (ops.gemm(d["x"], d["w"], 0),)


def timeit(sets, kind, iters=30):
    for d in sets:
        run(d, kind)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        run(sets[i % NSET], kind)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


quick = "--quick" in sys.argv
shapes = [("stage-3 qkv fwd", 25088, 1152, 384, "bias"), ("stage-3 fc1 fwd", 25088, 1536, 384, "silu"), ("stage-3 proj dgrad", 25088, 384, 384, "plain"),
          ("stage-3 fc1 fwd, 21 364 rows", 21364, 1536, 384, "silu"), ("stage-2 fc1 fwd", 100352, 768, 192, "silu"), ("stage-2 qkv fwd", 100352, 576, 192, "bias"),
          ("ViT qkv fwd", 50432, 1152, 384, "bias"), ("ViT fc1 fwd", 50432, 1536, 384, "silu"), ("K 256", 25088, 1024, 256, "silu"), ("K 320", 25088, 1280, 320, "bias")]
if quick:
    shapes = shapes[:3]
for name, M, N, K, kind in shapes:
    sets = [dict(x=rn(M, K).bfloat16(), w=(rn(N, K) * 0.05).bfloat16(), bias=rn(N)) for _ in range(NSET)]
    with options.override(GEMM_ASTAT=2, ASTAT_TG=0):
        ref = run(sets[0], kind)
        t0 = timeit(sets, kind)
    with options.override(GEMM_ASTAT=2, ASTAT_TG=1):
        got = run(sets[0], kind)
        torch.cuda.synchronize()
        t1 = timeit(sets, kind)
        reps = [run(sets[1], kind) for _ in range(5)]
    same = all(torch.equal(r, o) for r, o in zip(ref, got)) and all(torch.equal(a, b) for r in reps[1:] for a, b in zip(reps[0], r))
    print(f"{name:30s} M {M:6d} N {N:5d} K {K:4d} {kind:5s}: lockstep {t0:7.1f} us  two groups {t1:7.1f} us  ({t0 / t1:4.2f}x)  {2.0 * M * N * K / t1 / 1e6:6.1f} TFLOP/s   "
          f"bitwise equal: {same}", flush=True)
    del sets
