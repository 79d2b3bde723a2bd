"""A-stationary GEMM (csrc/gemm_astat.hip) vs the tiled LDS-DMA kernels: bitwise comparison and cold-ish timing on the
Swin-S stage-3 / ViT-S/16 shapes with K <= 384.      python tools/r4/astat_check.py [--vit] [--iters 30]
Timing: 6 rotating buffer sets (outputs of one launch are not the next one's inputs; ~1 GB of traffic between two uses of a
buffer, beyond the 256-MB memory-side cache)."""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops, options

ap = argparse.ArgumentParser()
ap.add_argument("--vit", action="store_true")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--rows", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda")
M = a.rows or (256 * 197 if a.vit else 128 * 196)
C = 384
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
NSET = 6


def mk(N, K, kind):
    sets = []
    for _ in range(NSET):
        x = rn(M, K).bfloat16()
        d = dict(x=x, w=(rn(N, K) * 0.05).bfloat16(), bias=rn(N))
        if kind in ("resid", "resid_dp"):
            d["resid"] = rn(M, N).bfloat16()
        if kind == "resid_dp":
            d["rowscale"] = (torch.rand(M // 196 if M % 196 == 0 else M // 197, device=dev, generator=g) > 0.3).float() / 0.7
        if kind == "dsilu":
            d["z"] = rn(M, N).bfloat16()
        sets.append(d)
    return sets


def run(d, kind, T):
    if kind == "bias":
        return (ops.gemm(d["x"], d["w"], 0, bias=d["bias"]),)
    if kind == "silu":
        return ops.gemm(d["x"], d["w"], 0, bias=d["bias"], act=ops.ACT_SILU, want_aux=True)
    if kind == "resid":
        return (ops.gemm(d["x"], d["w"], 0, bias=d["bias"], resid=d["resid"]),)
    if kind == "resid_dp":
        return (ops.gemm(d["x"], d["w"], 0, bias=d["bias"], resid=d["resid"], rowscale=d["rowscale"], rows_per_scale=T),)
    if kind == "dsilu":
        return (ops.gemm(d["x"], d["w"], 0, act=ops.ACT_DSILU, aux_in=d["z"]),)
    if kind == "plain":
        return (ops.gemm(d["x"], d["w"], 0),)


def timeit(sets, kind, T, iters):
    for d in sets:
        run(d, kind, T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        run(sets[i % NSET], kind, T)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


T = 197 if (a.vit or M % 196) else 196
shapes = [("qkv fwd", 3 * C, C, "bias"), ("fc1 fwd", 4 * C, C, "silu"), ("proj fwd", C, C, "resid_dp"), ("proj dgrad", C, C, "plain"),
          ("fc2 dgrad", 4 * C, C, "dsilu"), ("stage-2 qkv (K 192)", 576 + 192, 192, "bias"), ("K 128", 512, 128, "resid")]
print(f"M = {M}")
for name, N, K, kind in shapes:
    sets = mk(N, K, kind)
    with options.override(GEMM_ASTAT=0):
        ref = run(sets[0], kind, T)
        t0 = timeit(sets, kind, T, a.iters)
    with options.override(GEMM_ASTAT=2):
        import time
        torch.cuda.synchronize(); w0 = time.time()
        got = run(sets[0], kind, T)
        torch.cuda.synchronize()
        if time.time() - w0 > 1.0:
            print(f"{name}: a-stationary launch took {time.time() - w0:.1f} s -- protocol hang (bounded spin); skipped", flush=True)
            continue
        t1 = timeit(sets, kind, T, a.iters)
    torch.cuda.synchronize()
    same = all(torch.equal(r, o) for r, o in zip(ref, got))
    md = max(float((r.float() - o.float()).abs().max()) for r, o in zip(ref, got))
    nb = 2.0 * (M * K + N * K + M * N * (1 + (kind in ("resid", "resid_dp", "dsilu")) + (kind == "silu")))
    print(f"{name:22s} N {N:5d} K {K:4d} {kind:9s}: tiled {t0:7.1f} us  a-stationary {t1:7.1f} us  ({t0 / t1:4.2f}x)  "
          f"{nb / t1 / 1e6:5.2f} TB/s algorithmic, {2.0 * M * N * K / t1 / 1e6:6.1f} TFLOP/s   bitwise equal: {same} (max |d| {md:.2e})", flush=True)
    del sets
