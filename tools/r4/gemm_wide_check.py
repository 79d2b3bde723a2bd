"""Full-width stream-K GEMM (csrc/gemm_wide.hip, option GEMM_WIDE) vs the 128 x 128 LDS-DMA kernel: values and us per launch.
   python tools/r4/gemm_wide_check.py [--case substr]"""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops, options

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="")
ap.add_argument("--modes", default="012")
ARGS = ap.parse_args()
SKIP = bool(os.environ.get("VTX_CHECK_SKIP"))
dev = torch.device("cuda")
BF = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


CASES = (("vit fc2 fwd resid+droppath", 256, 197, 1536, 384, True, True), ("vit fc1 dgrad", 256, 197, 1536, 384, False, False),
         ("vit qkv dgrad", 256, 197, 1152, 384, False, False), ("swin s3 fc2 fwd resid", 128, 196, 1536, 384, True, True),
         ("swin s3 qkv dgrad", 128, 196, 1152, 384, False, False), ("swin s3 kept 100 fc1 dgrad", 100, 196, 1536, 384, False, False),
         ("ragged rows 12345 fc2 fwd", 1, 12345, 1536, 384, True, False), ("dino local b640 t36 fc2", 640, 36, 1536, 384, True, True))
for name, B, T, K, N, epi, dp in CASES:
    if ARGS.case not in name:
        continue
    g = torch.Generator(device="cpu").manual_seed(11)
    M = B * T
    a = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
    bias = torch.randn(N, generator=g).to(dev) if epi else None
    resid = torch.randn(M, N, generator=g).to(BF).to(dev) if epi else None
    rs = ((torch.rand(B, generator=g) < 0.8).float() / 0.8).to(dev) if dp else None
    out, tm = {}, {}
    run = lambda: ops.gemm(a, w, 0, bias=bias, resid=resid, rowscale=rs, rows_per_scale=T)
    for mode in [int(c) for c in ARGS.modes]:
        with options.override(GEMM_WIDE=mode):
            out[mode] = run()
            again = run()
            assert SKIP or torch.equal(out[mode], again), f"{name}: GEMM_WIDE={mode} not deterministic"
            tm[mode] = timeit(run)
    msg = f"{name:30s} M {M:6d} K {K:4d}: " + "  ".join(f"mode {m}: {tm[m]:6.1f} us" for m in tm)
    if 0 in out:
        ref = out[0].float()
        for m in out:
            if m:
                d = (out[m].float() - ref).abs().max().item()
                msg += f"   mode {m} vs tiled: {'bitwise' if torch.equal(out[m], out[0]) else f'max diff {d:.3e} (values ~{ref.abs().max().item():.1f})'}"
                assert SKIP or d <= 0.07 * max(1.0, ref.abs().max().item() / 8), f"{name}: mode {m} differs by {d}"
    fl = 2.0 * M * N * K
    msg += f"   best {fl / min(tm.values()) / 1e6:5.0f} TFLOP/s"
    print(msg, flush=True)
