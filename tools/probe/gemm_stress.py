"""Stress of the default LDS-DMA GEMM path (wave-private epilogue, deferred vector loads, full-ring prologue): random shapes with N a
multiple of 128 and K a multiple of 64 (1 .. 24 k-tiles), ragged and large M, every epilogue combination -- bitwise against the
shared-epilogue kernel (GLDS_EPI=0) and, for the small ones, against fp64.   python tools/probe/gemm_stress.py [cases] [seed]"""
import random, sys
sys.path.insert(0, "/root/repo/vision-transformers-pytorch_amd")
import torch
from vtx import ops, options
dev = torch.device("cuda")
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for ci in range(ncase):
    g = torch.Generator(device=dev).manual_seed(rng.randrange(1 << 30))
    T = rng.choice([1, 7, 49, 64, 196, 197])
    B = rng.choice([1, 2, 3, 17, 64, 128, 256])
    M = B * T
    N = 128 * rng.randint(1, 12)
    K = 64 * rng.choice([1, 2, 3, 4, 6, 12, 24])
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    kw = {}
    if rng.random() < 0.7: kw["bias"] = torch.randn(N, device=dev, generator=g)
    act = rng.choice([None, None, "silu", "gelu", "dsilu", "dgelu"])
    if act in ("silu", "gelu"):
        kw.update(act=ops.ACT_SILU if act == "silu" else ops.ACT_GELU, want_aux=rng.random() < 0.8)
    elif act in ("dsilu", "dgelu"):
        kw.update(act=ops.ACT_DSILU if act == "dsilu" else ops.ACT_DGELU, aux_in=torch.randn(M, N, device=dev, generator=g).bfloat16())
    if act not in ("silu", "gelu"):
        if rng.random() < 0.5: kw.update(rowscale=(torch.rand(B, device=dev, generator=g) > 0.3).float() / 0.7, rows_per_scale=T)
        if rng.random() < 0.6: kw["resid"] = torch.randn(M, N, device=dev, generator=g).bfloat16()
    for bm in (0, 64, 128):
        with options.override(GLDS_BM=bm):
            got = ops.gemm(x, w, 0, **kw)
            with options.override(GLDS_EPI=0):
                ref = ops.gemm(x, w, 0, **kw)
        got = got if isinstance(got, tuple) else (got,)
        ref = ref if isinstance(ref, tuple) else (ref,)
        ok = all(torch.equal(a, b) for a, b in zip(got, ref))
        if not ok:
            bad += 1
            print(f"MISMATCH case {ci}: M={M} N={N} K={K} BM={bm} kw={ {k: (v.shape if torch.is_tensor(v) else v) for k, v in kw.items()} }")
    if M * N * K < 2e9 and act is None:
        r = x.double() @ w.double().t()
        if "bias" in kw: r = r + kw["bias"].double()
        if "rowscale" in kw: r = r * kw["rowscale"].double().repeat_interleave(T)[:, None]
        if "resid" in kw: r = r + kw["resid"].double()
        err = ((got[0].double() - r).abs().max() / r.abs().max().clamp_min(1e-9)).item()
        if err > 1.5e-2:
            bad += 1
            print(f"ERROR case {ci}: M={M} N={N} K={K} rel err {err:.3e}")
torch.cuda.synchronize()
print(f"{ncase} cases x 3 tile heights: {'ALL OK' if bad == 0 else str(bad) + ' FAILURES'}")
