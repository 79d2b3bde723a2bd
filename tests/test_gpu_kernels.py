"""-m gpu: each HIP kernel through the C ABI vs the CPU oracle on the same seeded inputs."""
import pytest
import torch

from gpu_util import TOL, check, dev
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _mk(shape, seed, dtype, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    t = (torch.randn(shape, generator=g) * scale).to(dtype)
    return t


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C", [(98, 96), (392, 192), (197 * 2, 384), (130, 768), (50, 1536), (7, 1024)])
def test_layernorm_fwd_bwd(dtype, rows, C):
    from vtx import ops
    d = dev()
    x = _mk((rows, C), 1, dtype, 2.0) + 0.3
    dy = _mk((rows, C), 2, dtype)
    dres = _mk((rows, C), 3, dtype)
    g = 1 + 0.1 * _mk((C,), 4, torch.float32)
    b = 0.1 * _mk((C,), 5, torch.float32)
    y, mean, rstd = ops.layernorm_fwd(x.to(d), g.to(d), b.to(d), 1e-6)
    dx, dg, db = ops.layernorm_bwd(dy.to(d), x.to(d), mean, rstd, g.to(d), dres=dres.to(d))
    xr = x.double().requires_grad_(True)
    gr = g.double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = R.layer_norm(xr, gr, br, 1e-6)
    dxr, dgr, dbr = torch.autograd.grad(yr, [xr, gr, br], dy.double())
    t = TOL[dtype]
    check(f"layernorm fwd {dtype} {rows}x{C}", y, yr, t["out"])
    check(f"layernorm dx {dtype} {rows}x{C}", dx, dxr + dres.double(), t["out"])
    check(f"layernorm dgamma {dtype} {rows}x{C}", dg, dgr, 1e-5 if dtype == torch.float32 else 2e-4)
    check(f"layernorm dbeta {dtype} {rows}x{C}", db, dbr, 1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_merge_gather(dtype):
    """PatchMerge's patchify(2) folded into the LN row addressing (swin_transformer.py:216-229)."""
    from vtx import ops
    d = dev()
    B, H, W, Cs = 2, 14, 14, 96
    x = _mk((B, H, W, Cs), 6, dtype)
    g = 1 + 0.1 * _mk((4 * Cs,), 7, torch.float32)
    b = 0.1 * _mk((4 * Cs,), 8, torch.float32)
    dy = _mk((B, H // 2, W // 2, 4 * Cs), 9, dtype)
    y, mean, rstd = ops.layernorm_fwd(x.to(d), g.to(d), b.to(d), 1e-5, merge_hw=(H, W))
    dx, dg, db = ops.layernorm_bwd(dy.to(d), x.to(d), mean, rstd, g.to(d), merge_hw=(H, W))
    xr = x.double().requires_grad_(True)
    yr = R.layer_norm(R.patchify(xr, 2), g.double(), b.double(), 1e-5)
    (dxr,) = torch.autograd.grad(yr, [xr], dy.double())
    t = TOL[dtype]
    check(f"merge-LN fwd {dtype}", y, yr, t["out"])
    check(f"merge-LN dx {dtype}", dx, dxr, t["out"])


GEMM_SHAPES = [  # (M, N, K)
    (256, 128, 64), (128, 96, 96), (98, 288, 96), (394, 1152, 384), (130, 384, 1536),
    (98, 2304, 768), (2, 1000, 768), (392, 192, 384), (6272, 96, 64),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_forward_plain(dtype, M, N, K):
    from vtx import ops
    d = dev()
    a = _mk((M, K), 11, dtype)
    w = _mk((N, K), 12, dtype, 0.1)
    bias = _mk((N,), 13, torch.float32)
    c = ops.gemm(a.to(d), w.to(d), 0, bias=bias.to(d))
    ref = a.double() @ w.double().t() + bias.double()
    check(f"gemm NT {dtype} {M}x{N}x{K}", c, ref, TOL[dtype]["out"])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(394, 384, 1152), (98, 96, 288), (130, 1536, 384), (2, 768, 1000)])
def test_gemm_dgrad(dtype, M, N, K):
    """dx[M,N] = dy[M,K] @ W[K,N]  (mode 1: contraction over the weight's rows)."""
    from vtx import ops
    d = dev()
    dy = _mk((M, K), 21, dtype)
    w = _mk((K, N), 22, dtype, 0.1)
    c = ops.gemm(dy.to(d), w.to(d), 1)
    check(f"gemm NN {dtype} {M}x{N}x{K}", c, dy.double() @ w.double(), TOL[dtype]["out"])


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(dtype):
    from vtx import ops
    d = dev()
    B, T, Cin, Cff = 3, 49, 96, 384
    M = B * T
    x = _mk((M, Cin), 31, dtype)
    w1 = _mk((Cff, Cin), 32, dtype, 0.1)
    b1 = _mk((Cff,), 33, torch.float32, 0.1)
    w2 = _mk((Cin, Cff), 34, dtype, 0.1)
    b2 = _mk((Cin,), 35, torch.float32, 0.1)
    res = _mk((M, Cin), 36, dtype)
    scale = torch.tensor([0.0, 1.0 / 0.8, 1.0 / 0.8])
    t = TOL[dtype]
    q = (lambda v: v.to(dtype).double())
    # fc1 + SiLU epilogue (aux = pre-activation z, output h = silu(rounded z))
    h, z = ops.gemm(x.to(d), w1.to(d), 0, bias=b1.to(d), act=ops.ACT_SILU, want_aux=True)
    zr = x.double() @ w1.double().t() + b1.double()
    check(f"fc1 z {dtype}", z, zr, t["out"])
    zq = z.cpu()
    check(f"fc1 silu {dtype}", h, R.silu(zq.double()), t["out"])
    hq = h.cpu().double()
    # fc2 + bias + DropPath scale + residual
    y = ops.gemm(h, w2.to(d), 0, bias=b2.to(d), resid=res.to(d), rowscale=scale.to(d), rows_per_scale=T)
    yr = res.double() + scale.double().repeat_interleave(T)[:, None] * (hq @ w2.double().t() + b2.double())
    check(f"fc2 resid+droppath {dtype}", y, yr, t["out"])
    # dgrad through SiLU: dz = (dh) * silu'(z)
    dyv = _mk((M, Cin), 37, dtype)
    dz = ops.gemm(dyv.to(d), w2.to(d), 1, act=ops.ACT_DSILU, aux_in=z, rowscale=scale.to(d), rows_per_scale=T)
    s = torch.sigmoid(zq.double())
    dzr = scale.double().repeat_interleave(T)[:, None] * (dyv.double() @ w2.double()) * (s * (1 + zq.double() * (1 - s)))
    check(f"dgrad dsilu {dtype}", dz, dzr, t["out"])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(394, 1152, 384), (98, 96, 384), (6272, 288, 96), (130, 768, 3072), (2, 1000, 768),
                                   (25088, 384, 384)])
def test_wgrad(dtype, M, N, K):
    from vtx import ops
    d = dev()
    dy = _mk((M, N), 41, dtype)
    x = _mk((M, K), 42, dtype)
    dW, db = ops.wgrad(dy.to(d), x.to(d))
    check(f"wgrad dW {dtype} {M}x{N}x{K}", dW, dy.double().t() @ x.double(), 2e-5)
    check(f"wgrad db {dtype} {M}x{N}x{K}", db, dy.double().sum(0), 2e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_wgrad_droppath_scale(dtype):
    from vtx import ops
    d = dev()
    B, T, N, K = 4, 49, 96, 384
    dy = _mk((B * T, N), 43, dtype)
    x = _mk((B * T, K), 44, dtype)
    scale = torch.tensor([1.25, 0.0, 1.25, 0.0])
    dW, db = ops.wgrad(dy.to(d), x.to(d), rowscale=scale.to(d), rows_per_scale=T)
    sd = (scale.double().repeat_interleave(T)[:, None] * dy.double()).to(dtype).double()
    check(f"wgrad scaled dW {dtype}", dW, sd.t() @ x.double(), 2e-5 if dtype == torch.float32 else 3e-3)
    check(f"wgrad scaled db {dtype}", db, sd.sum(0), 2e-5)   # bias grad sums the same (rounded) scaled rows


# ------------------------------------------------------------------ attention cores
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,L,nH,D", [(2, 197, 6, 64), (3, 37, 6, 64), (1, 49, 2, 32), (2, 64, 1, 64), (2, 101, 3, 64), (1, 128, 2, 64), (2, 224, 1, 64), (1, 5, 2, 64),
                                      # beyond 224 tokens: the key-block / online-softmax kernels (attention_long.hip): 577 = ViT-S/16 at 384^2
                                      (2, 577, 6, 64), (1, 225, 2, 64), (1, 1025, 1, 64), (2, 300, 3, 32), (1, 256, 2, 64)])
def test_global_attention_core(dtype, B, L, nH, D):
    from vtx import ops
    d = dev()
    qkv = _mk((B, L, 3 * nH * D), 51, dtype)
    do = _mk((B, L, nH * D), 52, dtype)
    o, lse = ops.attention_fwd(qkv.to(d), B, L, nH, D)
    dqkv, _ = ops.attention_bwd(qkv.to(d), o, do.to(d), lse, B, L, nH, D)
    qr = qkv.double().requires_grad_(True)
    orf = R.global_attention_core(qr, nH)
    (dqr,) = torch.autograd.grad(orf, [qr], do.double())
    t = TOL[dtype]
    check(f"global attn fwd {dtype} L{L} h{nH} d{D}", o, orf, t["out"] * 1.5)
    check(f"global attn dqkv {dtype} L{L} h{nH} d{D}", dqkv, dqr, 2e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("L", [197, 224, 193, 177, 161, 145, 130, 37, 101])
def test_vit_attention_wave_counts_are_bitwise_interchangeable(L):
    """Round 6: the ViT attention fast path runs 6 / 7 / 8 waves per workgroup, whichever leaves the fewest idle 16-token tile slots
    (option SATTN_WAVES = 1; L = 197: 13 live tiles on 7 waves).  A tile's arithmetic does not depend on the wave that runs it: o, lse and
    dqkv of every wave count (and of round 1's four waves on pairs of tiles) agree bit for bit, NaN-filled outputs fully written."""
    from vtx import _lib, ops, options
    d = dev()
    B, nH, D = 3, 6, 64
    qkv = _mk((B, L, 3 * nH * D), 71, torch.bfloat16).to(d)
    do = _mk((B, L, nH * D), 72, torch.bfloat16).to(d)
    res = {}
    for w in (1, 8, 7, 6, 4):
        with options.override(SATTN_WAVES=w):
            picked = _lib.load().vtx_sattn_waves(L)
            o, lse = ops.attention_fwd(qkv, B, L, nH, D)
            dqkv, _ = ops.attention_bwd(qkv, o, do, lse, B, L, nH, D)
        res[w] = (picked, o, lse, dqkv)
    nt = (L + 15) // 16
    if L > 128:
        waste = {w: -(-nt // w) * w - nt for w in (8, 7, 6)}
        assert res[1][0] == min((8, 7, 6), key=lambda w: (waste[w], -w)), (L, res[1][0], waste)
    else:
        assert res[1][0] == 8
    for w in (8, 7, 6, 4):
        for name, a, b in zip(("o", "lse", "dqkv"), res[1][1:], res[w][1:]):
            assert torch.isfinite(b.float()).all()
            assert torch.equal(a, b), f"L = {L}: {name} differs between SATTN_WAVES = 1 (-> {res[1][0]} waves) and {w}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,nH,shift", [(2, 14, 14, 3, True), (2, 14, 14, 3, False), (3, 7, 7, 24, True),
                                            (1, 28, 28, 6, True), (2, 56, 56, 3, True)])
def test_window_attention_core(dtype, B, H, W, nH, shift):
    from vtx import ops
    from oracle import tables
    d = dev()
    D, win = 32, 7
    L, ntab = win * win, (2 * win - 1) ** 2
    qkv = _mk((B, H, W, 3 * nH * D), 61, dtype)
    do = _mk((B, H, W, nH * D), 62, dtype)
    rel = _mk((ntab, nH), 63, torch.float32, 0.5)
    pos_np, mask_np = tables.make_pos_mask((H, W), win, shift)
    pos = torch.from_numpy(pos_np)
    mask = torch.from_numpy(mask_np).to(d) if shift else None
    csr = tuple(t.to(d) for t in ops.pos_csr(pos, ntab))
    bias = ops.relpos_bias(rel.to(d), pos.to(d), nH)
    check(f"relpos bias gather h{nH}", bias, rel[pos.reshape(-1)].reshape(L, L, nH).permute(2, 0, 1), 0.0)
    swin = (H, W, win, shift)
    o, lse = ops.attention_fwd(qkv.to(d), B, L, nH, D, swin=swin, bias=bias, mask=mask)
    dqkv, drel = ops.attention_bwd(qkv.to(d), o, do.to(d), lse, B, L, nH, D, swin=swin, bias=bias, mask=mask,
                                   csr=csr, ntab=ntab)
    qr = qkv.double().requires_grad_(True)
    rr = rel.double().requires_grad_(True)
    orf = R.window_attention_core(qr, rr, nH, D, win, shift)
    dqr, drr = torch.autograd.grad(orf, [qr, rr], do.double())
    t = TOL[dtype]
    tag = f"{dtype} {H}x{W} h{nH} s{int(shift)}"
    check(f"window attn fwd {tag}", o, orf, t["out"] * 1.5)
    check(f"window attn dqkv {tag}", dqkv, dqr, 2e-5 if dtype == torch.float32 else 1e-2)
    check(f"window attn drel_pos {tag}", drel, drr, 2e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,nH,shift", [(2, 14, 14, 3, True), (2, 14, 14, 3, False), (3, 7, 7, 24, True),
                                            (1, 28, 28, 6, True), (2, 56, 56, 3, True), (5, 14, 14, 12, True)])
def test_window_attention_fast_path(dtype, B, H, W, nH, shift):
    """One-wave-per-window kernels (attention_win.hip; LDS-resident bias table, region-id masks) vs the oracle; the
    rel_pos gradient must also be bitwise reproducible run to run (per-wave accumulation, fixed-order reduce)."""
    from vtx import ops
    from oracle import tables
    d = dev()
    D, win = 32, 7
    L, ntab = win * win, (2 * win - 1) ** 2
    qkv = _mk((B, H, W, 3 * nH * D), 71, dtype)
    do = _mk((B, H, W, nH * D), 72, dtype)
    rel = _mk((ntab, nH), 73, torch.float32, 0.5)
    pos_np, mask_np = tables.make_pos_mask((H, W), win, shift)
    pos = torch.from_numpy(pos_np).to(d)
    mask = torch.from_numpy(mask_np).to(d) if shift else None
    swin = (H, W, win, shift)
    from vtx.tables import mask_regions
    region = None
    if shift:
        region, ok = mask_regions(mask)
        assert ok, "the reference's local_mask must have the region structure"
    reld = rel.to(d)
    o, lse = ops.wattn_fwd(qkv.to(d), reld, pos, region, B, L, nH, swin)
    dqkv, drel = ops.wattn_bwd(qkv.to(d), o, do.to(d), lse, reld, pos, region, B, L, nH, swin, ntab)
    dqkv2, drel2 = ops.wattn_bwd(qkv.to(d), o, do.to(d), lse, reld, pos, region, B, L, nH, swin, ntab)
    assert torch.equal(drel, drel2) and torch.equal(dqkv, dqkv2), "window attention backward is not deterministic"
    qr = qkv.double().requires_grad_(True)
    rr = rel.double().requires_grad_(True)
    orf = R.window_attention_core(qr, rr, nH, D, win, shift)
    dqr, drr = torch.autograd.grad(orf, [qr, rr], do.double())
    t = TOL[dtype]
    tag = f"{dtype} {H}x{W} h{nH} s{int(shift)} B{B}"
    check(f"wattn fwd {tag}", o, orf, t["out"] * 1.5)
    check(f"wattn dqkv {tag}", dqkv, dqr, 2e-5 if dtype == torch.float32 else 1e-2)
    check(f"wattn drel_pos {tag}", drel, drr, 2e-5 if dtype == torch.float32 else 1e-2)
    # the LDS-atomic scatter (callers of the C ABI without the inverse pos map) sums the same dS in another order
    dqkv3, drel3 = ops.wattn_bwd(qkv.to(d), o, do.to(d), lse, reld, pos, region, B, L, nH, swin, ntab, use_inverse=False)
    assert torch.equal(dqkv3, dqkv)
    check(f"wattn drel_pos, scatter vs gather {tag}", drel3, drel.double(), 2e-5)


def test_wattn_rel_pos_gradient_with_an_arbitrary_pos_table():
    """The gather over tables.pos_inverse makes no assumption about pos (any [L, L] table into the (2 win - 1)^2 bins, one
    bin crowded beyond L entries): against the generic relative-position-bias gradient path of the same cores."""
    from vtx import ops
    d = dev()
    B, H, W, nH, D, win = 3, 14, 14, 4, 32, 7
    L, ntab = win * win, (2 * win - 1) ** 2
    g = torch.Generator().manual_seed(91)
    pos = torch.randint(0, ntab, (L, L), generator=g)
    pos[0, :] = 7
    pos[1, :10] = 7
    pos = pos.to(d)
    qkv = _mk((B, H, W, 3 * nH * D), 92, torch.float32).to(d)
    do = _mk((B, H, W, nH * D), 93, torch.float32).to(d)
    rel = _mk((ntab, nH), 94, torch.float32, 0.5).to(d)
    swin = (H, W, win, False)
    o, lse = ops.wattn_fwd(qkv, rel, pos, None, B, L, nH, swin)
    dqkv, drel = ops.wattn_bwd(qkv, o, do, lse, rel, pos, None, B, L, nH, swin, ntab)
    dqkv2, drel2 = ops.wattn_bwd(qkv, o, do, lse, rel, pos, None, B, L, nH, swin, ntab, use_inverse=False)
    assert torch.equal(dqkv, dqkv2)
    check("wattn drel_pos, arbitrary pos: gather vs scatter", drel, drel2.double(), 2e-5)
    # fp64 reference through autograd on the window partition
    q = qkv.double().cpu().view(B, H // win, win, W // win, win, 3, nH, D).permute(5, 0, 1, 3, 6, 2, 4, 7).reshape(3, -1, nH, L, D)
    rr = rel.double().cpu().requires_grad_(True)
    bias = rr[pos.cpu().reshape(-1)].view(L, L, nH).permute(2, 0, 1)
    att = torch.softmax(q[0] @ q[1].transpose(-1, -2) / D ** 0.5 + bias, -1) @ q[2]          # [B nW, nH, L, D]
    dor = do.double().cpu().view(B, H // win, win, W // win, win, nH, D).permute(0, 1, 3, 5, 2, 4, 6).reshape(-1, nH, L, D)
    (drr,) = torch.autograd.grad(att, [rr], dor)
    check("wattn drel_pos, arbitrary pos vs fp64 autograd", drel, drr, 2e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_wgrad_droppath_zero_rows(dtype):
    """DropPath through the weight gradient with scale_const: dropped samples' rows are skipped, c applied once
    (LDS-DMA kernel for bf16 at these shapes, register-staged kernel for fp32)."""
    from vtx import ops
    d = dev()
    B, T, N, K = 6, 196, 384, 1536
    dy = _mk((B * T, N), 45, dtype)
    x = _mk((B * T, K), 46, dtype)
    c = 1.0 / (1.0 - 0.2)
    scale = torch.tensor([c, 0.0, c, c, 0.0, c])
    dW, db = ops.wgrad(dy.to(d), x.to(d), rowscale=scale.to(d), rows_per_scale=T, scale_const=c)
    keep = (scale > 0).double().repeat_interleave(T)[:, None]
    check(f"wgrad zero-row dW {dtype}", dW, c * (keep * dy.double()).t() @ x.double(), 2e-5)
    check(f"wgrad zero-row db {dtype}", db, c * (keep * dy.double()).sum(0), 2e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues_gelu(dtype):
    """GELU / GELU' epilogues (PVT MLP, DINO head): the kernels use a branch-free erf (|err| < 1e-6); the check is
    against the exact erf GELU of the oracle at the same tolerances as the SiLU epilogues."""
    from vtx import ops
    d = dev()
    M, Cin, Cff = 197, 128, 512
    x = _mk((M, Cin), 131, dtype)
    w1 = _mk((Cff, Cin), 132, dtype, 0.3)
    b1 = _mk((Cff,), 133, torch.float32, 0.5)
    w2 = _mk((Cin, Cff), 134, dtype, 0.1)
    res = _mk((M, Cff), 136, dtype)
    t = TOL[dtype]
    h, z = ops.gemm(x.to(d), w1.to(d), 0, bias=b1.to(d), act=ops.ACT_GELU, want_aux=True)
    zq = z.cpu().double()
    check(f"fc1 gelu {dtype}", h, 0.5 * zq * (1 + torch.erf(zq / 2 ** 0.5)), t["out"])
    dyv = _mk((M, Cin), 137, dtype)
    dgel = 0.5 * (1 + torch.erf(zq / 2 ** 0.5)) + zq * torch.exp(-0.5 * zq * zq) / (2 * torch.pi) ** 0.5
    dz = ops.gemm(dyv.to(d), w2.to(d), 1, act=ops.ACT_DGELU, aux_in=z)
    check(f"dgrad dgelu {dtype}", dz, (dyv.double() @ w2.double()) * dgel, t["out"])
    # a residual next to act' (not on the hot path: the residual is loaded late in the epilogue)
    dz2 = ops.gemm(dyv.to(d), w2.to(d), 1, act=ops.ACT_DGELU, aux_in=z, resid=res.to(d))
    check(f"dgrad dgelu + resid {dtype}", dz2, res.double() + (dyv.double() @ w2.double()) * dgel, t["out"])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,N,K", [(5, 49, 96, 384), (7, 3, 96, 96), (9, 50, 288, 96), (40, 49, 128, 256), (300, 49, 256, 128)])
def test_wgrad_droppath_mask_paths(dtype, B, T, N, K):
    """scale_const weight gradients on both kernels: register-staged (ragged N / K: rows masked in the staged
    registers, one division per 4 rows -- T = 3 takes the per-row division) and LDS-DMA (liveness table per
    workgroup; 300 x 49 tokens span several split-K slices and many samples per slice)."""
    from vtx import ops
    d = dev()
    g = torch.Generator().manual_seed(B * 1000 + T)
    dy = _mk((B * T, N), 145, dtype)
    x = _mk((B * T, K), 146, dtype)
    c = 1.0 / (1.0 - 0.3)
    scale = (torch.rand(B, generator=g) >= 0.3).float() * c
    dW, db = ops.wgrad(dy.to(d), x.to(d), rowscale=scale.to(d), rows_per_scale=T, scale_const=c)
    keep = (scale > 0).double().repeat_interleave(T)[:, None]
    tol = 2e-5 if dtype == torch.float32 else 1e-4
    check(f"wgrad mask dW {dtype} {B}x{T}x{N}x{K}", dW, c * (keep * dy.double()).t() @ x.double(), tol)
    check(f"wgrad mask db {dtype} {B}x{T}x{N}x{K}", db, c * (keep * dy.double()).sum(0), tol)


def test_cast_weights_multi_tensor():
    """One-launch fp32 -> bf16 cast of a list of matrices, plain + transposed (csrc/cast.hip), incl. ragged sizes,
    a 4-D conv weight and a weight-normed layer (whose weight is not a leaf parameter and must be skipped)."""
    from vtx import functional as VF
    d = dev()
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(48, 96), torch.nn.Linear(96, 1000, bias=False),
                              torch.nn.Conv2d(3, 40, 16, stride=16), torch.nn.LayerNorm(7), torch.nn.Linear(1, 1),
                              torch.nn.utils.weight_norm(torch.nn.Linear(64, 32))).to(d)
    plan = VF.WeightPlan(net)
    assert len(plan.params) == 4
    got = plan.cast_all()
    for p in plan.params:
        w, wt = got[id(p)]
        assert w.shape == p.shape and w.dtype == torch.bfloat16
        assert torch.equal(w, p.detach().to(torch.bfloat16))
        assert torch.equal(wt, p.detach().reshape(p.shape[0], -1).t().contiguous().to(torch.bfloat16))
        assert w.data_ptr() % 128 == 0 and wt.data_ptr() % 128 == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N", [10, 100])
def test_linear_ragged_out_features(dtype, N):
    """nn.Linear with out-features that are not a multiple of the 16-byte vector (10- / 100-way classifiers)."""
    from vtx import functional as VF
    d = dev()
    x = _mk((24, 768), 91, dtype)
    w = _mk((N, 768), 92, torch.float32, 0.05)
    b = _mk((N,), 93, torch.float32, 0.1)
    dy = _mk((24, N), 94, dtype)
    xd, wd, bd = x.to(d).requires_grad_(True), w.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
    y = VF.LinearFn.apply(xd, wd, bd)
    assert tuple(y.shape) == (24, N)
    gx, gw, gb = torch.autograd.grad(y, [xd, wd, bd], dy.to(d))
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = R.linear(xr, wr, br)
    rx, rw, rb = torch.autograd.grad(yr, [xr, wr, br], dy.double())
    t = TOL[dtype]
    check(f"linear N={N} {dtype} y", y, yr, t["out"])
    check(f"linear N={N} {dtype} dx", gx, rx, t["grad"])
    check(f"linear N={N} {dtype} dW", gw, rw, t["grad"])
    check(f"linear N={N} {dtype} db", gb, rb, t["grad"])
    assert gw.shape == w.shape and gb.shape == b.shape


@pytest.mark.parametrize("max_norm", [0.0, 0.05])
def test_fused_adamw_vs_torch_and_oracle(max_norm):
    """clip_grad_norm_ + AdamW as two multi-tensor kernels (csrc/optim.hip) vs torch.optim.AdamW (+ clip_grad_norm_) and vs
    the oracle's restatement, 3 steps, two param groups, ragged sizes and a 4-byte-aligned view (DDP bucket style)."""
    from vtx.optim import FusedAdamW
    d = dev()
    torch.manual_seed(0)
    flat = torch.randn(3 * 507 + 8, device=d)
    shapes = [(1000, 768), (4097,), (169, 3), (5,), (96, 48)]
    def make():
        ps = [torch.nn.Parameter(_mk(s, 200 + i, torch.float32, 0.3).to(d)) for i, s in enumerate(shapes)]
        ps.append(torch.nn.Parameter(flat[1:1 + 507].clone().view(169, 3)))
        return ps
    pa, pb = make(), make()
    unaligned = torch.zeros(1 + 507, device=d)[1:].view(169, 3)           # gradient living at a 4-byte offset
    ga = lambda ps: [{"params": ps[:2], "weight_decay": 0.05}, {"params": ps[2:], "weight_decay": 0.0, "lr": 3e-3}]
    oa = FusedAdamW(ga(pa), lr=1e-2, betas=(0.9, 0.95), eps=1e-8)
    ob = torch.optim.AdamW(ga(pb), lr=1e-2, betas=(0.9, 0.95), eps=1e-8)
    ref = [dict(p=p.detach().double().cpu(), m=torch.zeros_like(p, dtype=torch.float64, device="cpu"),
                v=torch.zeros_like(p, dtype=torch.float64, device="cpu")) for p in pa]
    hyp = [(1e-2, 0.05)] * 2 + [(3e-3, 0.0)] * 4
    for step in range(1, 4):
        grads = [_mk(tuple(p.shape), 300 + 10 * step + i, torch.float32, 0.02 * (i + 1)) for i, p in enumerate(pa)]
        for ps in (pa, pb):
            for p, g in zip(ps, grads):
                p.grad = g.to(d).clone()
        unaligned.copy_(grads[5].to(d)); pa[5].grad = unaligned
        total = oa.step(max_grad_norm=max_norm)
        if max_norm > 0:
            tref = torch.nn.utils.clip_grad_norm_(pb, max_norm)
            check(f"fused clip total norm step {step}", total, tref, 1e-6)
        ob.step()
        gs = [g.double() for g in grads]
        if max_norm > 0:
            gs, _ = R.clip_grad_norm(gs, max_norm)
        for r, g, (lr, wd) in zip(ref, gs, hyp):
            r["p"], r["m"], r["v"] = R.adamw_step(r["p"], g, r["m"], r["v"], step, lr, 0.9, 0.95, 1e-8, wd)
    for i, (a, b, r) in enumerate(zip(pa, pb, ref)):
        check(f"fused adamw vs torch p{i} clip={max_norm}", a, b, 2e-6)
        check(f"fused adamw vs oracle p{i} clip={max_norm}", a, r["p"], 2e-6)
        check(f"fused adamw exp_avg_sq p{i}", oa.state[a]["exp_avg_sq"], r["v"], 2e-6)
    assert float(oa.state[pa[0]]["step"]) == 3.0
    sd = oa.state_dict()                                                  # same state layout as torch.optim.AdamW
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,K,eps", [(128, 1000, 0.1), (7, 10, 0.0), (5, 100, 0.2)])
def test_mix_loss_kernel(dtype, B, K, eps):
    """Fused MixLoss (value + d logits) vs the oracle restatement of reference loss.py:53-86 (pinned by golden G6)."""
    from vtx.train_step import MixLoss
    d = dev()
    gen = torch.Generator().manual_seed(21)
    logits = (torch.randn(B, K, generator=gen) * 3).to(dtype)
    l1 = torch.randint(0, K, (B,), generator=gen)
    l2 = torch.randint(0, K, (B,), generator=gen)
    r = torch.rand(B, generator=gen)
    r[0], r[-1] = 1.0, 0.0                                   # pure targets: the 0 log 0 = 0 convention
    x = logits.to(d).requires_grad_(True)
    loss = MixLoss(eps)(x, l1.to(d), l2.to(d), r.to(d))
    (loss * 0.5).backward()
    xr = logits.double().requires_grad_(True)
    lr = R.mix_loss(xr, l1, l2, r.double(), eps)
    (gr,) = torch.autograd.grad(lr * 0.5, [xr])
    check(f"mix loss {dtype} B{B} K{K}", loss, lr, 2e-6)
    check(f"mix loss d logits {dtype} B{B} K{K}", x.grad, gr, 2e-5 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize("reduction", ["none", "sum", "batchmean"])
@pytest.mark.parametrize("dtype", DTYPES + [torch.float16])
def test_mix_loss_reductions(dtype, reduction):
    """the reference's MixLoss takes every reduction (loss.py:73-84: 'none' -> per-sample sums, 'mean' -> sum / B,
    any other string -> the plain sum) and any floating dtype; all of them run on the fused kernel."""
    from vtx.train_step import MixLoss
    d = dev()
    B, K, eps = 9, 40, 0.1
    gen = torch.Generator().manual_seed(23)
    logits = (torch.randn(B, K, generator=gen) * 2).to(dtype)
    l1, l2 = torch.randint(0, K, (B,), generator=gen), torch.randint(0, K, (B,), generator=gen)
    r = torch.rand(B, generator=gen)
    cot = torch.randn(B, generator=gen).double() if reduction == "none" else torch.tensor(0.7).double()
    x = logits.to(d).requires_grad_(True)
    loss = MixLoss(eps, reduction)(x, l1.to(d), l2.to(d), r.to(d))
    (loss.double() * cot.to(d)).sum().backward()
    xr = logits.double().requires_grad_(True)
    lr = R.mix_loss(xr, l1, l2, r.double(), eps, reduction)
    (gr,) = torch.autograd.grad((lr * cot).sum(), [xr])
    assert loss.shape == lr.shape and x.grad.dtype == dtype
    lo = dtype != torch.float32
    check(f"mix loss reduction={reduction} {dtype}", loss, lr, 2e-6)
    check(f"mix loss reduction={reduction} d logits {dtype}", x.grad, gr, 6e-3 if lo else 2e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_deferred_column_reductions_match_the_kernels_own_bit_for_bit(dtype):
    """LayerNorm dgamma / dbeta and the window-attention rel_pos gradient, reduced later by ONE colreduce_multi launch
    (what TransformerLayerFn.backward does per layer) vs the reductions the kernels run themselves."""
    from oracle import tables
    from vtx import ops
    from vtx.tables import mask_regions
    d = dev()
    rows, C = 2 * 14 * 14, 384
    x, dy, res = _mk((rows, C), 501, dtype).to(d), _mk((rows, C), 502, dtype).to(d), _mk((rows, C), 503, dtype).to(d)
    g, b = (1 + 0.1 * _mk((C,), 504, torch.float32)).to(d), _mk((C,), 505, torch.float32, 0.1).to(d)
    _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
    dx_a, dg_a, db_a = ops.layernorm_bwd(dy, x, mean, rstd, g, dres=res)
    dx_b, part_ln = ops.layernorm_bwd(dy, x, mean, rstd, g, dres=res, defer=True)
    B, H, nH, win = 2, 14, 12, 7
    L, ntab = win * win, (2 * win - 1) ** 2
    qkv, do = _mk((B, H, H, 3 * nH * 32), 506, dtype).to(d), _mk((B, H, H, nH * 32), 507, dtype).to(d)
    rel = _mk((ntab, nH), 508, torch.float32, 0.5).to(d)
    pos_np, mask_np = tables.make_pos_mask((H, H), win, True)
    pos = torch.from_numpy(pos_np).to(d)
    region, ok = mask_regions(torch.from_numpy(mask_np).to(d))
    assert ok
    swin = (H, H, win, True)
    o, lse = ops.wattn_fwd(qkv, rel, pos, region, B, L, nH, swin)
    dq_a, drel_a = ops.wattn_bwd(qkv, o, do, lse, rel, pos, region, B, L, nH, swin, ntab)
    dq_b, part_rel = ops.wattn_bwd(qkv, o, do, lse, rel, pos, region, B, L, nH, swin, ntab, defer=True)
    (dg_b, db_b), (dg_c, db_c), (drel_b, none) = ops.colreduce_multi([part_ln, part_ln, part_rel])
    assert none is None
    assert torch.equal(dx_a, dx_b) and torch.equal(dq_a, dq_b)
    assert torch.equal(dg_a, dg_b) and torch.equal(db_a, db_b) and torch.equal(dg_a, dg_c) and torch.equal(db_a, db_c)
    assert torch.equal(drel_a, drel_b.view(ntab, nH))
    if dtype == torch.bfloat16:
        # round 3: the same reductions riding in the reduce launch of a grouped weight gradient (one launch less per layer)
        M = 25088                                        # 108 tiles -> 4 split-K slices (slab segments + column segments) ...
        jobs = [(_mk((M, 384), 511, dtype).to(d), _mk((M, 1536), 512, dtype).to(d), True, None),
                (_mk((M, 1536), 513, dtype).to(d), _mk((M, 384), 514, dtype).to(d), True, None)]
        small = [(j[0][:256].contiguous(), j[1][:256].contiguous(), True, None) for j in jobs]   # ... and nz = 1 (columns only)
        for jj in (jobs, small):
            assert ops.wgrad_group_ok(jj)
            plain = ops.wgrad_group(jj)
            got, red = ops.wgrad_group(jj, colparts=[part_ln, part_ln, part_rel])
            for (a, ab), (g2, gb) in zip(plain, got):
                assert torch.equal(a, g2) and torch.equal(ab, gb)
            assert torch.equal(red[0][0], dg_a) and torch.equal(red[0][1], db_a) and torch.equal(red[1][0], dg_a)
            assert torch.equal(red[2][0].view(ntab, nH), drel_a) and red[2][1] is None
