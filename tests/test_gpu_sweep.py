"""-m gpu: seeded random sweep of shapes / epilogue combinations through the C ABI vs the CPU oracle.

The fixed cases of test_gpu_kernels.py are the model shapes; this sweep walks the supported range around them: ragged
row counts, widths that are only multiples of 8, every epilogue combination, window sizes 2..8 with and without shift,
sequence lengths 1..224, split-K slices cut at arbitrary sample boundaries."""
import random

import pytest
import torch

from gpu_util import TOL, check, dev
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]


def _mk(shape, gen, dtype, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).to(dtype)


def _cases(seed, n):
    rng = random.Random(seed)
    return [rng.randrange(1 << 30) for _ in range(n)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", _cases(101, 14))
def test_sweep_gemm_epilogues(dtype, case):
    from vtx import ops
    rng = random.Random(case)
    g = torch.Generator().manual_seed(case)
    d = dev()
    T = rng.choice([1, 3, 49, 64, 197])
    B = rng.randint(1, 5)
    M = B * T
    N = 8 * rng.randint(1, 96)
    K = 8 * rng.randint(1, 96)
    mode = rng.choice([0, 0, 1])
    a = _mk((M, K), g, dtype)
    w = _mk((N, K) if mode == 0 else (K, N), g, dtype, 0.1)
    kw, ref = {}, a.double() @ (w.double().t() if mode == 0 else w.double())
    if rng.random() < 0.7:
        bias = _mk((N,), g, torch.float32, 0.3)
        kw["bias"] = bias.to(d); ref = ref + bias.double()
    act = rng.choice([None, "silu", "gelu", "dsilu", "dgelu"])
    z = None
    if act in ("dsilu", "dgelu"):
        z = _mk((M, N), g, dtype)
        zd = z.double()
        if act == "dsilu":
            s = torch.sigmoid(zd); der = s * (1 + zd * (1 - s))
        else:
            der = 0.5 * (1 + torch.erf(zd / 2 ** 0.5)) + zd * torch.exp(-0.5 * zd * zd) / (2 * torch.pi) ** 0.5
        kw.update(act=ops.ACT_DSILU if act == "dsilu" else ops.ACT_DGELU, aux_in=z.to(d)); ref = ref * der
    if rng.random() < 0.5 and act not in ("silu", "gelu"):     # DropPath scale rides on the residual GEMMs / dgrads only
        sc = (torch.rand(B, generator=g) > 0.3).float() / 0.7
        kw.update(rowscale=sc.to(d), rows_per_scale=T); ref = ref * sc.double().repeat_interleave(T)[:, None]
    if rng.random() < 0.5 and act not in ("silu", "gelu"):
        res = _mk((M, N), g, dtype)
        kw["resid"] = res.to(d); ref = ref + res.double()
    tag = f"sweep gemm {dtype} {M}x{N}x{K} mode{mode} {act} {sorted(k for k in kw if k not in ('act', 'aux_in'))}"
    if act in ("silu", "gelu"):
        h, zz = ops.gemm(a.to(d), w.to(d), mode, act=ops.ACT_SILU if act == "silu" else ops.ACT_GELU, want_aux=True, **kw)
        check(tag + " z", zz, ref, TOL[dtype]["out"])
        zq = zz.cpu().double()
        check(tag + " h", h, R.silu(zq) if act == "silu" else 0.5 * zq * (1 + torch.erf(zq / 2 ** 0.5)), TOL[dtype]["out"])
    else:
        check(tag, ops.gemm(a.to(d), w.to(d), mode, **kw), ref, TOL[dtype]["out"])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", _cases(202, 12))
def test_sweep_wgrad(dtype, case):
    from vtx import ops
    rng = random.Random(case)
    g = torch.Generator().manual_seed(case)
    d = dev()
    T = rng.choice([1, 5, 49, 196, 197])
    B = rng.randint(1, 40)
    N, K = 8 * rng.randint(1, 64), 8 * rng.randint(1, 64)
    dy, x = _mk((B * T, N), g, dtype), _mk((B * T, K), g, dtype)
    kind = rng.choice(["none", "const", "free"])
    kw, rowf = {}, torch.ones(B * T, 1, dtype=torch.float64)
    if kind == "const":
        c = 1 / (1 - 0.25)
        sc = (torch.rand(B, generator=g) > 0.25).float() * c
        kw = dict(rowscale=sc.to(d), rows_per_scale=T, scale_const=c)
        rowf = sc.double().repeat_interleave(T)[:, None]
    elif kind == "free":
        sc = torch.rand(B, generator=g) * 2
        kw = dict(rowscale=sc.to(d), rows_per_scale=T)
    want_bias = rng.random() < 0.7
    dW, db = ops.wgrad(dy.to(d), x.to(d), want_bias=want_bias, **kw)
    sd = dy.double() * rowf if kind != "free" else (sc.double().repeat_interleave(T)[:, None] * dy.double()).to(dtype).double()
    tol = 2e-5 if dtype == torch.float32 else (3e-3 if kind == "free" else 1e-4)
    check(f"sweep wgrad {dtype} {B}x{T} {N}x{K} {kind}", dW, sd.t() @ x.double(), tol)
    if want_bias:
        check(f"sweep wgrad bias {dtype} {B}x{T} {N} {kind}", db, sd.sum(0), tol)
    else:
        assert db is None


def test_wgrad_droppath_many_samples_per_slice():
    """One-token sequences: a split-K slice spans more samples than the LDS-DMA kernel's liveness table holds -> the
    register-staged kernel takes over (same results)."""
    from vtx import ops
    d = dev()
    g = torch.Generator().manual_seed(9)
    B, T, N, K = 20000, 1, 128, 128
    dy, x = _mk((B * T, N), g, torch.bfloat16), _mk((B * T, K), g, torch.bfloat16)
    c = 1 / 0.8
    sc = (torch.rand(B, generator=g) > 0.2).float() * c
    dW, db = ops.wgrad(dy.to(d), x.to(d), rowscale=sc.to(d), rows_per_scale=T, scale_const=c)
    keep = (sc > 0).double()[:, None]
    check("wgrad mask, 20000 one-token samples", dW, c * (keep * dy.double()).t() @ x.double(), 1e-4)
    check("wgrad mask bias, 20000 one-token samples", db, c * (keep * dy.double()).sum(0), 1e-4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", _cases(303, 10))
def test_sweep_layernorm(dtype, case):
    from vtx import ops
    rng = random.Random(case)
    g = torch.Generator().manual_seed(case)
    d = dev()
    rows, C = rng.randint(1, 3000), 8 * rng.randint(2, 192)
    x = _mk((rows, C), g, dtype, 2.0) + 0.3
    dy, dres = _mk((rows, C), g, dtype), _mk((rows, C), g, dtype)
    gm, bt = 1 + 0.1 * _mk((C,), g, torch.float32), 0.1 * _mk((C,), g, torch.float32)
    eps = rng.choice([1e-5, 1e-6])
    y, mean, rstd = ops.layernorm_fwd(x.to(d), gm.to(d), bt.to(d), eps)
    use_res = rng.random() < 0.5
    dx, dg, db = ops.layernorm_bwd(dy.to(d), x.to(d), mean, rstd, gm.to(d), dres=dres.to(d) if use_res else None)
    xr, gr, br = x.double().requires_grad_(True), gm.double().requires_grad_(True), bt.double().requires_grad_(True)
    yr = R.layer_norm(xr, gr, br, eps)
    dxr, dgr, dbr = torch.autograd.grad(yr, [xr, gr, br], dy.double())
    t = TOL[dtype]
    check(f"sweep ln fwd {dtype} {rows}x{C}", y, yr, t["out"])
    check(f"sweep ln dx {dtype} {rows}x{C}", dx, dxr + (dres.double() if use_res else 0), t["out"])
    check(f"sweep ln dgamma {dtype} {rows}x{C}", dg, dgr, 1e-5 if dtype == torch.float32 else 3e-4)
    check(f"sweep ln dbeta {dtype} {rows}x{C}", db, dbr, 1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", _cases(404, 10))
def test_sweep_global_attention(dtype, case):
    from vtx import ops
    rng = random.Random(case)
    g = torch.Generator().manual_seed(case)
    d = dev()
    B, L, nH, D = rng.randint(1, 5), rng.choice([1, 2, 15, 16, 17, 37, 63, 64, 65, 128, 129, 197, 223, 224]), rng.randint(1, 6), 64
    qkv, do = _mk((B, L, 3 * nH * D), g, dtype), _mk((B, L, nH * D), g, dtype)
    if rng.random() < 0.3:
        qkv = qkv * 4                                  # sharp softmax rows
    o, lse = ops.attention_fwd(qkv.to(d), B, L, nH, D)
    dqkv, _ = ops.attention_bwd(qkv.to(d), o, do.to(d), lse, B, L, nH, D)
    qr = qkv.double().requires_grad_(True)
    orf = R.global_attention_core(qr, nH)
    (dqr,) = torch.autograd.grad(orf, [qr], do.double())
    check(f"sweep attn fwd {dtype} B{B} L{L} h{nH}", o, orf, TOL[dtype]["out"] * 1.5)
    check(f"sweep attn dqkv {dtype} B{B} L{L} h{nH}", dqkv, dqr, 2e-5 if dtype == torch.float32 else 1.2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", _cases(505, 10))
def test_sweep_window_attention(dtype, case):
    from oracle import tables
    from vtx import ops
    from vtx.tables import mask_regions
    rng = random.Random(case)
    g = torch.Generator().manual_seed(case)
    d = dev()
    win = rng.randint(2, 8)
    H, W = win * rng.randint(1, 4), win * rng.randint(1, 4)
    B, nH, D, shift = rng.randint(1, 4), rng.randint(1, 7), 32, rng.random() < 0.6
    L, ntab = win * win, (2 * win - 1) ** 2
    qkv, do = _mk((B, H, W, 3 * nH * D), g, dtype), _mk((B, H, W, nH * D), g, dtype)
    rel = _mk((ntab, nH), g, torch.float32, 0.5)
    pos_np, mask_np = tables.make_pos_mask((H, W), win, shift)
    pos = torch.from_numpy(pos_np).to(d)
    mask = torch.from_numpy(mask_np).to(d) if shift else None
    swin = (H, W, win, shift)
    if ops.wattn_supported(D, win):                    # one-wave-per-window kernels (window <= 7), as the module dispatches
        region = None
        if shift:
            region, ok = mask_regions(mask)
            assert ok
        o, lse = ops.wattn_fwd(qkv.to(d), rel.to(d), pos, region, B, L, nH, swin)
        dqkv, drel = ops.wattn_bwd(qkv.to(d), o, do.to(d), lse, rel.to(d), pos, region, B, L, nH, swin, ntab)
    else:                                              # window 8: generic masked kernels
        csr = tuple(t.to(d) for t in ops.pos_csr(torch.from_numpy(pos_np), ntab))
        bias = ops.relpos_bias(rel.to(d), pos, nH)
        o, lse = ops.attention_fwd(qkv.to(d), B, L, nH, D, swin=swin, bias=bias, mask=mask)
        dqkv, drel = ops.attention_bwd(qkv.to(d), o, do.to(d), lse, B, L, nH, D, swin=swin, bias=bias, mask=mask, csr=csr,
                                       ntab=ntab)
    qr, rr = qkv.double().requires_grad_(True), rel.double().requires_grad_(True)
    orf = R.window_attention_core(qr, rr, nH, D, win, shift)
    dqr, drr = torch.autograd.grad(orf, [qr, rr], do.double())
    tag = f"{dtype} {H}x{W} w{win} h{nH} s{int(shift)} B{B}"
    check(f"sweep wattn fwd {tag}", o, orf, TOL[dtype]["out"] * 1.5)
    check(f"sweep wattn dqkv {tag}", dqkv, dqr, 2e-5 if dtype == torch.float32 else 1e-2)
    check(f"sweep wattn drel_pos {tag}", drel, drr, 2e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", _cases(606, 8))
def test_sweep_sr_attention(dtype, case):
    """PVT / Twins cross attention: Lq queries against Lk <= 64 reduced keys, head dim 64 or 32."""
    from vtx import ops
    rng = random.Random(case)
    g = torch.Generator().manual_seed(case)
    d = dev()
    B, nH = rng.randint(1, 4), rng.randint(1, 8)
    Lq, Lk = rng.choice([1, 7, 49, 50, 196, 197, 784, 1000]), rng.choice([1, 5, 16, 49, 50, 64])
    C = nH * rng.choice([64, 32])
    q, kv, do = _mk((B, Lq, C), g, dtype), _mk((B, Lk, 2 * C), g, dtype), _mk((B, Lq, C), g, dtype)
    o, lse = ops.srattn_fwd(q.to(d), kv.to(d), B, Lq, Lk, nH)
    dq, dkv = ops.srattn_bwd(q.to(d), kv.to(d), o, do.to(d), lse, B, Lq, Lk, nH)
    qr, kvr = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    orf = R.sr_attention_core(qr, kvr, nH)
    dqr, dkvr = torch.autograd.grad(orf, [qr, kvr], do.double())
    tag = f"{dtype} B{B} Lq{Lq} Lk{Lk} h{nH} d{C // nH}"
    check(f"sweep srattn fwd {tag}", o, orf, TOL[dtype]["out"] * 1.5)
    if Lk == 1:                     # a single key: the true dq is exactly 0 (see test_gpu_pvt.test_sr_attention_core)
        assert float(dq.abs().max()) <= 1e-4
    else:
        check(f"sweep srattn dq {tag}", dq, dqr, 2e-5 if dtype == torch.float32 else 1e-2)
    check(f"sweep srattn dkv {tag}", dkv, dkvr, 2e-5 if dtype == torch.float32 else 1.5e-2)


# ------------------------------------------------------------------ whole modules on random small configurations (fp32)
def _model_vs_oracle_fp32(model, oracle_fwd, x, what):
    """fp32 parity mode of the drop-in module vs the fp32 CPU oracle on the same seeded weights: logits and the
    relative L2 of ALL parameter gradients taken together."""
    from test_gpu_models import _seeded_init
    sd = _seeded_init(model, 5)
    model.to(dev()).train()
    out = model(x.to(dev()))
    P = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
    ref = oracle_fwd(P, x.double())
    check(f"{what} logits vs oracle", out, ref, 2e-4)
    cot = torch.randn(ref.shape, generator=torch.Generator().manual_seed(7))
    (out * cot.to(dev())).sum().backward()
    names = [n for n, _ in model.named_parameters()]
    rg = torch.autograd.grad((ref * cot.double()).sum(), [P[n] for n in names])
    num = den = 0.0
    for n, r in zip(names, rg):
        gp = dict(model.named_parameters())[n].grad.double().cpu()
        num += (gp - r).norm().item() ** 2
        den += r.norm().item() ** 2
    from gpu_util import report
    assert report(f"{what} all-parameter gradient rel-L2 vs oracle", (num / den) ** 0.5, 5e-4)


@pytest.mark.parametrize("case", _cases(707, 5))
def test_sweep_swin_configurations(case):
    from models import SwinTransformer
    from oracle import ref_models as M
    rng = random.Random(case)
    win = rng.choice([2, 4, 7])
    # every stage's grid (H/4 ... H/32) must split into windows: sides are multiples of 32 * win
    ka, kb = rng.randint(1, 2), rng.randint(1, 2)
    if 32 * win * ka * 32 * win * kb > 224 * 448:
        kb = 1
    size = (32 * win * ka, 32 * win * kb)
    heads = tuple(rng.randint(1, 3) for _ in range(4))
    cfg = dict(image_size=size, n_class=8 * rng.randint(1, 4), depths=tuple(rng.randint(1, 2) for _ in range(4)),
               dims=tuple(32 * h for h in heads), dim_head=32, n_heads=heads, dim_ffs=tuple(8 * rng.randint(4, 16) for _ in range(4)),
               window_size=win)
    torch.manual_seed(case)
    model = SwinTransformer(**cfg, drop_path=0.0)
    x = torch.randn(rng.randint(1, 3), 3, *size, generator=torch.Generator().manual_seed(case))
    _model_vs_oracle_fp32(model, lambda P, xx: M.swin_forward(P, xx, cfg), x, f"sweep swin {size} w{win} h{heads} d{cfg['depths']}")


@pytest.mark.parametrize("case", _cases(808, 4))
def test_sweep_vit_configurations(case):
    from models import VisionTransformer
    from oracle import ref_models as M
    from vtx.nn import Linear
    rng = random.Random(case)
    patch = rng.choice([8, 16, 32])
    side = patch * rng.randint(2, min(14, 224 // patch))
    n_head = rng.randint(1, 4)
    cfg = dict(image_size=side, window_size=patch, depth=rng.randint(1, 3), dim=64 * n_head, n_head=n_head, dim_ff=8 * rng.randint(8, 48))
    n_class = 8 * rng.randint(1, 8)
    torch.manual_seed(case)
    model = VisionTransformer(Linear(cfg["dim"], n_class), side, patch, cfg["depth"], cfg["dim"], n_head, cfg["dim_ff"], 0.0, 0.0, 0.0, 0.0)
    x = torch.randn(rng.randint(1, 4), 3, side, side, generator=torch.Generator().manual_seed(case))

    def oracle(P, xx):
        feat = M.vit_forward(P, xx, cfg)
        return feat @ P["head.weight"].t() + P["head.bias"]
    _model_vs_oracle_fp32(model, oracle, x, f"sweep vit {side}/{patch} h{n_head} depth{cfg['depth']}")


def test_empty_batch_is_refused_loudly():
    """A batch of zero images is not a supported input: the first op says so (VtxError), nothing is launched on NULL."""
    from models import VisionTransformer
    from vtx._lib import VtxError
    from vtx.nn import Linear
    model = VisionTransformer(Linear(128, 16), 64, 16, 2, 128, 2, 256, 0.0, 0.0, 0.0, 0.0).to(dev()).train()
    with pytest.raises(VtxError, match="empty tensor"):
        model(torch.zeros(0, 3, 64, 64, device=dev()))


def test_tensors_beyond_2g_elements():
    """64-bit addressing: 6 M rows x 384 bf16 (2.3 G elements, 4.6 GB) through LayerNorm, the LDS-DMA GEMM (output rows
    past element 2^31) and the split-K weight gradient (contraction over all 6 M rows)."""
    from vtx import ops
    d = dev()
    rows, C = 6_000_000, 384
    gen = torch.Generator(device=d).manual_seed(3)
    x = torch.randn(rows, C, device=d, dtype=torch.bfloat16, generator=gen)
    g, b = torch.ones(C, device=d), torch.zeros(C, device=d)
    y, _, _ = ops.layernorm_fwd(x, g, b, 1e-6)
    for r in (0, 3_000_000, rows - 1):
        ref = torch.nn.functional.layer_norm(x[r].float(), (C,), eps=1e-6)
        check(f"big LN row {r}", y[r], ref, 4e-3)
    del y
    w = (torch.randn(C, 64, device=d, generator=gen) * 0.1).bfloat16()
    a = torch.randn(rows, 64, device=d, dtype=torch.bfloat16, generator=gen)
    c = ops.gemm(a, w, 0)
    for r in (0, 2_999_999, 5_592_406, rows - 1):                     # 5 592 406 * 384 > 2^31
        check(f"big GEMM row {r}", c[r], a[r].float() @ w.float().t(), 4e-3)
    dW, _ = ops.wgrad(c, a)
    ref = torch.zeros(C, 64, device=d, dtype=torch.float64)
    for i in range(0, rows, 500_000):
        ref += (c[i:i + 500_000].float().t() @ a[i:i + 500_000].float()).double()
    check("big wgrad", dW, ref, 1e-5)
