"""The fused MLP of the narrow stages (csrc/mlp_fused.hip: vtx_mlp_fwd / vtx_mlp_bwd; reference models/layer.py:186-196 inside the
block of models/swin_transformer.py:193-197) against the four vtx_gemm launches it replaces (bit for bit) and the fp64 oracle."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vision-transformers-pytorch_amd"))

from gpu_util import check, dev          # noqa: E402

pytestmark = pytest.mark.gpu


def _operands(M, C, ff, seed, drop=0.0, rps=49):
    d = dev()
    g = torch.Generator(device="cpu").manual_seed(seed)
    bf = lambda *s, std=1.0: (torch.randn(*s, generator=g) * std).to(torch.bfloat16).to(d)
    ln2, x1, dy = bf(M, C), bf(M, C), bf(M, C, std=0.05)
    w1, w2 = bf(ff, C, std=C ** -0.5), bf(C, ff, std=ff ** -0.5)
    b1 = (torch.randn(ff, generator=g) * 0.3).to(d)
    b2 = (torch.randn(C, generator=g) * 0.3).to(d)
    s = None
    if drop > 0:
        ns = (M + rps - 1) // rps
        s = ((torch.rand(ns, generator=g) >= drop).float() / (1.0 - drop)).to(d)
    return ln2, x1, dy, w1, b1, w2, b2, s


def _fused(ln2, x1, dy, w1, b1, w2, b2, s, rps, want_zh=False):
    from vtx import _lib, ops
    lib = _lib.load()
    M, C = ln2.shape
    ff = w1.shape[0]
    p = lambda t: None if t is None else t.data_ptr()
    nan = lambda *sh: torch.full(sh, float("nan"), dtype=torch.bfloat16, device=ln2.device)
    y, z, h = nan(M, C), (nan(M, ff) if want_zh else None), (nan(M, ff) if want_zh else None)
    _lib.check(lib.vtx_mlp_fwd(1, p(ln2), p(w1), p(b1), p(w2), p(b2), p(x1), p(s), rps, p(y), p(z), p(h), M, C, ff, ops._stream()), "vtx_mlp_fwd")
    hb, dz, dln2 = nan(M, ff), nan(M, ff), nan(M, C)
    _lib.check(lib.vtx_mlp_bwd(1, p(ln2), p(dy), p(w1), p(b1), p(w2), p(s), rps, p(hb), p(dz), p(dln2), M, C, ff, ops._stream()), "vtx_mlp_bwd")
    return y, z, h, hb, dz, dln2


def _unfused(ln2, x1, dy, w1, b1, w2, b2, s, rps):
    """the launches of vtx_layer_fwd / vtx_layer_bwd's MLP branch (csrc/layer.hip), one vtx_gemm each"""
    from vtx import ops
    h, z = ops.gemm(ln2, w1, 0, bias=b1, act=ops.ACT_SILU, want_aux=True)
    y = ops.gemm(h, w2, 0, bias=b2, resid=x1, rowscale=s, rows_per_scale=rps)
    dz = ops.gemm(dy, w2, 1, act=ops.ACT_DSILU, aux_in=z, rowscale=s, rows_per_scale=rps)
    dln2 = ops.gemm(dz, w1, 1)
    return y, z, h, dz, dln2


@pytest.mark.parametrize("waves", [1, 404, 809, 812, 816, 708, 512, 612])      # option MLP_FUSED: default | 100 b + f variant codes
@pytest.mark.parametrize("M,C,ff,drop", [(34496, 96, 384, 0.0), (34496, 96, 384, 0.25), (32777, 96, 384, 0.25), (401, 96, 384, 0.0),
                                         (37632, 64, 256, 0.25), (33001, 64, 512, 0.1)])
def test_fused_mlp_is_bitwise_the_four_gemm_launches(M, C, ff, drop, waves):
    from vtx import _lib, options
    rps = 49
    ops_ = _operands(M, C, ff, 5 + M % 7, drop, rps)
    assert _lib.load().vtx_mlp_fused_ok(1, 1 << 20, C, ff) == 1
    with options.override(MLP_FUSED=waves):
        y, z, h, hb, dz, dln2 = _fused(*ops_, rps, want_zh=True)
    ry, rz, rh, rdz, rdln2 = _unfused(*ops_, rps)
    for name, a, b in (("y", y, ry), ("z", z, rz), ("h", h, rh), ("h (backward)", hb, rh), ("dz", dz, rdz), ("dln2", dln2, rdln2)):
        assert torch.isfinite(a.float()).all(), f"{name}: non-finite or unwritten elements"
        assert torch.equal(a, b), f"fused MLP {name} differs from the unfused launches ({(a.float() - b.float()).abs().max().item():.3e} max)"


@pytest.mark.parametrize("M,C,ff", [(33001, 96, 288), (32801, 96, 160), (33001, 64, 96), (33001, 64, 480)])
def test_fused_mlp_with_an_odd_number_of_32_column_pairs(M, C, ff):
    """ff = 32 x odd: mlp_fused_ok admits it, the default backward (paired stores: two 32-column pairs per step) must not refuse
    it -- the dispatcher hands it to the unpaired variant of the same kernel; bit-identical to the four launches either way."""
    from vtx import _lib
    rps = 49
    ops_ = _operands(M, C, ff, 3 + ff % 11, 0.25, rps)
    assert _lib.load().vtx_mlp_fused_ok(1, 1 << 20, C, ff) == 1
    y, z, h, hb, dz, dln2 = _fused(*ops_, rps, want_zh=True)
    ry, rz, rh, rdz, rdln2 = _unfused(*ops_, rps)
    for name, a, b in (("y", y, ry), ("z", z, rz), ("h", h, rh), ("h (backward)", hb, rh), ("dz", dz, rdz), ("dln2", dln2, rdln2)):
        assert torch.isfinite(a.float()).all(), f"{name}: non-finite or unwritten elements"
        assert torch.equal(a, b), f"fused MLP {name} differs from the unfused launches ({(a.float() - b.float()).abs().max().item():.3e} max)"


def test_fused_mlp_is_bitwise_the_four_launches_at_the_bench_size():
    """the Swin-S stage-1 row count of the benchmark (128 x 3 136 = 401 408 rows, C = 96, ff = 384, per-sample DropPath
    scales), fused against unfused, bit for bit -- and against the fp64 oracle on a strided sample of the rows."""
    from oracle import ref_ops as R
    M, C, ff, rps = 401408, 96, 384, 3136
    ops_ = _operands(M, C, ff, 17, 0.3, rps)
    y, z, h, hb, dz, dln2 = _fused(*ops_, rps, want_zh=True)
    ry, rz, rh, rdz, rdln2 = _unfused(*ops_, rps)
    for name, a, b in (("y", y, ry), ("z", z, rz), ("h", h, rh), ("h (backward)", hb, rh), ("dz", dz, rdz), ("dln2", dln2, rdln2)):
        assert torch.isfinite(a.float()).all(), f"{name}: non-finite or unwritten elements"
        assert torch.equal(a, b), f"fused MLP {name} differs from the unfused launches at M = {M}"
    ln2, x1, dy, w1, b1, w2, b2, s = ops_
    idx = torch.arange(0, M, 97, device=ln2.device)
    f = lambda t: t.detach().double().cpu()
    sr = f(s).repeat_interleave(rps)[:M][idx.cpu(), None]
    a = f(ln2[idx]).requires_grad_(True)
    ref_y = f(x1[idx]) + sr * R.feed_forward(a, f(w1), f(b1), f(w2), f(b2))
    check("fused MLP forward at M = 401 408 vs fp64 oracle (every 97th row)", y[idx], ref_y, 4e-3)
    (ref_dln2,) = torch.autograd.grad(ref_y, a, f(dy[idx]))
    check("fused MLP dln2 at M = 401 408 vs fp64 oracle (every 97th row)", dln2[idx], ref_dln2, 8e-3)


def test_fused_mlp_vs_fp64_oracle():
    from oracle import ref_ops as R
    M, C, ff, rps = 2 * 3136 + 5, 96, 384, 3136
    ln2, x1, dy, w1, b1, w2, b2, _ = _operands(M, C, ff, 11)
    s = torch.tensor([1.25, 0.0, 1.25], device=ln2.device)
    y, _, _, hb, dz, dln2 = _fused(ln2, x1, dy, w1, b1, w2, b2, s, rps)
    f = lambda t: t.detach().double().cpu()
    q = R.bf16_round if hasattr(R, "bf16_round") else (lambda t: t.to(torch.bfloat16).double())
    sr = f(s).repeat_interleave(rps)[:M, None]
    a = f(ln2).requires_grad_(True)
    W1, W2 = f(w1).requires_grad_(True), f(w2).requires_grad_(True)
    mlp = R.feed_forward(a, W1, f(b1), W2, f(b2))
    ref_y = f(x1) + sr * mlp
    check("fused MLP forward vs fp64 oracle", y, ref_y, 4e-3)
    (ref_dln2,) = torch.autograd.grad(ref_y, a, f(dy))
    check("fused MLP dln2 vs fp64 oracle", dln2, ref_dln2, 8e-3)
    zz = R.linear(f(ln2), f(w1), f(b1))
    check("fused MLP h vs fp64 oracle", hb, R.silu(zz), 4e-3)
    sg = torch.sigmoid(zz)
    check("fused MLP dz vs fp64 oracle", dz, sr * (f(dy) @ f(w2)) * (sg * (1 + zz * (1 - sg))), 8e-3)


@pytest.mark.parametrize("dim,ff", [(96, 384), (64, 256)])
def test_swin_layers_with_the_fused_mlp_are_bitwise_the_unfused_ones(dim, ff, monkeypatch):
    """Whole model, stage 1 wide enough for the fused MLP (11 x 3 136 rows): logits and every parameter gradient with MLP_FUSED on equal
    those with it off (one-call layers) and those of the call-by-call path, bit for bit; every torch.empty() buffer NaN-filled."""
    from models import SwinTransformer
    from vtx import functional as VF
    from vtx import options
    from test_gpu_dispatch import _layer_io
    d = dev()
    torch.manual_seed(41)
    model = SwinTransformer(image_size=(224, 224), n_class=16, depths=(2, 2, 2, 2), dims=(dim, 2 * dim, 4 * dim, 8 * dim), dim_head=32,
                            n_heads=(dim // 32, dim // 16, dim // 8, dim // 4), dim_ffs=(ff, 2 * ff, 4 * ff, 8 * ff), window_size=7, drop_path=0.2)
    for m in model.modules():
        if hasattr(m, "rel_pos"):
            torch.nn.init.normal_(m.rel_pos.weight, std=0.3)
    x = torch.randn(11, 3, 224, 224, device=d)
    model.to(d).train()
    monkeypatch.setattr(VF, "_LAYER_CALL", True)
    # every torch.empty() buffer NaN-filled (torch's deterministic-mode debug fill): the fused path leaves the layer's z buffer untouched
    # and writes its h buffer in the backward -- nothing may read them before that
    prev = torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled()
    torch.use_deterministic_algorithms(True, warn_only=True)
    monkeypatch.setattr(torch.utils.deterministic, "fill_uninitialized_memory", True)
    try:
        assert torch.isnan(torch.empty(1024, device=d)).all(), "the debug fill of torch.empty is not active"
        with options.override(MLP_FUSED=1, LN_FOLD=0):       # (the LayerNorm fold regroups the rows of dgamma / dbeta: its own test below)
            out_a, g_a = _layer_io(model, x, True, 78, True)
    finally:
        torch.use_deterministic_algorithms(prev[0], warn_only=prev[1])
    with options.override(MLP_FUSED=812, LN_FOLD=0):
        out_c, g_c = _layer_io(model, x, True, 78, True)
    with options.override(MLP_FUSED=0, LN_FOLD=0):
        out_b, g_b = _layer_io(model, x, True, 78, True)
    assert torch.isfinite(out_a).all()
    for tag, out_o, g_o in (("MLP_FUSED = 0", out_b, g_b), ("MLP_FUSED = 812", out_c, g_c)):
        assert torch.equal(out_a, out_o), f"logits differ from {tag}"
        assert g_a.keys() == g_o.keys() and len(g_a) > 20
        for k in g_a:
            assert torch.isfinite(g_a[k]).all(), k
            assert torch.equal(g_a[k], g_o[k]), f"gradient of {k} differs from {tag}: {(g_a[k] - g_o[k]).abs().max().item():.3e}"
    # the call-by-call path (no fused MLP there): same logits bit for bit; its weight-gradient launches split the tokens into another
    # number of slices at this batch size (stage 2 as well, where nothing is fused): gradients to fp32 summation order
    monkeypatch.setattr(VF, "_LAYER_CALL", False)
    out_d, g_d = _layer_io(model, x, True, 78, True)
    assert torch.equal(out_a, out_d), "logits differ from the call-by-call path"
    for k in g_a:
        err = ((g_a[k] - g_d[k]).norm() / g_d[k].norm().clamp_min(1e-30)).item()
        assert err < 1e-5, f"gradient of {k} differs from the call-by-call path: rel-L2 {err:.3e}"


def test_fused_mlp_timer_records():
    """bench.py's kernel table sees the fused launches under their own names with their algorithmic bytes"""
    from vtx import ops

    class Rec:
        pass
    r = Rec()
    r.tag, r.rows, r.n, r.k, r.flags, r.ms = 15, 401408, 96, 384, 32, 0.1
    name, fl, nb, ms = ops._describe_timer_rec(r)
    assert name == "mlp_fwd_kernel<3, 12, false, false>" and fl == 4.0 * 401408 * 96 * 384 and nb == 2 * 401408 * 3 * 96 + 4 * 96 * 384
    r.tag = 16
    name, fl, nb, ms = ops._describe_timer_rec(r)
    assert name == "mlp_bwd_kernel<3, 4, true, false, 0, true>" and nb == 2 * 401408 * (3 * 96 + 2 * 384) + 4 * 96 * 384


@pytest.mark.parametrize("C,ff", [(96, 384), (64, 512)])
def test_fused_mlp_does_not_depend_on_the_lds_contents(C, ff):
    """both kernels fill their weight images before any wave reads them: zero / NaN / inf LDS fills behind them, same bits
    (the check of tests/test_gpu_lds_poison.py, which also runs them with every torch.empty() buffer NaN-filled)"""
    from test_gpu_lds_poison import _check
    ops_ = _operands(33001, C, ff, 21, 0.2, 49)
    _check(f"fused MLP C = {C}, ff = {ff}", lambda: _fused(*ops_, 49, want_zh=True))


# ------------------------------------------------------------------------------------------------------------ round 6: LayerNorm fold
@pytest.mark.parametrize("M,C,ff,drop", [(34496, 96, 384, 0.0), (32777, 96, 384, 0.25), (401, 96, 384, 0.0), (33001, 96, 288, 0.25),
                                         (37632, 64, 256, 0.25), (33001, 64, 512, 0.1), (33001, 64, 96, 0.0)])
def test_layernorm_backward_folded_into_the_fused_mlp_backward(M, C, ff, drop):
    """vtx_mlp_bwd_ln (option LN_FOLD, round 6) = vtx_mlp_bwd + vtx_layernorm_bwd(dln2, x1, mean, rstd, gamma, dres = dy) in ONE launch:
    h, dz and dx1 bit for bit (the row sums are formed in the stand-alone kernel's association), dgamma / dbeta to fp32 summation order
    (another grouping of the rows) and against fp64.  Buffers NaN-filled: every row and every partial row must be written."""
    from vtx import _lib, ops
    lib = _lib.load()
    rps = 49
    ln2_in, x1, dy, w1, b1, w2, b2, s = _operands(M, C, ff, 7 + M % 5, drop, rps)
    d = x1.device
    g = torch.Generator().manual_seed(3)
    gamma = (1.0 + 0.2 * torch.randn(C, generator=g)).to(d)
    beta = (0.1 * torch.randn(C, generator=g)).to(d)
    ln2, mean, rstd = ops.layernorm_fwd(x1, gamma, beta, 1e-6)
    p = lambda t: None if t is None else t.data_ptr()
    nan = lambda *sh: torch.full(sh, float("nan"), dtype=torch.bfloat16, device=d)
    # stand-alone: fused-MLP backward, then the LayerNorm backward with dy as the residual-stream gradient
    h0, dz0, dln2 = nan(M, ff), nan(M, ff), nan(M, C)
    _lib.check(lib.vtx_mlp_bwd(1, p(ln2), p(dy), p(w1), p(b1), p(w2), p(s), rps, p(h0), p(dz0), p(dln2), M, C, ff, ops._stream()), "vtx_mlp_bwd")
    dx_ref, dg_ref, db_ref = ops.layernorm_bwd(dln2, x1, mean, rstd, gamma, dres=dy)
    # folded
    nb = lib.vtx_layernorm_bwd_blocks(M, C)
    nb = max(nb, lib.vtx_cu_count())
    part = torch.full((nb, 2 * C), float("nan"), dtype=torch.float32, device=d)
    h1, dz1, dx1 = nan(M, ff), nan(M, ff), nan(M, C)
    _lib.check(lib.vtx_mlp_bwd_ln(1, p(ln2), p(dy), p(w1), p(b1), p(w2), p(s), rps, p(h1), p(dz1), p(x1), p(mean), p(rstd), p(gamma),
                                  p(dx1), p(part), nb, M, C, ff, ops._stream()), "vtx_mlp_bwd_ln")
    torch.cuda.synchronize()
    assert torch.isfinite(part).all(), "a dgamma / dbeta partial row was left unwritten"
    for name, a, b in (("h", h1, h0), ("dz", dz1, dz0), ("dx1", dx1, dx_ref)):
        assert torch.isfinite(a.float()).all(), f"{name}: non-finite or unwritten elements"
        assert torch.equal(a, b), f"LayerNorm fold: {name} differs from the two launches ({(a.float() - b.float()).abs().max().item():.3e} max)"
    dg, db = part[:, :C].sum(0), part[:, C:].sum(0)
    check("LayerNorm fold: dgamma vs the stand-alone launch", dg, dg_ref.double().cpu(), 2e-5)
    check("LayerNorm fold: dbeta vs the stand-alone launch", db, db_ref.double().cpu(), 2e-5)
    f = lambda t: t.detach().double().cpu()
    xh = (f(x1) - f(mean)[:, None]) * f(rstd)[:, None]
    check("LayerNorm fold: dgamma vs fp64", dg, (f(dln2) * xh).sum(0), 2e-5)
    check("LayerNorm fold: dbeta vs fp64", db, f(dln2).sum(0), 2e-5)


def test_layernorm_fold_at_the_bench_size_and_in_the_model(monkeypatch):
    """M = 401 408 (Swin-S stage 1 of the benchmark) bit for bit; and a whole model with LN_FOLD on / off: logits, every gradient except
    the folded norms' weight / bias bit for bit, those two to fp32 summation order."""
    from models import SwinTransformer
    from vtx import _lib, ops, options
    from vtx import functional as VF
    from test_gpu_dispatch import _layer_io
    lib = _lib.load()
    M, C, ff, rps = 401408, 96, 384, 3136
    ln2_in, x1, dy, w1, b1, w2, b2, s = _operands(M, C, ff, 19, 0.3, rps)
    d = x1.device
    gamma = (1.0 + 0.2 * torch.randn(C, generator=torch.Generator().manual_seed(5))).to(d)
    ln2, mean, rstd = ops.layernorm_fwd(x1, gamma, torch.zeros_like(gamma), 1e-6)
    p = lambda t: None if t is None else t.data_ptr()
    nan = lambda *sh: torch.full(sh, float("nan"), dtype=torch.bfloat16, device=d)
    h0, dz0, dln2 = nan(M, ff), nan(M, ff), nan(M, C)
    _lib.check(lib.vtx_mlp_bwd(1, p(ln2), p(dy), p(w1), p(b1), p(w2), p(s), rps, p(h0), p(dz0), p(dln2), M, C, ff, ops._stream()), "vtx_mlp_bwd")
    dx_ref, dg_ref, db_ref = ops.layernorm_bwd(dln2, x1, mean, rstd, gamma, dres=dy)
    nb = lib.vtx_layernorm_bwd_blocks(M, C)
    part = torch.full((nb, 2 * C), float("nan"), dtype=torch.float32, device=d)
    h1, dz1, dx1 = nan(M, ff), nan(M, ff), nan(M, C)
    _lib.check(lib.vtx_mlp_bwd_ln(1, p(ln2), p(dy), p(w1), p(b1), p(w2), p(s), rps, p(h1), p(dz1), p(x1), p(mean), p(rstd), p(gamma),
                                  p(dx1), p(part), nb, M, C, ff, ops._stream()), "vtx_mlp_bwd_ln")
    assert torch.equal(dx1, dx_ref) and torch.equal(h1, h0) and torch.equal(dz1, dz0)
    check("LayerNorm fold at M = 401 408: dgamma", part[:, :C].sum(0), dg_ref.double().cpu(), 2e-5)
    del h0, dz0, h1, dz1, dln2
    torch.manual_seed(43)
    model = SwinTransformer(image_size=(224, 224), n_class=16, depths=(2, 2, 2, 2), dims=(96, 192, 384, 768), dim_head=32,
                            n_heads=(3, 6, 12, 24), dim_ffs=(384, 768, 1536, 3072), window_size=7, drop_path=0.2)
    x = torch.randn(11, 3, 224, 224, device=d)
    model.to(d).train()
    monkeypatch.setattr(VF, "_LAYER_CALL", True)
    with options.override(LN_FOLD=1):
        out_a, g_a = _layer_io(model, x, True, 78, True)
    with options.override(LN_FOLD=0):
        out_b, g_b = _layer_io(model, x, True, 78, True)
    assert torch.equal(out_a, out_b)
    folded = [k for k in g_a if k.startswith("block1.") and ".norm_ff." in k]
    assert len(folded) == 4
    for k in g_a:
        if k in folded:
            err = ((g_a[k] - g_b[k]).norm() / g_b[k].norm().clamp_min(1e-30)).item()
            assert err < 1e-5, f"{k}: rel-L2 {err:.3e}"
        else:
            assert torch.equal(g_a[k], g_b[k]), f"gradient of {k} differs with LN_FOLD: {(g_a[k] - g_b[k]).abs().max().item():.3e}"
