"""-m gpu: LayerNorm backward folded into the row-streaming launch that produces its input gradient (round 6, option LN_FOLD; VERDICT r5
item 2, built): csrc/ln_fold.h used by mlp_bwd_kernel<.., LNB> (tests/test_gpu_mlp_fused.py) and by dgrad_ln_kernel (csrc/gemm_skinny.hip:
the qkv input gradient of a narrow window-attention layer + the norm_attn backward, reference models/swin_transformer.py:128,193-194).
The fold must reproduce the two launches it replaces: dx bit for bit, dgamma / dbeta to fp32 summation order and against fp64."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vision-transformers-pytorch_amd"))

from gpu_util import check, dev          # noqa: E402

pytestmark = pytest.mark.gpu


def _case(M, C, K, seed):
    d = dev()
    g = torch.Generator().manual_seed(seed)
    bf = lambda *s, std=1.0: (torch.randn(*s, generator=g) * std).to(torch.bfloat16).to(d)
    dy, x, dres = bf(M, K, std=0.05), bf(M, C), bf(M, C, std=0.05)
    w = bf(K, C, std=C ** -0.5)                      # the layer's weight [out = K][in = C]; the dgrad multiplies by it: dln = dy . w
    gamma = (1.0 + 0.2 * torch.randn(C, generator=g)).to(d)
    beta = (0.1 * torch.randn(C, generator=g)).to(d)
    return dy, x, dres, w, gamma, beta


@pytest.mark.parametrize("M,C,K", [(34496, 96, 288), (32777, 96, 288), (401, 96, 288), (33001, 64, 192), (32801, 128, 384), (33001, 96, 96),
                                   (33001, 64, 64), (32801, 128, 128)])
def test_dgrad_with_the_layernorm_backward_in_its_epilogue(M, C, K):
    from vtx import _lib, ops
    lib = _lib.load()
    dy, x, dres, w, gamma, beta = _case(M, C, K, 11 + M % 7)
    d = x.device
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6)
    wt = w.t().contiguous()                           # [C][K]: the transposed bf16 copy a weight scope keeps for the dgrads
    dln = ops.gemm(dy, wt, 0)                         # what the layer call launches (forward-layout kernel on the transposed copy)
    dln2 = ops.gemm(dy, w, 1)                         # (the register-staged NN kernel on W itself must agree bit for bit)
    assert torch.equal(dln, dln2)
    dx_ref, dg_ref, db_ref = ops.layernorm_bwd(dln, x, mean, rstd, gamma, dres=dres)
    p = lambda t: t.data_ptr()
    nb = max(lib.vtx_layernorm_bwd_blocks(M, C), lib.vtx_cu_count())
    part = torch.full((nb, 2 * C), float("nan"), dtype=torch.float32, device=d)
    dx = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=d)
    _lib.check(lib.vtx_dgrad_ln(1, p(dy), p(wt), p(x), p(mean), p(rstd), p(gamma), p(dres), p(dx), p(part), nb, M, C, K, ops._stream()), "vtx_dgrad_ln")
    torch.cuda.synchronize()
    assert torch.isfinite(part).all() and torch.isfinite(dx.float()).all()
    assert torch.equal(dx, dx_ref), f"dgrad + LayerNorm fold: dx differs from the two launches ({(dx.float() - dx_ref.float()).abs().max().item():.3e} max)"
    dg, db = part[:, :C].sum(0), part[:, C:].sum(0)
    check("dgrad + LayerNorm fold: dgamma vs the stand-alone launch", dg, dg_ref.double().cpu(), 2e-5)
    check("dgrad + LayerNorm fold: dbeta vs the stand-alone launch", db, db_ref.double().cpu(), 2e-5)
    f = lambda t: t.detach().double().cpu()
    xh = (f(x) - f(mean)[:, None]) * f(rstd)[:, None]
    check("dgrad + LayerNorm fold: dgamma vs fp64", dg, (f(dln) * xh).sum(0), 2e-5)
    # the whole thing against fp64 (dln rounded to bf16 where the product rounds it)
    dl64 = (f(dy) @ f(w)).to(torch.bfloat16).double()
    gv = dl64 * f(gamma)
    ref = f(dres) + f(rstd)[:, None] * (gv - gv.mean(1, keepdim=True) - xh * (gv * xh).mean(1, keepdim=True))
    check("dgrad + LayerNorm fold: dx vs fp64", dx, ref, 4e-3)


def test_dgrad_layernorm_fold_at_the_bench_size_and_in_the_model(monkeypatch):
    """M = 401 408 (Swin-S stage 1 of the benchmark) bit for bit; a whole Swin model with LN_FOLD = 3 / 1 / 0: logits and every gradient
    except the folded norms' weight / bias bit for bit, those to fp32 summation order."""
    from models import SwinTransformer
    from vtx import _lib, ops, options
    from vtx import functional as VF
    from test_gpu_dispatch import _layer_io
    lib = _lib.load()
    M, C, K = 401408, 96, 288
    dy, x, dres, w, gamma, beta = _case(M, C, K, 23)
    d = x.device
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6)
    wt = w.t().contiguous()
    dln = ops.gemm(dy, wt, 0)
    dx_ref, dg_ref, db_ref = ops.layernorm_bwd(dln, x, mean, rstd, gamma, dres=dres)
    p = lambda t: t.data_ptr()
    nb = lib.vtx_layernorm_bwd_blocks(M, C)
    part = torch.full((nb, 2 * C), float("nan"), dtype=torch.float32, device=d)
    dx = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=d)
    _lib.check(lib.vtx_dgrad_ln(1, p(dy), p(wt), p(x), p(mean), p(rstd), p(gamma), p(dres), p(dx), p(part), nb, M, C, K, ops._stream()), "vtx_dgrad_ln")
    assert torch.equal(dx, dx_ref)
    check("dgrad + LayerNorm fold at M = 401 408: dgamma", part[:, :C].sum(0), dg_ref.double().cpu(), 2e-5)
    del dy, x, dres, dln, dx, dx_ref
    torch.manual_seed(45)
    model = SwinTransformer(image_size=(224, 224), n_class=16, depths=(2, 2, 2, 2), dims=(96, 192, 384, 768), dim_head=32,
                            n_heads=(3, 6, 12, 24), dim_ffs=(384, 768, 1536, 3072), window_size=7, drop_path=0.2)
    for m in model.modules():
        if hasattr(m, "rel_pos"):
            torch.nn.init.normal_(m.rel_pos.weight, std=0.3)
    xin = torch.randn(11, 3, 224, 224, device=d)
    model.to(d).train()
    monkeypatch.setattr(VF, "_LAYER_CALL", True)
    res = {}
    for v in (15, 12, 3, 1, 0):
        with options.override(LN_FOLD=v):
            res[v] = _layer_io(model, xin, True, 78, True)
    # the FORWARD folds (bits 2, 3) alone change no bit anywhere
    assert torch.equal(res[12][0], res[0][0])
    for k, g0 in res[0][1].items():
        assert torch.equal(res[12][1][k], g0), f"LN_FOLD = 12 (forward folds only): gradient of {k} differs"
    folded = [k for k in res[0][1] if k.startswith("block1.") and (".norm_ff." in k or ".norm_attn." in k)]
    assert len(folded) == 8
    for v in (15, 3, 1):
        assert torch.equal(res[v][0], res[0][0])
        for k, g0 in res[0][1].items():
            g = res[v][1][k]
            if k in folded:
                err = ((g - g0).norm() / g0.norm().clamp_min(1e-30)).item()
                assert err < 1e-5, f"LN_FOLD = {v}, {k}: rel-L2 {err:.3e}"
            else:
                assert torch.equal(g, g0), f"LN_FOLD = {v}: gradient of {k} differs: {(g - g0).abs().max().item():.3e}"


@pytest.mark.parametrize("M,C,ff,drop", [(34496, 96, 384, 0.0), (32777, 96, 384, 0.25), (401, 96, 384, 0.0), (33001, 96, 288, 0.25),
                                         (37632, 64, 256, 0.25), (33001, 64, 512, 0.1)])
def test_layernorm_forward_on_the_row_operands_of_the_fused_mlp(M, C, ff, drop):
    """vtx_mlp_fwd_ln = vtx_layernorm_fwd(x1) + vtx_mlp_fwd(ln2, resid = x1) in one launch: ln2, mean, rstd and y bit for bit."""
    from vtx import _lib, ops
    from test_gpu_mlp_fused import _operands
    lib = _lib.load()
    rps = 49
    _, x1, _, w1, b1, w2, b2, s = _operands(M, C, ff, 9 + M % 5, drop, rps)
    d = x1.device
    g = torch.Generator().manual_seed(4)
    gamma = (1.0 + 0.2 * torch.randn(C, generator=g)).to(d)
    beta = (0.1 * torch.randn(C, generator=g)).to(d)
    p = lambda t: None if t is None else t.data_ptr()
    ln_ref, mean_ref, rstd_ref = ops.layernorm_fwd(x1, gamma, beta, 1e-6)
    y_ref = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=d)
    _lib.check(lib.vtx_mlp_fwd(1, p(ln_ref), p(w1), p(b1), p(w2), p(b2), p(x1), p(s), rps, p(y_ref), None, None, M, C, ff, ops._stream()), "vtx_mlp_fwd")
    ln = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=d)
    y = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=d)
    mean = torch.full((M,), float("nan"), device=d)
    rstd = torch.full((M,), float("nan"), device=d)
    _lib.check(lib.vtx_mlp_fwd_ln(1, p(x1), p(gamma), p(beta), 1e-6, p(ln), p(mean), p(rstd), p(w1), p(b1), p(w2), p(b2), p(s), rps, p(y), M, C, ff,
                                  ops._stream()), "vtx_mlp_fwd_ln")
    for name, a, b in (("ln2", ln, ln_ref), ("mean", mean, mean_ref), ("rstd", rstd, rstd_ref), ("y", y, y_ref)):
        assert torch.isfinite(a.float()).all(), f"{name}: non-finite or unwritten elements"
        assert torch.equal(a, b), f"LayerNorm forward fold (MLP): {name} differs ({(a.float() - b.float()).abs().max().item():.3e} max)"


@pytest.mark.parametrize("M,C,N", [(34496, 96, 288), (32777, 96, 288), (401, 96, 288), (33001, 64, 192), (32801, 128, 384), (33001, 64, 64),
                                   (401408, 96, 288)])
def test_layernorm_forward_on_the_row_operands_of_the_streaming_gemm(M, C, N):
    """vtx_ln_gemm = vtx_layernorm_fwd(x) + vtx_gemm(ln, w, bias) in one launch: ln, mean, rstd and the product bit for bit."""
    from vtx import _lib, ops
    lib = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(13 + M % 3)
    bf = lambda *sh, std=1.0: (torch.randn(*sh, generator=g) * std).to(torch.bfloat16).to(d)
    x, w = bf(M, C), bf(N, C, std=C ** -0.5)
    bias = (0.2 * torch.randn(N, generator=g)).to(d)
    gamma = (1.0 + 0.2 * torch.randn(C, generator=g)).to(d)
    beta = (0.1 * torch.randn(C, generator=g)).to(d)
    p = lambda t: t.data_ptr()
    ln_ref, mean_ref, rstd_ref = ops.layernorm_fwd(x, gamma, beta, 1e-6)
    y_ref = ops.gemm(ln_ref, w, 0, bias=bias)
    ln = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=d)
    y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=d)
    mean = torch.full((M,), float("nan"), device=d)
    rstd = torch.full((M,), float("nan"), device=d)
    _lib.check(lib.vtx_ln_gemm(1, p(x), p(gamma), p(beta), 1e-6, p(ln), p(mean), p(rstd), p(w), p(bias), p(y), M, C, N, ops._stream()), "vtx_ln_gemm")
    for name, a, b in (("ln", ln, ln_ref), ("mean", mean, mean_ref), ("rstd", rstd, rstd_ref), ("y", y, y_ref)):
        assert torch.isfinite(a.float()).all(), f"{name}: non-finite or unwritten elements"
        assert torch.equal(a, b), f"LayerNorm forward fold (GEMM): {name} differs ({(a.float() - b.float()).abs().max().item():.3e} max)"
