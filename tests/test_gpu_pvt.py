"""GPU parity of the PVT path through the C ABI: spatial-reduction attention, patchify
gather, position add vs the oracle; PVT-Small whole model in fp32 vs the reference goldens (G7) and in bf16 vs the
fp32 oracle on a seeded default-init model."""
import ast

import numpy as np
import pytest
import torch

from golden_util import Golden
from gpu_util import TOL, check, dev, report
from oracle import ref_models as M
from oracle import ref_ops as R
from oracle.formula import check_summary, fill, fill_state_dict, name_seed

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]


def _mk(shape, seed, dtype, scale=1.0):
    return fill(shape, seed, scale).to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Lq,Lk,nH,D", [(2, 784, 49, 2, 64), (3, 50, 50, 8, 64), (1, 3136, 49, 1, 64), (2, 196, 49, 5, 64),
                                         (2, 70, 17, 1, 64),
                                         # head dim 32: the global sub-sampled attention of Twins-SVT (twins.py:56-93) at
                                         # its four stage geometries (7 x 7 sub-sampling of 56^2 / 28^2 / 14^2 / 7^2 tokens)
                                         (2, 3136, 64, 2, 32), (2, 784, 16, 4, 32), (3, 196, 4, 8, 32), (2, 49, 1, 16, 32),
                                         (1, 70, 17, 3, 32),
                                         # more than 64 reduced keys (round 5: key-block kernels): PVT at 256 x 256 stage 4 (65), at
                                         # 384 x 384 (144 / 145), Twins at 448 x 448 (256), a ragged odd case
                                         (2, 65, 65, 8, 64), (2, 2304, 144, 2, 64), (1, 145, 145, 8, 64), (2, 1024, 256, 4, 32),
                                         (1, 333, 97, 3, 32)])
def test_sr_attention_core(dtype, B, Lq, Lk, nH, D):
    from vtx import ops
    d = dev()
    C = nH * D
    # random (not formula) data: the key-side gradients are sums over up to 3136 queries, and the smooth sin-hash fill
    # cancels almost exactly there -- the bf16 rounding of P / dS would then be measured against a vanishing signal
    gen = torch.Generator().manual_seed(401)
    q = torch.randn((B, Lq, C), generator=gen).to(dtype)
    kv = torch.randn((B, Lk, 2 * C), generator=gen).to(dtype)
    do = torch.randn((B, Lq, C), generator=gen).to(dtype)
    o, lse = ops.srattn_fwd(q.to(d), kv.to(d), B, Lq, Lk, nH)
    dq, dkv = ops.srattn_bwd(q.to(d), kv.to(d), o, do.to(d), lse, B, Lq, Lk, nH)
    dq2, dkv2 = ops.srattn_bwd(q.to(d), kv.to(d), o, do.to(d), lse, B, Lq, Lk, nH)
    assert torch.equal(dq, dq2) and torch.equal(dkv, dkv2), "sr attention backward is not deterministic"
    qr, kvr = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    orf = R.sr_attention_core(qr, kvr, nH)
    dqr, dkvr = torch.autograd.grad(orf, [qr, kvr], do.double())
    t = TOL[dtype]
    tag = f"{dtype} B{B} Lq{Lq} Lk{Lk} h{nH} d{D}"
    check(f"srattn fwd {tag}", o, orf, t["out"] * 1.5)
    if Lk == 1:
        # one key: softmax = 1, dS = P (dP - sum P dP) vanishes and the true dq is exactly 0; the kernel's two sums come out of
        # different MFMA / shuffle orders, a rounding-sized remainder survives (inputs are O(1))
        assert float(dqr.abs().max()) == 0.0
        assert report(f"srattn |dq| with a single key {tag}", float(dq.abs().max()), 1e-4)
    else:
        check(f"srattn dq {tag}", dq, dqr, 2e-5 if dtype == torch.float32 else 1e-2)
    check(f"srattn dkv {tag}", dkv, dkvr, 2e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,W,C,p,skip", [(28, 28, 128, 4, 0), (14, 14, 320, 2, 0), (8, 12, 64, 2, 1)])
def test_patchify_and_add_pos(dtype, H, W, C, p, skip):
    from vtx import ops
    d = dev()
    B = 3
    x = _mk((B, skip + H * W, C), 411, dtype)
    out = ops.patchify_fwd(x.to(d), B, H, W, C, p, skip)
    ref = x[:, skip:].reshape(B, H // p, p, W // p, p, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, p * p * C)
    assert torch.equal(out.cpu(), ref), "patchify gather is a permutation: must be exact"
    g = _mk(tuple(out.shape), 412, dtype)
    base = _mk(tuple(x.shape), 413, dtype)
    dx = base.to(d).clone()
    ops.patchify_bwd(g.to(d), dx, B, H, W, C, p, skip, accumulate=True)
    back = g.reshape(B, H // p, W // p, p, p, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H * W, C)
    want = base.clone().float()
    want[:, skip:] += back.float()
    check(f"patchify bwd accumulate {dtype}", dx, want, TOL[dtype]["out"])
    # position add with / without cls token
    for has_cls in (False, True):
        T_ = H * W
        xx = _mk((B, T_, C), 414, dtype)
        pos = _mk((T_ + has_cls, C), 415, torch.float32, 0.1)
        cls = _mk((C,), 416, torch.float32, 0.1) if has_cls else None
        o = ops.add_pos_fwd(xx.to(d), None if cls is None else cls.to(d), pos.to(d))
        want = xx.double()
        if has_cls:
            want = torch.cat((cls.double().view(1, 1, C).expand(B, -1, -1), want), 1)
        want = want + pos.double()
        check(f"add_pos fwd cls={has_cls} {dtype}", o, want, TOL[dtype]["out"])
        go = _mk(tuple(o.shape), 417, dtype)
        dxx, dcls, dpos = ops.add_pos_bwd(go.to(d), has_cls)
        check(f"add_pos dx cls={has_cls} {dtype}", dxx, go[:, int(has_cls):].double(), 1e-7)
        check(f"add_pos dpos cls={has_cls} {dtype}", dpos, go.double().sum(0), 1e-5)
        if has_cls:
            check(f"add_pos dcls {dtype}", dcls, go[:, 0].double().sum(0), 1e-5)


def _pvt(drop_path=0.0):
    from models.pvt import PyramidVisionTransformer
    return PyramidVisionTransformer(**M.PVT_SMALL, drop_path=drop_path)


def test_pvt_small_fp32_vs_reference():
    """fp32 parity mode vs the reference's own outputs (golden G7): logits and all 0f the per-parameter grad norms."""
    g = Golden("g7_pvt")
    model = _pvt()
    model.load_state_dict(fill_state_dict(model.state_dict()))
    model.to(dev()).train()
    x = fill((2, 3, 224, 224), 21, 1.0).to(dev())
    out = model(x)
    e = check_summary(out, g.rec("pvt_small.train64.logits"), 1e-4, "pvt logits vs reference fp64")
    report("pvt-small fp32 logits vs reference (fp64 golden)", e, 1e-4)
    cot = fill(out.shape, name_seed("pvt_small.train64.cot"), 1.0).to(dev())
    (out * cot).sum().backward()
    names = [str(n) for n in g.arr("pvt_small.train64.grad_names")]
    norms = g.arr("pvt_small.train64.grad_norms")
    got = dict(model.named_parameters())
    assert names == list(got.keys())
    worst, worst_n = 0.0, ""
    for n, ref in zip(names, norms):
        rel = abs(got[n].grad.double().norm().item() - ref) / max(ref, 1e-12)
        if rel > worst:
            worst, worst_n = rel, n
    # fp32 through 16 layers vs an fp64 reference: the oracle's own fp32 run deviates up to ~2e-3 on the deepest small
    # gradients (tests/test_oracle_pvt.py uses 5e-3 for the same comparison)
    assert report(f"pvt-small fp32: worst per-param grad-norm deviation ({worst_n})", worst, 5e-3)
    for k in g.keys("pvt_small.train64.grad."):                       # sampled gradient values (same cotangent)
        pn = k[len("pvt_small.train64.grad."):]
        e = check_summary(got[pn].grad, g.rec(k), 5e-3, k)
        report(f"pvt-small fp32: grad {pn}", e, 5e-3)


@pytest.mark.parametrize("size", [256, 384])
def test_pvt_at_other_resolutions_fp32_vs_oracle(size):
    """256 x 256 (stage 4: 64 + 1 = 65 keys) and 384 x 384 (144 reduced keys in stages 1-3, 145 in stage 4): beyond the 64 keys the
    register-resident kernel holds -- the key-block kernels behind vtx_srattn_* (reference models/pvt.py:38-66 runs at any size).
    One layer per stage, fp32 parity mode: logits and every parameter gradient vs the CPU oracle; the standalone module's score too."""
    from models import PyramidVisionTransformer
    from test_gpu_models import _seeded_init
    cfg = dict(M.PVT_SMALL, image_size=size, depths=(1, 1, 1, 1), n_class=10)
    model = PyramidVisionTransformer(**cfg)
    sd = _seeded_init(model, 8)
    model.to(dev()).train()
    x = torch.randn(2, 3, size, size, generator=torch.Generator().manual_seed(18))
    out = model(x.to(dev()))
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = M.pvt_forward(P, x, cfg)
    check(f"pvt {size}x{size} fp32 logits vs oracle", out, ref, 1e-4)
    cot = torch.randn(ref.shape, generator=torch.Generator().manual_seed(19))
    (out * cot.to(dev())).sum().backward()
    names = [n for n, _ in model.named_parameters()]
    rg = torch.autograd.grad((ref * cot).sum(), [P[n] for n in names])
    got = dict(model.named_parameters())
    for n, r in zip(names, rg):
        check(f"pvt {size}x{size} fp32 grad {n}", got[n].grad, r, 2e-3)
    # the standalone module (returns the pre-softmax score) at 144 keys
    import models.pvt as PV
    attn = PV.MultiHeadedAttention(128, 2, reduction=2).to(dev())
    xa = torch.randn(2, 24 * 24, 128, generator=torch.Generator().manual_seed(20))
    o, score = attn(xa.to(dev()), 24, 24)
    assert score.shape == (2, 2, 576, 144)
    Pa = {k: v.detach().cpu().double() for k, v in attn.state_dict().items()}
    check("pvt attn standalone 144 keys out", o, R.pvt_attention(xa.double(), 24, 24, Pa, 2, 2), 2e-5)


def test_pvt_small_bf16_autocast_vs_oracle():
    from test_gpu_models import _bf16_vs_oracle, _seeded_init
    model = _pvt()
    sd = _seeded_init(model, 2)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(12))
    _bf16_vs_oracle(model, lambda P, xx: M.pvt_forward(P, xx, M.PVT_SMALL), sd, x, "pvt-small")


def test_pvt_small_drop_path_runs_and_is_deterministic_given_masks(monkeypatch):
    """DropPath through the PVT layer (per-sample scale in the GEMM epilogues / dropped rows in wgrad) vs the oracle
    with the same injected masks."""
    from test_gpu_models import _seeded_init
    model = _pvt(0.2)
    sd = _seeded_init(model, 3)
    model.to(dev()).train()
    B = 4
    gen = torch.Generator().manual_seed(5)
    rates = torch.linspace(0, 0.2, sum(M.PVT_SMALL["depths"])).tolist()
    masks = [(None, None) if r == 0 else ((torch.rand(B, generator=gen) < 1 - r).float(), (torch.rand(B, generator=gen) < 1 - r).float())
             for r in rates]
    it = iter([m for pair in masks for m in pair if m is not None])

    def fake(p, training, batch, device):
        if not training or p == 0:
            return None
        return next(it).to(device) / (1.0 - p)

    monkeypatch.setattr("models.pvt.drop_path_scale", fake)
    x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(13))
    out = model(x.to(dev()))
    P = {k: v.clone() for k, v in sd.items()}
    ref = M.pvt_forward(P, x, M.PVT_SMALL, drop_masks=masks, drop_path=0.2)
    check("pvt-small fp32 drop-path logits vs oracle", out, ref, 1e-4)


@pytest.mark.parametrize("reduction,height,width,n_head", [(4, 28, 28, 2), (1, 7, 7, 8), (2, 14, 14, 5)])
def test_pvt_attention_standalone_forward_returns_out_and_score(reduction, height, width, n_head):
    """models.pvt.MultiHeadedAttention.forward(input, height, width) -> (out, score) as in the reference (pvt.py:31-69):
    out and the parameter / input gradients vs the oracle's pvt_attention, score vs q k^T / sqrt(d) (fp32 parity mode)."""
    from models.pvt import MultiHeadedAttention
    d = dev()
    dim = n_head * 64
    torch.manual_seed(3)
    attn = MultiHeadedAttention(dim, n_head, reduction=reduction)
    sd = {k: v.detach().clone() for k, v in attn.state_dict().items()}
    attn.to(d).train()
    gen = torch.Generator().manual_seed(4)
    B, L = 2, height * width
    x = torch.randn(B, L, dim, generator=gen)
    cot = torch.randn(B, L, dim, generator=gen)
    xg = x.to(d).requires_grad_(True)
    out, score = attn(xg, height, width)
    (out * cot.to(d)).sum().backward()
    P = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    xr = x.double().requires_grad_(True)
    ref = R.pvt_attention(xr, height, width, P, n_head, reduction)
    names = list(P.keys())
    grads = torch.autograd.grad((ref * cot.double()).sum(), [xr] + [P[n] for n in names])
    check(f"pvt attn standalone out r{reduction}", out, ref, 2e-5)
    check(f"pvt attn standalone dx r{reduction}", xg.grad, grads[0], 5e-5)
    got = dict(attn.named_parameters())
    for n, g in zip(names, grads[1:]):
        check(f"pvt attn standalone grad {n} r{reduction}", got[n].grad, g, 1e-4)
    # score = (B, heads, L, Lk) pre-softmax
    q = R.linear(xr, P["linear_q.weight"], None)
    kvin = xr if reduction == 1 else R.pvt_reduce(xr, height, width, P["reduce_conv.weight"], P["reduce_conv.bias"],
                                                  P["reduce_norm.weight"], P["reduce_norm.bias"], reduction)
    k = R.linear(kvin, P["linear_kv.weight"], None)[..., :dim]
    sref = torch.einsum("bihd,bjhd->bhij", q.view(B, L, n_head, 64), k.view(B, -1, n_head, 64)) / 8.0
    assert score.shape == sref.shape
    check(f"pvt attn standalone score r{reduction}", score, sref, 2e-5)
    with pytest.raises(NotImplementedError):
        attn(xg, height, width, prev=score)


@pytest.mark.parametrize("family", ["pvt_small", "twins_svt_s"])
def test_sr_layers_through_one_c_call_are_bitwise_the_call_by_call_path(monkeypatch, family):
    """vtx_srlayer_fwd / bwd (one C call per PVT block / Twins global half) enqueue the launches of the call-by-call path in its
    order with its arguments: logits and every gradient of a bf16 step with DropPath agree BIT FOR BIT, with and without the
    side stream; and the one-call path really is the one taken."""
    from vtx import functional as VF
    if not (VF._LAYER_CALL and VF._DEFER_REDUCE):
        pytest.skip("the one-call layers are switched off in this process (VTX_LAYER_CALL=0 / VTX_DEFER_REDUCE=0)")
    if family == "pvt_small":
        model = _pvt(0.1)
    else:
        from models.twins import TwinsSVT
        model = TwinsSVT(**M.TWINS_SVT_S, drop_path=0.1)
    model.to(dev()).train()
    B = 16 if family == "pvt_small" else 64          # (64: the Twins late stages take their split-K path, 64 * 1 rows)
    x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(21)).to(dev())
    calls = []
    real = VF.PvtLayerFn._forward_one_call

    def spy(ctx, *a):
        y = real(ctx, *a)
        calls.append(y is not None)
        return y

    monkeypatch.setattr(VF.PvtLayerFn, "_forward_one_call", staticmethod(spy))
    res = {}
    from vtx import options
    # (round 6: with option LN_FOLD the one-call layers run the norm_ff backward inside the fused-MLP backward -- another grouping of the rows
    #  of dgamma / dbeta; that path has its own tests, tests/test_gpu_mlp_fused.py and test_gpu_ln_fold.py.  Here: the launch-for-launch contract.)
    with options.override(LN_FOLD=0):
        for one_call in (True, False):
            for side in (True, False):
                monkeypatch.setattr(VF, "_LAYER_CALL", one_call)
                torch.manual_seed(5)
                model.zero_grad(set_to_none=True)
                with VF.deferred_wgrad(side):
                    with torch.autocast("cuda", dtype=torch.bfloat16):
                        out = model(x)
                    out.float().square().mean().backward()
                    VF.side_join()
                res[(one_call, side)] = (out.clone(), [p.grad.clone() for p in model.parameters()])
    n_layers = sum(M.PVT_SMALL["depths"]) if family == "pvt_small" else sum(M.TWINS_SVT_S["depths"])
    assert calls == [True] * (2 * n_layers), f"one-call path not taken on every layer: {calls}"
    ref = res[(False, False)]
    assert torch.isfinite(ref[0].float()).all()
    for key, (out, grads) in res.items():
        assert torch.equal(out, ref[0]), f"logits differ {key}"
        for (n, _), a, b in zip(model.named_parameters(), grads, ref[1]):
            assert torch.equal(a, b), f"gradient of {n} differs {key}"
