"""GPU parity of the attention-probability dropout (reference models/vit.py:39, models/swin_transformer.py:144, models/pvt.py:60,
models/twins.py:88,147) through the C ABI.

* every attention module of the five families in fp32, train mode, with the keep mask the REFERENCE run drew (golden G11): output,
  input gradient and every parameter gradient vs the reference's;
* the product's own mask (counter-based hash regenerated in the backward): identical bits to the same call with the hash's
  decisions passed as an explicit mask, keep rate = 1 - p, new mask per call, reproducible under torch.manual_seed;
* drop_attn = 0.1 trains in all five model families (forward + backward finite, eval mode unaffected)."""
import pytest
import torch

import attn_dropout_cases as ADC
from golden_util import Golden
from gpu_util import dev, report
from oracle.formula import check_summary, fill, fill_state_dict, name_seed

pytestmark = pytest.mark.gpu


def _module(name):
    import models.pvt as PV
    import models.swin_transformer as SW
    import models.twins as TW
    import models.vit as VT
    P = ADC.P_DROP
    return {
        "vit_L37": lambda: (VT.MultiHeadedAttention(128, 2, dropout=P), lambda m, x, k: m(x, keep=k)),
        "vit_L197": lambda: (VT.MultiHeadedAttention(128, 2, dropout=P), lambda m, x, k: m(x, keep=k)),
        "swin_s1": lambda: (SW.MultiHeadedLocalAttention(64, 2, 32, (14, 14), 7, True, P), lambda m, x, k: m(x, keep=k)),
        "swin_s0": lambda: (SW.MultiHeadedLocalAttention(64, 2, 32, (14, 14), 7, False, P), lambda m, x, k: m(x, keep=k)),
        "pvt_r2": lambda: (PV.MultiHeadedAttention(128, 2, reduction=2, dropout=P), lambda m, x, k: m(x, 8, 8, keep=k)[0]),
        "pvt_r1_cls": lambda: (PV.MultiHeadedAttention(128, 2, reduction=1, dropout=P), lambda m, x, k: m(x, 4, 4, keep=k)[0]),
        "twins_local": lambda: (TW.MultiHeadedLocalAttention(64, 2, 32, 7, P), lambda m, x, k: m(x, keep=k)),
        "twins_global": lambda: (TW.MultiHeadedAttention(64, 2, reduction=7, dropout=P), lambda m, x, k: m(x, keep=k)),
        "halo_w7a3": lambda: (__import__("models.halo_transformer", fromlist=["x"]).MultiHeadedHaloAttention(64, 2, 32, 7, 3, P),
                              lambda m, x, k: m(x, keep=k)),
    }[name]()


def _load(mod):
    mod.load_state_dict(fill_state_dict(mod.state_dict()))
    return mod.to(dev()).train()


@pytest.mark.parametrize("name", sorted(ADC.CASES))
def test_modules_fp32_with_the_reference_keep_mask(name):
    g = Golden("g11_attn_dropout")
    mod, call = _module(name)
    _load(mod)
    keep = ADC.keep_mask(g, name)
    keep_d = ADC.kernel_order(name, keep).to(dev())          # [problems, Lq, Lk], the kernels' order
    x = ADC.case_input(name, torch.float32).to(dev()).requires_grad_(True)
    out = call(mod, x, keep_d)
    tol = 2e-4
    e = check_summary(out, g.rec(f"{name}.out"), tol, f"{name} out")
    report(f"attention dropout {name} fp32 out vs reference (recorded mask)", e, tol)
    (out * fill(out.shape, name_seed(name + ".cot"), 1.0).to(dev())).sum().backward()
    e = check_summary(x.grad, g.rec(f"{name}.dx"), 1e-3, f"{name} dx")
    report(f"attention dropout {name} fp32 dx vs reference", e, 1e-3)
    for n, p in mod.named_parameters():
        e = check_summary(p.grad, g.rec(f"{name}.d.{n}"), 1e-3, f"{name} {n}")
        report(f"attention dropout {name} fp32 grad {n} vs reference", e, 1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["global", "window", "sr", "sr_long", "global_long", "cross_bias"])
def test_hashed_mask_is_the_exported_mask_and_the_backward_regenerates_it(kind, dtype):
    """The kernels' hash decisions == vtx_attn_keep_mask's; forward + backward with the hash are bit-identical to forward +
    backward with that mask passed explicitly (so the backward regenerates exactly the forward's mask)."""
    from vtx import ops
    d = dev()
    gen = torch.Generator().manual_seed(7)
    p, seed = 0.2, 0x1234ABCD5678
    if kind == "cross_bias":                       # halo attention's core: a score bias, its gradient, 169 keys
        B, Lq, Lk, nH, D = 6, 49, 169, 2, 32
        q = torch.randn(B * Lq, nH * D, generator=gen).to(dtype).to(d)
        kv = torch.randn(B * Lk, 2 * nH * D, generator=gen).to(dtype).to(d)
        do = torch.randn(B * Lq, nH * D, generator=gen).to(dtype).to(d)
        bias = torch.randn(nH, Lq, Lk, generator=gen).to(d)
        mask = ops.attn_keep_mask(B * nH, Lq, Lk, p, seed, d)
        o1, l1 = ops.xattn_fwd(q, kv, B, Lq, Lk, nH, bias, drop=(p, seed, None))
        o2, l2 = ops.xattn_fwd(q, kv, B, Lq, Lk, nH, bias, drop=(p, seed, mask))
        g1 = ops.xattn_bwd(q, kv, o1, do, l1, B, Lq, Lk, nH, bias, drop=(p, seed, None))
        g2 = ops.xattn_bwd(q, kv, o2, do, l2, B, Lq, Lk, nH, bias, drop=(p, seed, mask))
        o0, _ = ops.xattn_fwd(q, kv, B, Lq, Lk, nH, bias)
        # and against the oracle with the exported mask
        from oracle import ref_ops as R
        qr, kvr, br = q.double().cpu().requires_grad_(True), kv.double().cpu().requires_grad_(True), bias.double().cpu().requires_grad_(True)
        Q = qr.view(B, Lq, nH, D)
        K, V = kvr.view(B, Lk, 2, nH, D)[:, :, 0], kvr.view(B, Lk, 2, nH, D)[:, :, 1]
        P = R.attn_dropout(torch.softmax(torch.einsum("bqhd,bkhd->bhqk", Q, K) / D ** 0.5 + br, -1), mask.cpu().view(B, nH, Lq, Lk), p)
        oref = torch.einsum("bhqk,bkhd->bqhd", P, V).reshape(B * Lq, nH * D)
        gq, gkv, gb = torch.autograd.grad(oref, [qr, kvr, br], do.double().cpu())
        from gpu_util import check
        tol = 2e-5 if dtype == torch.float32 else 1e-2
        check(f"xattn dropout fwd vs oracle {dtype}", o1, oref, 6e-3 if dtype == torch.bfloat16 else 3e-5)
        check(f"xattn dropout dq vs oracle {dtype}", g1[0], gq, tol)
        check(f"xattn dropout dkv vs oracle {dtype}", g1[1], gkv, tol)
        check(f"xattn dropout dbias vs oracle {dtype}", g1[2], gb, tol)
    elif kind in ("sr", "sr_long"):
        B, Lq, Lk, nH, D = (3, 200, 49, 2, 64) if kind == "sr" else (2, 300, 145, 2, 64)
        q = torch.randn(B * Lq, nH * D, generator=gen).to(dtype).to(d)
        kv = torch.randn(B * Lk, 2 * nH * D, generator=gen).to(dtype).to(d)
        do = torch.randn(B * Lq, nH * D, generator=gen).to(dtype).to(d)
        mask = ops.attn_keep_mask(B * nH, Lq, Lk, p, seed, d)
        o1, l1 = ops.srattn_fwd(q, kv, B, Lq, Lk, nH, drop=(p, seed, None))
        o2, l2 = ops.srattn_fwd(q, kv, B, Lq, Lk, nH, drop=(p, seed, mask))
        g1 = ops.srattn_bwd(q, kv, o1, do, l1, B, Lq, Lk, nH, drop=(p, seed, None))
        g2 = ops.srattn_bwd(q, kv, o2, do, l2, B, Lq, Lk, nH, drop=(p, seed, mask))
        o0, _ = ops.srattn_fwd(q, kv, B, Lq, Lk, nH)
    else:
        if kind in ("global", "global_long"):
            B, L, nH, D, swin = 3, (197 if kind == "global" else 577), 2, 64, None
            rows, nW = B * L, 1
        else:
            B, L, nH, D, swin = 2, 49, 3, 32, (14, 14, 7, False)
            rows, nW = B * 14 * 14, 4
        qkv = torch.randn(rows, 3 * nH * D, generator=gen).to(dtype).to(d)
        do = torch.randn(rows, nH * D, generator=gen).to(dtype).to(d)
        mask = ops.attn_keep_mask(B * nW * nH, L, L, p, seed, d)
        o1, l1 = ops.attention_fwd(qkv, B, L, nH, D, swin=swin, drop=(p, seed, None))
        o2, l2 = ops.attention_fwd(qkv, B, L, nH, D, swin=swin, drop=(p, seed, mask))
        g1 = ops.attention_bwd(qkv, o1, do, l1, B, L, nH, D, swin=swin, drop=(p, seed, None))[:1]
        g2 = ops.attention_bwd(qkv, o2, do, l2, B, L, nH, D, swin=swin, drop=(p, seed, mask))[:1]
        o0, _ = ops.attention_fwd(qkv, B, L, nH, D, swin=swin)
    torch.cuda.synchronize()
    rate = mask.float().mean().item()
    assert abs(rate - (1 - p)) < 0.01, f"keep rate {rate}"
    assert torch.equal(o1, o2) and torch.equal(l1, l2), "hashed forward != forward with the exported mask"
    for a, b in zip(g1, g2):
        assert torch.equal(a, b), "hashed backward != backward with the exported mask"
    assert not torch.equal(o1, o0), "dropout changed nothing"
    other = ops.attn_keep_mask(mask.shape[0], mask.shape[1], mask.shape[2], p, seed + 1, d)
    assert (other != mask).float().mean().item() > 0.2, "another seed gives (nearly) the same mask"


def test_seeds_advance_per_call_and_follow_torch_manual_seed():
    from vtx import functional as VF
    torch.manual_seed(11)
    a = [VF.attn_drop(0.1, True)[1] for _ in range(3)]
    assert len(set(a)) == 3
    assert VF.attn_drop(0.1, False) is None and VF.attn_drop(0.0, True) is None
    import models.vit as VT
    m = _load(VT.MultiHeadedAttention(128, 2, dropout=0.3))
    x = fill((2, 37, 128), 5, 1.0).to(dev())
    y1, y2 = m(x), m(x)
    assert not torch.equal(y1, y2), "two calls drew the same mask"
    m.eval()
    assert torch.equal(m(x), m(x)), "eval mode must not drop"


@pytest.mark.parametrize("family", ["vit", "swin", "pvt", "twins", "dino"])
def test_drop_attn_trains_in_every_family(family):
    """drop_attn = 0.1: a train step's forward + backward runs through the dropout kernels (finite outputs and gradients, differs
    from the drop_attn = 0 forward of the same weights); eval mode is the drop-free model."""
    d = dev()
    torch.manual_seed(3)
    if family == "vit" or family == "dino":
        from models.vit import VisionTransformer
        mk = lambda pa: VisionTransformer(None, 224, 16, 2, 384, 6, 1536, 0.0, pa, 0.0, 0.0)
        x = fill((2, 3, 224, 224), 9, 1.0)
    elif family == "swin":
        from models import SwinTransformer
        mk = lambda pa: SwinTransformer(image_size=(224, 224), n_class=10, depths=(1, 1, 2, 1), dims=(96, 192, 384, 768), dim_head=32,
                                        n_heads=(3, 6, 12, 24), dim_ffs=(384, 768, 1536, 3072), window_size=7, drop_attn=pa)
        x = fill((2, 3, 224, 224), 9, 1.0)
    elif family == "pvt":
        from models import PyramidVisionTransformer
        from oracle import ref_models as M
        cfg = dict(M.PVT_SMALL)
        cfg["depths"] = (1, 1, 1, 1)
        mk = lambda pa: PyramidVisionTransformer(**cfg, drop_attn=pa)
        x = fill((2, 3, 224, 224), 9, 1.0)
    else:
        from models.twins import TwinsSVT
        from oracle import ref_models as M
        cfg = dict(M.TWINS_SVT_S)
        cfg["depths"] = (1, 1, 1, 1)
        mk = lambda pa: TwinsSVT(**cfg, drop_attn=pa)
        x = fill((2, 3, 224, 224), 9, 1.0)
    base = mk(0.0)
    sd = fill_state_dict(base.state_dict(), weight_scale=0.03)
    base.load_state_dict(sd)
    model = mk(0.1)
    model.load_state_dict(sd)
    base.to(d).train()
    model.to(d).train()
    xin = [x.to(d), x.to(d)[:, :, :96, :96].contiguous()] if family == "dino" else x.to(d)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(xin)
        ref = base(xin)
    out = out[0] if isinstance(out, (tuple, list)) else out
    ref = ref[0] if isinstance(ref, (tuple, list)) else ref
    assert torch.isfinite(out.float()).all()
    assert not torch.equal(out, ref), "drop_attn = 0.1 left the forward unchanged"
    out.float().square().mean().backward()
    n = 0
    for name, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), name
            n += 1
    assert n > 10
    model.eval()
    base.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        a, b = model(xin), base(xin)
    a = a[0] if isinstance(a, (tuple, list)) else a
    b = b[0] if isinstance(b, (tuple, list)) else b
    assert torch.equal(a, b), "eval mode differs between drop_attn = 0.1 and 0"
