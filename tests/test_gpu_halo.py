"""GPU parity of the Halo family (reference models/halo_transformer.py; SURVEY section 8 row F4's last sentence) through the C ABI:
the window gather / scatter kernels (exact: permutation / zero padding; the scatter is the gather's adjoint), the cross-attention
kernels with a relative-position term vs the oracle, the attention module in fp32 vs the reference's own outputs (golden G12), the
model in fp32 vs the reference's logits and vs the oracle's gradients, bf16 within the whole-model band."""
import numpy as np
import pytest
import torch

from golden_util import Golden
from gpu_util import TOL, check, dev, report
from oracle import ref_models as M
from oracle import ref_ops as R
from oracle.formula import check_summary, fill, fill_state_dict, name_seed
from test_oracle_halo import CASES

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C,c0,nc,win,halo", [(2, 14, 14, 192, 64, 128, 7, 3), (1, 8, 12, 96, 0, 32, 4, 1), (3, 16, 16, 64, 0, 64, 8, 0),
                                                     (2, 56, 56, 192, 64, 128, 7, 3), (2, 6, 6, 24, 8, 16, 3, 2), (1, 4, 4, 16, 0, 16, 2, 5)])
def test_window_gather_is_unfold_with_zero_padding_and_scatter_is_its_adjoint(dtype, B, H, W, C, c0, nc, win, halo):
    from vtx import ops
    d = dev()
    x = fill((B, H, W, C), 501, 1.0).to(dtype)
    out = ops.window_gather(x.to(d), B, H, W, c0, nc, win, halo)
    side = win + 2 * halo
    # F.unfold of the channel slice, as the reference builds its neighbourhoods (halo_transformer.py:70-76): (B, nc * side^2, nW)
    un = torch.nn.functional.unfold(x[..., c0:c0 + nc].float().permute(0, 3, 1, 2), side, stride=win, padding=halo)
    want = un.view(B, nc, side * side, -1).permute(0, 3, 2, 1).reshape(-1, side * side, nc).to(dtype)
    assert torch.equal(out.cpu(), want), "window gather differs from F.unfold"
    # adjoint: <gather(x), y> = <x, scatter(y)>; exact sums of <= 9 terms in fp32, one rounding on store
    y = fill(tuple(out.shape), 502, 1.0).to(dtype)
    m = torch.full((B, H, W, C), 7.0, dtype=dtype, device=d)
    ops.window_scatter(y.to(d), m, B, H, W, c0, nc, win, halo)
    fold = torch.nn.functional.fold(y.float().view(B, -1, side * side, nc).permute(0, 3, 2, 1).reshape(B, nc * side * side, -1), (H, W), side,
                                    stride=win, padding=halo).permute(0, 2, 3, 1)
    got = m.cpu().float()
    untouched = torch.equal(got[..., :c0], torch.full_like(got[..., :c0], 7.0)) and torch.equal(got[..., c0 + nc:], torch.full_like(got[..., c0 + nc:], 7.0))
    assert untouched, "scatter touched channels outside its range"
    check(f"window scatter {dtype} win{win} halo{halo}", got[..., c0:c0 + nc], fold.double(), 1e-6 if dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Lq,Lk,nH,D", [(8, 49, 169, 2, 32), (3, 64, 196, 2, 64), (5, 16, 36, 3, 32), (2, 100, 70, 1, 64)])
def test_cross_attention_with_a_score_bias_vs_oracle(dtype, B, Lq, Lk, nH, D):
    from vtx import ops
    d = dev()
    gen = torch.Generator().manual_seed(503)
    hd = nH * D
    q = torch.randn(B * Lq, hd, generator=gen).to(dtype)
    kv = torch.randn(B * Lk, 2 * hd, generator=gen).to(dtype)
    bias = torch.randn(nH, Lq, Lk, generator=gen)
    do = torch.randn(B * Lq, hd, generator=gen).to(dtype)
    o, lse = ops.xattn_fwd(q.to(d), kv.to(d), B, Lq, Lk, nH, bias.to(d))
    dq, dkv, dbias = ops.xattn_bwd(q.to(d), kv.to(d), o, do.to(d), lse, B, Lq, Lk, nH, bias.to(d))
    dq2, dkv2, dbias2 = ops.xattn_bwd(q.to(d), kv.to(d), o, do.to(d), lse, B, Lq, Lk, nH, bias.to(d))
    assert torch.equal(dq, dq2) and torch.equal(dkv, dkv2) and torch.equal(dbias, dbias2), "cross-attention backward is not deterministic"
    qr, kvr, br = q.double().requires_grad_(True), kv.double().requires_grad_(True), bias.double().requires_grad_(True)
    Q = qr.view(B, Lq, nH, D)
    K, V = kvr.view(B, Lk, 2, nH, D)[:, :, 0], kvr.view(B, Lk, 2, nH, D)[:, :, 1]
    S = torch.einsum("bqhd,bkhd->bhqk", Q, K) / D ** 0.5 + br
    oref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(S, -1), V).reshape(B * Lq, hd)
    gq, gkv, gb = torch.autograd.grad(oref, [qr, kvr, br], do.double())
    t = TOL[dtype]
    tag = f"{dtype} B{B} Lq{Lq} Lk{Lk} h{nH} d{D}"
    check(f"xattn fwd {tag}", o, oref, t["out"] * 1.5)
    check(f"xattn dq {tag}", dq, gq, 2e-5 if dtype == torch.float32 else 1e-2)
    check(f"xattn dkv {tag}", dkv, gkv, 2e-5 if dtype == torch.float32 else 1e-2)
    check(f"xattn dbias {tag}", dbias, gb, 2e-5 if dtype == torch.float32 else 1e-2)
    # without a bias the same kernels are the plain cross attention
    o0, l0 = ops.xattn_fwd(q.to(d), kv.to(d), B, Lq, Lk, nH)
    o00 = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(torch.einsum("bqhd,bkhd->bhqk", Q, K) / D ** 0.5, -1), V).reshape(B * Lq, hd)
    check(f"xattn fwd no bias {tag}", o0, o00, t["out"] * 1.5)


@pytest.mark.parametrize("tag", sorted(CASES))
def test_halo_attention_module_fp32_vs_reference(tag):
    import models.halo_transformer as HT
    g = Golden("g12_halo")
    dim, nh, dh, w, a, hw = CASES[tag]
    mod = HT.MultiHeadedHaloAttention(dim, nh, dh, w, a)
    assert np.array_equal(mod.pos.numpy(), g.arr(f"halo_{tag}.pos").astype(np.int64)) and mod.rel_pos.num_embeddings == int(g.arr(f"halo_{tag}.ntab"))
    mod.load_state_dict(fill_state_dict(mod.state_dict()))
    mod.to(dev()).train()
    x = fill((2, hw[0], hw[1], dim), 91, 1.0).to(dev()).requires_grad_(True)
    out = mod(x)
    e = check_summary(out, g.rec(f"halo_{tag}.out"), 2e-4, f"halo {tag} out")
    report(f"halo attention {tag} fp32 out vs reference (fp64 golden)", e, 2e-4)
    (out * fill(out.shape, name_seed(f"halo_{tag}.cot"), 1.0).to(dev())).sum().backward()
    e = check_summary(x.grad, g.rec(f"halo_{tag}.dx"), 1e-3, f"halo {tag} dx")
    report(f"halo attention {tag} fp32 dx vs reference", e, 1e-3)
    for n, p in mod.named_parameters():
        e = check_summary(p.grad, g.rec(f"halo_{tag}.d.{n}"), 1e-3, f"halo {tag} {n}")
        report(f"halo attention {tag} fp32 grad {n} vs reference", e, 1e-3)


def test_halo_transformer_fp32_vs_reference_and_oracle_and_bf16():
    """state_dict inventory and parameter order = the reference's; fp32 logits vs the reference's (golden G12); every parameter
    gradient vs the CPU oracle (the reference model's own backward raises: in-place residual adds); bf16 within the band."""
    from models.halo_transformer import HaloTransformer
    g = Golden("g12_halo")
    model = HaloTransformer(**M.HALO_TINY)
    assert [k for k in model.state_dict()] == [str(k) for k in g.arr("halo_tiny.state_keys")]
    assert [str(tuple(v.shape)) for v in model.state_dict().values()] == [str(s) for s in g.arr("halo_tiny.state_shapes")]
    assert [n for n, _ in model.named_parameters()] == [str(k) for k in g.arr("halo_tiny.param_names")]
    assert sum(p.numel() for p in model.parameters()) == int(g.arr("halo_tiny.n_params"))
    sd = fill_state_dict(model.state_dict())
    model.load_state_dict(sd)
    model.to(dev()).eval()
    x = fill((2, 3, 224, 224), 21, 1.0)
    with torch.no_grad():
        out = model(x.to(dev()))
    e = check_summary(out, g.rec("halo_tiny.eval.logits"), 1e-4, "halo tiny logits")
    report("halo tiny fp32 eval logits vs reference", e, 1e-4)
    model.train()
    out = model(x.to(dev()))
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items() if torch.is_floating_point(v)}
    ref = M.halo_forward(P, x, M.HALO_TINY)
    check("halo tiny fp32 train logits vs oracle", out, ref, 1e-4)
    cot = torch.randn(ref.shape, generator=torch.Generator().manual_seed(31))
    (out * cot.to(dev())).sum().backward()
    names = [n for n, _ in model.named_parameters()]
    rg = torch.autograd.grad((ref * cot).sum(), [P[n] for n in names])
    got = dict(model.named_parameters())
    for n, r in zip(names, rg):
        check(f"halo tiny fp32 grad {n}", got[n].grad, r, 2e-3)
    # bf16 autocast on a reference-style random init (std 0.02; the formula fill's larger weights amplify the bf16 rounding of four
    # stages of activations: 4.8e-2 there) vs the fp32 oracle: the whole-model band of the other families
    from test_gpu_models import _seeded_init
    mb = HaloTransformer(**M.HALO_TINY)
    sdb = _seeded_init(mb, 41)
    mb.to(dev()).train()
    xb = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(42))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ob = mb(xb.to(dev()))
    check("halo tiny bf16 logits (seeded init) vs oracle", ob.float(), M.halo_forward(sdb, xb, M.HALO_TINY), 2e-2)
    # DropPath (the same rate in every layer) runs and is reproducible under the seed
    from models.halo_transformer import HaloTransformer as HTm
    dp = HTm(**M.HALO_TINY, drop_path=0.2)
    dp.load_state_dict(sd)
    dp.to(dev()).train()
    torch.manual_seed(5)
    a = dp(x.to(dev()))
    torch.manual_seed(5)
    b = dp(x.to(dev()))
    assert torch.equal(a, b) and torch.isfinite(a).all()
