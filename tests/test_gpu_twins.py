"""GPU parity of the Twins-SVT path (the row after SURVEY section 8 F1-F4) through the C ABI: the positional-encoding
generator's depthwise-convolution kernels and the sub-sampling operand gather vs torch / the oracle; the modules and
Twins-SVT-S in fp32 vs the reference's own outputs (golden G10, incl. the stochastic-depth masks the reference drew) and
in bf16 vs the fp32 oracle on a seeded default-init model."""
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


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C", [(2, 14, 14, 64), (3, 56, 56, 64), (2, 7, 7, 512), (2, 8, 12, 320), (1, 1, 5, 8), (2, 28, 28, 128)])
def test_positional_encoding_generator_kernels(dtype, B, H, W, C):
    """y = x + dwconv3x3(x), its input gradient (mirrored taps) and its weight gradient vs torch's conv2d in fp64 on the
    same (rounded) operands; the weight gradient twice: deterministic."""
    from vtx import ops
    d = dev()
    g = torch.Generator().manual_seed(H * 131 + C)
    x = torch.randn((B, H, W, C), generator=g).to(dtype)
    w = (0.3 * torch.randn((C, 1, 3, 3), generator=g)).float()
    dy = torch.randn((B, H, W, C), generator=g).to(dtype)
    y = ops.dwconv3_fwd(x.to(d), w.to(d))
    dx = ops.dwconv3_fwd(dy.to(d), w.to(d), adjoint=True)
    dw = ops.dwconv3_wgrad(x.to(d), dy.to(d))
    dw2 = ops.dwconv3_wgrad(x.to(d), dy.to(d))
    assert torch.equal(dw, dw2), "dwconv3 weight gradient is not deterministic"
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = R.twins_peg(xr, wr)
    dxr, dwr = torch.autograd.grad(yr, [xr, wr], dy.double())
    tag = f"{dtype} B{B} {H}x{W} C{C}"
    check(f"peg fwd {tag}", y, yr, TOL[dtype]["out"])
    check(f"peg dx {tag}", dx, dxr, TOL[dtype]["out"])
    check(f"peg dw {tag}", dw, dwr, 2e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C,r", [(2, 28, 28, 128, 7), (1, 56, 56, 64, 7), (3, 7, 7, 512, 7), (2, 8, 12, 24, 4), (2, 14, 14, 256, 7),
                                       (1, 56, 56, 512, 7)])         # (the last one: beyond the reciprocal-division range)
def test_subsample_gather_is_the_reference_reshape(dtype, B, H, W, C, r):
    """The operand gather of the reduction conv vs the reference's ``input.transpose(1, 2).reshape(B, C, H, W)`` followed by
    the unfold of a stride = kernel convolution (twins.py:69-71): a permutation, so exact; backward = inverse (+ add)."""
    from vtx import ops
    d = dev()
    x = fill((B, H, W, C), 431, 1.0).to(dtype)
    out = ops.twins_subsample_fwd(x.to(d), B, H, W, C, r)
    img = x.transpose(1, 2).reshape(B, C, H, W)                               # as written in the reference
    unfold = lambda t: torch.nn.functional.unfold(t, r, stride=r).transpose(1, 2).reshape(-1, C * r * r)   # (c', py, px) columns
    cols = unfold(img.float()).to(dtype)
    assert torch.equal(out.cpu(), cols), "subsample gather differs from the reference's reshape"
    out2, out_t = ops.twins_subsample_fwd(x.to(d), B, H, W, C, r, transposed=True)
    assert torch.equal(out2, out) and torch.equal(out_t, out.t()), "the transposed copy is not the transpose"
    g = fill(tuple(out.shape), 432, 1.0).to(dtype)
    dx = torch.full((B, H, W, C), float("nan"), dtype=dtype, device=d)
    ops.twins_subsample_bwd(g.to(d), dx, B, H, W, C, r)
    xr = x.double().requires_grad_(True)
    colr = unfold(xr.transpose(1, 2).reshape(B, C, H, W))
    (want,) = torch.autograd.grad(colr, xr, g.double())
    assert torch.equal(dx.cpu().double(), want), "subsample scatter is not the inverse permutation"
    base = fill((B, H, W, C), 433, 1.0).to(dtype)
    acc = base.to(d).clone()
    ops.twins_subsample_bwd(g.to(d), acc, B, H, W, C, r, accumulate=True)
    check(f"subsample bwd accumulate {dtype}", acc, base.double() + want, TOL[dtype]["out"])


def _module_vs_golden(g, tag, mod, x, seed, tol=1e-4, gtol=None):
    gtol = gtol or tol
    mod.load_state_dict(fill_state_dict(mod.state_dict()))
    mod.to(dev()).train()
    xd = x.to(dev()).requires_grad_(True)
    out = mod(xd)
    e = check_summary(out, g.rec(f"{tag}.out"), tol, f"{tag} out")
    report(f"twins {tag} fp32 out vs reference (fp64 golden)", e, tol)
    (out * fill(out.shape, seed, 1.0).to(dev())).sum().backward()
    e = check_summary(xd.grad, g.rec(f"{tag}.dx"), gtol, f"{tag} dx")
    report(f"twins {tag} fp32 dx vs reference", e, gtol)
    for n, p in mod.named_parameters():
        e = check_summary(p.grad, g.rec(f"{tag}.grad.{n}"), gtol, f"{tag} {n}")
        report(f"twins {tag} fp32 grad {n} vs reference", e, gtol)


def test_twins_modules_fp32_vs_reference():
    """The four module classes standalone (the reference's forward contracts) vs the reference's outputs (golden G10)."""
    import models.twins as T
    g = Golden("g10_twins")
    _module_vs_golden(g, "peg", T.PositionalEncodingGenerator(64), fill((2, 14, 14, 64), 81, 1.0), 82)
    _module_vs_golden(g, "lsa", T.MultiHeadedLocalAttention(64, 2, 32, 7), fill((2, 14, 14, 64), 83, 1.0), 84)
    # (the gradients of this case come out of a 6 272-term contraction of the smooth formula fill with heavy cancellation:
    #  the CPU oracle's OWN fp32 run is 4.0e-4 (dx) / 4.7e-4 (reduce_conv.weight) off the fp64 golden -> 3e-3 for fp32 here)
    _module_vs_golden(g, "gsa", T.MultiHeadedAttention(128, 4, reduction=7), fill((2, 28, 28, 128), 85, 1.0), 86, gtol=3e-3)
    _module_vs_golden(g, "patch_embed", T.PatchEmbedding(64, 128, 2), fill((2, 28, 28, 64), 87, 1.0), 88)


def _twins(drop_path=0.0):
    from models.twins import TwinsSVT
    return TwinsSVT(**M.TWINS_SVT_S, drop_path=drop_path)


def test_twins_svt_s_fp32_vs_reference():
    """fp32 parity mode vs the reference's own outputs (golden G10): state_dict inventory, logits, every per-parameter
    gradient norm, sampled gradient values."""
    g = Golden("g10_twins")
    model = _twins()
    assert list(model.state_dict().keys()) == [str(k) for k in g.arr("twins_svt_s.state_keys")]
    assert [str(tuple(v.shape)) for v in model.state_dict().values()] == [str(s) for s in g.arr("twins_svt_s.state_shapes")]
    assert sum(p.numel() for p in model.parameters()) == int(g.arr("twins_svt_s.n_params"))
    model.load_state_dict(fill_state_dict(model.state_dict()))
    model.to(dev()).train()
    x = fill((2, 3, 224, 224), 21, 1.0).to(dev())
    out = model(x)
    e = check_summary(out, g.rec("twins_svt_s.train64.logits"), 1e-4, "twins logits vs reference fp64")
    report("twins-svt-s fp32 logits vs reference (fp64 golden)", e, 1e-4)
    cot = fill(out.shape, name_seed("twins_svt_s.train64.cot"), 1.0).to(dev())
    (out * cot).sum().backward()
    names = [str(n) for n in g.arr("twins_svt_s.train64.grad_names")]
    norms = g.arr("twins_svt_s.train64.grad_norms")
    got = dict(model.named_parameters())
    assert names == list(got.keys())
    worst, worst_n = 0.0, ""
    for n, ref in zip(names, norms):
        gn = got[n].grad.double().norm().item()
        if ref == 0.0:
            # stage 4 attends to ONE sub-sampled key (7 x 7 map, reduction 7): softmax over one key is 1, dS = P (dP - sum P dP)
            # vanishes and the reference's linear_q gradient is exactly 0; here P = exp(s - lse) and the two sums come out of
            # different MFMA orders, so a rounding-sized remainder survives -- bounded against the model's gradient scale
            assert "block4" in n and "linear_q" in n
            assert report(f"twins-svt-s fp32: |grad {n}| where the reference's is exactly 0", gn, 1e-9 * float(np.median(norms)))
            continue
        rel = abs(gn - ref) / ref
        if rel > worst:
            worst, worst_n = rel, n
    assert report(f"twins-svt-s fp32: worst per-param grad-norm deviation ({worst_n})", worst, 5e-3)
    for k in g.keys("twins_svt_s.train64.grad."):                      # sampled gradient values (same cotangent)
        pn = k[len("twins_svt_s.train64.grad."):]
        e = check_summary(got[pn].grad, g.rec(k), 5e-3, k)
        report(f"twins-svt-s fp32: grad {pn}", e, 5e-3)


def test_twins_svt_s_drop_path_with_reference_masks(monkeypatch):
    """Train mode, drop_path 0.3, with the masks the reference itself drew (four per layer, captured in the golden)."""
    import models.twins as T
    g = Golden("g10_twins")
    masks = torch.from_numpy(g.arr("twins_svt_s.dp.masks").astype(np.float32)).to(dev())
    model = _twins(0.3)
    model.load_state_dict(fill_state_dict(model.state_dict()))
    model.to(dev()).train()
    it = iter(range(masks.shape[0]))

    def fake_scale(p, training, batch, device):
        if not training or p == 0:
            return None
        return masks[next(it)] / (1.0 - p)

    monkeypatch.setattr(T, "drop_path_scale", fake_scale)
    x = fill((4, 3, 224, 224), 22, 1.0).to(dev())
    out = model(x)
    assert next(it, None) is None, "the model did not consume the reference's 32 draws"
    e = check_summary(out, g.rec("twins_svt_s.dp.logits"), 1e-4, "twins drop-path logits")
    report("twins-svt-s fp32 drop-path logits vs reference", e, 1e-4)


def test_twins_svt_s_bf16_autocast_vs_oracle():
    from test_gpu_models import _bf16_vs_oracle, _seeded_init
    model = _twins()
    sd = _seeded_init(model, 4)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(14))
    _bf16_vs_oracle(model, lambda P, xx: M.twins_forward(P, xx, M.TWINS_SVT_S), sd, x, "twins-svt-s")


def test_twins_svt_s_bf16_step_with_its_own_mask_draws_is_deterministic():
    """drop_path 0.2 through vtx.nn.drop_path_scope (four draws per layer from one batched draw): two runs from the same seed
    give bitwise-identical logits and gradients; finite."""
    model = _twins(0.2)
    model.to(dev()).train()
    x = torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(15)).to(dev())
    res = []
    for _ in range(2):
        torch.manual_seed(99)
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x)
        out.float().square().mean().backward()
        res.append((out.clone(), [p.grad.clone() for p in model.parameters()]))
    assert torch.isfinite(res[0][0].float()).all()
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("H,C,n_head,ff", [(7, 512, 16, 2048), (14, 256, 8, 1024)])
def test_late_stage_reduction_conv_on_the_split_k_launch(monkeypatch, H, C, n_head, ff):
    """Stages 3 / 4 at the benchmark's batch: the few-row / long-K reduction conv runs as a split-K (weight-gradient) launch on
    the transposed operand copy (functional._TWINS_SPLITK).  Same layer with the switch off = the plain GEMM: outputs and all
    gradients agree to bf16 rounding (the two paths differ in summation order only), and both match the fp32 oracle band."""
    import models.twins as T
    from vtx import functional as VF
    from vtx import ops
    torch.manual_seed(11)
    layer = T.TransformerLayer(C, n_head, 32, ff, 7).to(dev()).train()
    x = torch.randn(128, H, H, C, generator=torch.Generator().manual_seed(12)).to(dev())
    cot = torch.randn(128, H, H, C, generator=torch.Generator().manual_seed(13)).to(dev())
    res = {}
    taken = []
    real = VF._sr_layer_plan            # (the layer runs through ONE C call: the plan carries the split-K decision ...
    monkeypatch.setattr(VF, "_sr_layer_plan", lambda *a: (taken.append(bool(a[-1])), real(*a))[1])
    real_g = ops.twins_subsample_fwd    #  ... or call by call under VTX_LAYER_CALL=0 / VTX_DEFER_REDUCE=0: the gather shows it)
    monkeypatch.setattr(ops, "twins_subsample_fwd", lambda *a, transposed=False: (taken.append(transposed), real_g(*a, transposed=transposed))[1])
    for on in (True, False):
        monkeypatch.setattr(VF, "_TWINS_SPLITK", on)
        layer.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            with VF.weight_scope(layer, xi):
                out = layer(xi)
        (out.float() * cot).sum().backward()
        res[on] = (out.float(), xi.grad.float(), {n: p.grad.clone() for n, p in layer.named_parameters()})
    assert taken == [True, False], f"split-K path not taken where expected: {taken}"
    check("twins split-K reduction conv: layer output vs the GEMM path", res[True][0], res[False][0], 4e-3)
    check("twins split-K reduction conv: dx vs the GEMM path", res[True][1], res[False][1], 1e-2)
    for n in res[True][2]:
        if H == 7 and n == "attn_global.linear_q.weight":
            # one sub-sampled key: the true gradient is exactly 0, both paths hold rounding remainders (see the fp32 model test)
            scale = res[True][2]["attn_global.linear_kv.weight"].norm().item()
            assert res[True][2][n].norm().item() < 1e-3 * scale and res[False][2][n].norm().item() < 1e-3 * scale
            continue
        check(f"twins split-K reduction conv: grad {n} vs the GEMM path", res[True][2][n], res[False][2][n], 1e-2)


@pytest.mark.parametrize("cfg,size", [
    (dict(n_class=10, depths=(1, 1, 2, 1), dims=(32, 64, 128, 256), dim_head=32, n_heads=(1, 2, 4, 8), dim_ffs=(64, 128, 256, 512),
          window_size=4), 128),                        # window 4 on 32 / 16 / 8 / 4 maps: 64 / 16 / 4 / 1 sub-sampled keys
    (dict(n_class=10, depths=(1, 1, 1, 1), dims=(64, 64, 128, 256), dim_head=64, n_heads=(1, 1, 2, 4), dim_ffs=(128, 128, 256, 512),
          window_size=7), 224),                        # head dim 64: the locally-grouped half takes the generic attention kernels
    (dict(n_class=10, depths=(1, 1, 1, 1), dims=(64, 128, 256, 512), dim_head=32, n_heads=(2, 4, 8, 16), dim_ffs=(64, 128, 256, 512),
          window_size=7), (224, 448)),                 # a 56 x 112 map is 8 x 16 = 128 sub-sampled keys: the key-block kernels (round 5)
])
def test_twins_other_geometries_fp32_vs_oracle(cfg, size):
    """Other windows / head dims / widths than Twins-SVT-S through the same modules (fp32 parity mode vs the CPU oracle: logits and
    every parameter gradient), including more than 64 sub-sampled keys."""
    from models.twins import TwinsSVT
    from test_gpu_models import _seeded_init
    model = TwinsSVT(**cfg, drop_path=0.0)
    sd = _seeded_init(model, 6)
    model.to(dev()).train()
    hw = (size, size) if isinstance(size, int) else size
    x = torch.randn(2, 3, *hw, generator=torch.Generator().manual_seed(16))
    out = model(x.to(dev()))
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = M.twins_forward(P, x, cfg)
    check(f"twins {cfg['window_size']}/{cfg['dim_head']} fp32 logits vs oracle", out, ref, 1e-4)
    cot = torch.randn(ref.shape, generator=torch.Generator().manual_seed(17))
    (out * cot.to(dev())).sum().backward()
    names = [n for n, _ in model.named_parameters()]
    rg = torch.autograd.grad((ref * cot).sum(), [P[n] for n in names])
    got = dict(model.named_parameters())
    scale = max(r.norm().item() for r in rg)
    for n, r in zip(names, rg):
        if r.norm().item() < 1e-6 * scale:             # (a single sub-sampled key: linear_q's gradient is exactly 0)
            assert got[n].grad.norm().item() < 1e-5 * scale, n
            continue
        check(f"twins {cfg['window_size']}/{cfg['dim_head']} fp32 grad {n}", got[n].grad, r, 2e-3)
