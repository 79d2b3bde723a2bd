"""Pin the CPU oracle against the reference's own outputs (tests/golden/).

Integer / boolean tables: bit-exact.  Floating point: the goldens were produced by the
reference in fp64 (modules) or fp32 (full models); the oracle runs in fp64 for modules
(tolerance 5e-7 relative L2: fixtures are stored as fp32) and fp32 for full models (tolerance 2e-5: measured
fp32-vs-fp64 noise floor of the reference itself is 6e-7..1e-6).
"""
import numpy as np
import pytest
import torch

from golden_util import Golden
from oracle import ref_models as M
from oracle import ref_ops as R
from oracle import tables
from oracle.formula import check_summary, fill, fill_state_dict, name_seed

TOL64 = 5e-7  # goldens are stored as fp32 (6e-8 storage rounding)


# ------------------------------------------------------------------ G1 / G2
CASES = [((56, 56), 7), ((28, 28), 7), ((14, 14), 7), ((7, 7), 7), ((8, 12), 4),
         ((12, 8), 4), ((6, 6), 3), ((10, 15), 5), ((16, 16), 8)]


@pytest.mark.parametrize("size,w", CASES)
@pytest.mark.parametrize("shift", [False, True])
def test_tables_bit_exact(size, w, shift):
    g = Golden("g1_g2_tables")
    key = f"{size[0]}x{size[1]}_w{w}_s{int(shift)}"
    pos, mask = tables.make_pos_mask(size, w, shift)
    assert pos.dtype == np.int64 and pos.shape == (w * w, w * w)
    assert np.array_equal(pos, g.arr(f"pos_{key}").astype(np.int64))
    if shift:
        shape = tuple(g.arr(f"maskshape_{key}"))
        ref = np.unpackbits(g.arr(f"mask_{key}"))[: int(np.prod(shape))].astype(bool).reshape(shape)
        assert mask.dtype == np.bool_ and mask.shape == shape
        assert np.array_equal(mask, ref)
    else:
        assert mask is None


def test_stage4_shifted_pos_differs():
    """the single-window shifted layer has a wrap-around pos table."""
    p_common, _ = tables.make_pos_mask((14, 14), 7, True)
    p_wrap, m = tables.make_pos_mask((7, 7), 7, True)
    assert not np.array_equal(p_common, p_wrap)
    assert not m.any()
    assert list(p_wrap[0, :8]) == [84, 85, 86, 87, 81, 82, 83, 97]


# ------------------------------------------------------------------ G3
def _params(names_shapes, prefix=""):
    sd = {k: torch.zeros(s) for k, s in names_shapes.items()}
    sd = fill_state_dict(sd)
    return {k: v.double().requires_grad_(True) for k, v in sd.items()}


def _check_module(g, name, out, x, params, rename=None):
    cot = fill(out.shape, name_seed(name + ".cot"), 1.0).double()
    ins = ([x] if x is not None and x.requires_grad else []) + list(params.values())
    grads = torch.autograd.grad((out * cot).sum(), ins)
    check_summary(out, g.rec(f"{name}.out"), TOL64, f"{name}.out")
    i = 0
    if x is not None and x.requires_grad:
        check_summary(grads[0], g.rec(f"{name}.dx"), TOL64, f"{name}.dx")
        i = 1
    for (k, _), gr in zip(params.items(), grads[i:]):
        check_summary(gr, g.rec(f"{name}.d.{k}"), TOL64, f"{name}.d.{k}")


@pytest.mark.parametrize("size,shift,tag", [((14, 14), True, "s1"), ((14, 14), False, "s0"),
                                            ((7, 7), True, "wrap")])
def test_window_attention(size, shift, tag):
    g = Golden("g3_modules")
    P = _params({"weight.weight": (288, 96), "weight.bias": (288,), "linear.weight": (96, 96),
                 "linear.bias": (96,), "rel_pos.weight": (169, 3)})
    x = fill((2, size[0], size[1], 96), 11, 1.0, dtype=torch.float64).requires_grad_(True)
    out = R.window_attention(x, P["weight.weight"], P["weight.bias"], P["linear.weight"],
                             P["linear.bias"], P["rel_pos.weight"], 3, 32, 7, shift)
    _check_module(g, f"local_attn_{tag}", out, x, P)


@pytest.mark.parametrize("L", [197, 37])
def test_global_attention(L):
    g = Golden("g3_modules")
    P = _params({"qkv.weight": (1152, 384), "qkv.bias": (1152,), "linear.weight": (384, 384),
                 "linear.bias": (384,)})
    x = fill((2, L, 384), 12, 1.0, dtype=torch.float64).requires_grad_(True)
    out = R.global_attention(x, P["qkv.weight"], P["qkv.bias"], P["linear.weight"], P["linear.bias"], 6)
    _check_module(g, f"global_attn_L{L}", out, x, P)


def test_ffn():
    g = Golden("g3_modules")
    P = _params({"0.weight": (384, 96), "0.bias": (384,), "3.weight": (96, 384), "3.bias": (96,)})
    x = fill((2, 49, 96), 13, 1.0, dtype=torch.float64).requires_grad_(True)
    out = R.feed_forward(x, P["0.weight"], P["0.bias"], P["3.weight"], P["3.bias"])
    _check_module(g, "ffn", out, x, P)


def test_patch_embeddings_and_merge():
    g = Golden("g3_modules")
    x = fill((2, 3, 224, 224), 14, 1.0, dtype=torch.float64)
    P = _params({"linear.weight": (384, 3, 16, 16), "linear.bias": (384,)})
    out = R.vit_patch_embedding(x, P["linear.weight"], P["linear.bias"], 16)
    _check_module(g, "vit_patch", out, None, P)
    P = _params({"linear.weight": (96, 48), "linear.bias": (96,), "norm.weight": (96,), "norm.bias": (96,)})
    out = R.swin_patch_embedding(x, P["linear.weight"], P["linear.bias"], P["norm.weight"], P["norm.bias"])
    _check_module(g, "swin_patch", out, None, P)
    P = _params({"norm.weight": (384,), "norm.bias": (384,), "linear.weight": (192, 384)})
    xm = fill((2, 14, 14, 96), 15, 1.0, dtype=torch.float64).requires_grad_(True)
    out = R.patch_merge(xm, P["norm.weight"], P["norm.bias"], P["linear.weight"])
    _check_module(g, "patch_merge", out, xm, P)


@pytest.mark.parametrize("eps,tag", [(1e-6, "e6"), (1e-5, "e5")])
def test_layer_norm(eps, tag):
    g = Golden("g3_modules")
    P = {"weight": fill((96,), 3, 0.1, 1.0).double().requires_grad_(True),
         "bias": fill((96,), 4, 0.02).double().requires_grad_(True)}
    x = fill((2, 49, 96), 16, 2.0, 0.3, dtype=torch.float64).requires_grad_(True)
    out = R.layer_norm(x, P["weight"], P["bias"], eps)
    _check_module(g, f"ln_{tag}", out, x, P)


def test_transformer_layers():
    g = Golden("g3_modules")
    P = _params({"norm_attn.weight": (96,), "norm_attn.bias": (96,), "attn.weight.weight": (288, 96),
                 "attn.weight.bias": (288,), "attn.linear.weight": (96, 96), "attn.linear.bias": (96,),
                 "attn.rel_pos.weight": (169, 3), "norm_ff.weight": (96,), "norm_ff.bias": (96,),
                 "ff.0.weight": (384, 96), "ff.0.bias": (384,), "ff.3.weight": (96, 384), "ff.3.bias": (96,)})
    x = fill((2, 14, 14, 96), 17, 1.0, dtype=torch.float64).requires_grad_(True)
    h = R.layer_norm(x, P["norm_attn.weight"], P["norm_attn.bias"], 1e-6)
    a = R.window_attention(h, P["attn.weight.weight"], P["attn.weight.bias"], P["attn.linear.weight"],
                           P["attn.linear.bias"], P["attn.rel_pos.weight"], 3, 32, 7, True)
    y = x + a
    h = R.layer_norm(y, P["norm_ff.weight"], P["norm_ff.bias"], 1e-6)
    out = y + R.feed_forward(h, P["ff.0.weight"], P["ff.0.bias"], P["ff.3.weight"], P["ff.3.bias"])
    _check_module(g, "swin_layer", out, x, P)

    P = _params({"norm_attn.weight": (384,), "norm_attn.bias": (384,), "attn.qkv.weight": (1152, 384),
                 "attn.qkv.bias": (1152,), "attn.linear.weight": (384, 384), "attn.linear.bias": (384,),
                 "norm_ff.weight": (384,), "norm_ff.bias": (384,), "ff.0.weight": (1536, 384),
                 "ff.0.bias": (1536,), "ff.3.weight": (384, 1536), "ff.3.bias": (384,)})
    x = fill((2, 197, 384), 18, 1.0, dtype=torch.float64).requires_grad_(True)
    h = R.layer_norm(x, P["norm_attn.weight"], P["norm_attn.bias"], 1e-6)
    y = x + R.global_attention(h, P["attn.qkv.weight"], P["attn.qkv.bias"], P["attn.linear.weight"],
                               P["attn.linear.bias"], 6)
    h = R.layer_norm(y, P["norm_ff.weight"], P["norm_ff.bias"], 1e-6)
    out = y + R.feed_forward(h, P["ff.0.weight"], P["ff.0.bias"], P["ff.3.weight"], P["ff.3.bias"])
    _check_module(g, "vit_layer", out, x, P)


# ------------------------------------------------------------------ G11: attention-probability dropout with the reference's keep mask
import attn_dropout_cases as ADC


@pytest.mark.parametrize("name", sorted(ADC.CASES))
def test_attention_dropout_with_the_recorded_keep_mask(name):
    """F.dropout(attn, p, training) of vit.py:39 / swin_transformer.py:144 / pvt.py:60 / twins.py:88,147: the oracle with the keep
    mask the reference run drew reproduces that run's output and every gradient."""
    g = Golden("g11_attn_dropout")
    keep = ADC.keep_mask(g, name)
    assert abs(float(keep.double().mean()) - (1 - ADC.P_DROP)) < 0.02
    P = {k: v.requires_grad_(True) for k, v in ADC.params(name).items()}
    x = ADC.case_input(name).requires_grad_(True)
    out = ADC.CASES[name][3](x, P, keep)
    _check_module(g, name, out, x, P)
    # and the mask matters: without it the output is a different one
    plain = ADC.CASES[name][3](x, P, None)
    assert (plain - out).norm() / out.norm() > 1e-4
