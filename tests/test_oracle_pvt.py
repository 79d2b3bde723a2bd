"""Pin the PVT part of the CPU oracle (oracle/ref_ops.py sr_attention / pvt_*; oracle/ref_models.py pvt_forward)
against outputs of the reference's models/pvt.py (golden G7, tools/gen_goldens.py pvt) -- SURVEY section 8 row F1."""
import ast

import torch

from golden_util import Golden
from oracle import ref_models as M
from oracle import ref_ops as R
from oracle.formula import check_summary, fill, fill_state_dict, name_seed
from test_oracle_models import check_model_grads

torch.set_num_threads(8)


def pvt_params(g, dtype=torch.float32):
    keys = [str(k) for k in g.arr("pvt_small.state_keys")]
    shapes = [ast.literal_eval(str(s)) for s in g.arr("pvt_small.state_shapes")]
    sd = fill_state_dict({k: torch.zeros(s) for k, s in zip(keys, shapes)})
    return {k: v.to(dtype).requires_grad_(True) for k, v in sd.items()}


def module_params(shapes, dtype=torch.float64):
    sd = fill_state_dict({k: torch.zeros(s) for k, s in shapes.items()})
    return {k: v.to(dtype).requires_grad_(True) for k, v in sd.items()}


def test_pvt_small_inventory():
    g = Golden("g7_pvt")
    assert int(g.arr("pvt_small.n_params")) == 24_473_384             # SURVEY section 8 F1 [probe]
    P = pvt_params(g)
    assert sum(v.numel() for v in P.values()) == 24_473_384


def test_sr_attention_module_fp64():
    g = Golden("g7_pvt")
    P = module_params({"linear_q.weight": (128, 128), "linear_kv.weight": (256, 128), "linear.weight": (128, 128),
                       "linear.bias": (128,), "reduce_conv.weight": (128, 128, 4, 4), "reduce_conv.bias": (128,),
                       "reduce_norm.weight": (128,), "reduce_norm.bias": (128,)})
    x = fill((2, 784, 128), 71, 1.0).double().requires_grad_(True)
    out = R.pvt_attention(x, 28, 28, P, 2, 4)
    check_summary(out, g.rec("sr_attn.out"), 5e-7, "sr_attn out")
    grads = torch.autograd.grad((out * fill(out.shape, 72, 1.0).double()).sum(), [x] + list(P.values()))
    check_summary(grads[0], g.rec("sr_attn.dx"), 5e-7, "sr_attn dx")
    for (n, _), gr in zip(P.items(), grads[1:]):
        check_summary(gr, g.rec(f"sr_attn.grad.{n}"), 5e-7, n)


def test_attention_without_reduction_fp64():
    g = Golden("g7_pvt")
    P = module_params({"linear_q.weight": (512, 512), "linear_kv.weight": (1024, 512), "linear.weight": (512, 512),
                       "linear.bias": (512,)})
    x = fill((2, 50, 512), 73, 1.0).double().requires_grad_(True)
    out = R.pvt_attention(x, 7, 7, P, 8, 1)
    check_summary(out, g.rec("attn_r1.out"), 5e-7, "attn_r1 out")
    (dx,) = torch.autograd.grad((out * fill(out.shape, 74, 1.0).double()).sum(), [x])
    check_summary(dx, g.rec("attn_r1.dx"), 5e-7, "attn_r1 dx")


def test_patch_embedding_with_cls_fp64():
    g = Golden("g7_pvt")
    P = module_params({"pos": (50, 512), "cls_token": (512,), "conv.weight": (512, 320, 2, 2), "conv.bias": (512,),
                       "norm.weight": (512,), "norm.bias": (512,)})
    x = fill((2, 320, 14, 14), 75, 1.0).double().requires_grad_(True)
    out, hw = R.pvt_patch_embedding(x, P["conv.weight"], P["conv.bias"], P["norm.weight"], P["norm.bias"], P["pos"],
                                    P["cls_token"], 2)
    assert hw == (7, 7)
    check_summary(out, g.rec("patch_embed.out"), 5e-7, "patch_embed out")
    grads = torch.autograd.grad((out * fill(out.shape, 76, 1.0).double()).sum(), [x] + list(P.values()))
    check_summary(grads[0], g.rec("patch_embed.dx"), 5e-7, "patch_embed dx")
    for (n, _), gr in zip(P.items(), grads[1:]):
        check_summary(gr, g.rec(f"patch_embed.grad.{n}"), 5e-7, n)


def test_pvt_small_full_model_fp64():
    g = Golden("g7_pvt")
    P = pvt_params(g, torch.float64)
    x = fill((2, 3, 224, 224), 21, 1.0).double()
    out = M.pvt_forward(P, x, M.PVT_SMALL)
    check_summary(out, g.rec("pvt_small.train64.logits"), 5e-7, "pvt fp64 logits")
    cot = fill(out.shape, name_seed("pvt_small.train64.cot"), 1.0, dtype=torch.float64)
    grads = torch.autograd.grad((out * cot).sum(), list(P.values()))
    check_model_grads(g, "pvt_small.train64", P, grads, 2e-6)


def test_pvt_small_full_model_fp32():
    g = Golden("g7_pvt")
    P = pvt_params(g)
    x = fill((2, 3, 224, 224), 21, 1.0)
    out = M.pvt_forward(P, x, M.PVT_SMALL)
    check_summary(out, g.rec("pvt_small.eval.logits"), 2e-5, "pvt eval logits")
    check_summary(out, g.rec("pvt_small.train.logits"), 2e-5, "pvt train logits")
    cot = fill(out.shape, name_seed("pvt_small.train.cot"), 1.0)
    grads = torch.autograd.grad((out * cot).sum(), list(P.values()))
    check_model_grads(g, "pvt_small.train", P, grads, 5e-3)
