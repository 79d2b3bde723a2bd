"""Pin the Twins-SVT part of the CPU oracle (oracle/ref_ops.py twins_*; oracle/ref_models.py twins_forward) against outputs
of the reference's models/twins.py (golden G10, tools/gen_goldens.py twins) -- the row after SURVEY section 8 F1-F4."""
import ast

import torch

from golden_util import Golden
from oracle import ref_models as M
from oracle import ref_ops as R
from oracle.formula import check_summary, fill, fill_state_dict, name_seed
from test_oracle_models import check_model_grads
from test_oracle_pvt import module_params

torch.set_num_threads(8)


def twins_params(g, dtype=torch.float32):
    keys = [str(k) for k in g.arr("twins_svt_s.state_keys")]
    shapes = [ast.literal_eval(str(s)) for s in g.arr("twins_svt_s.state_shapes")]
    sd = fill_state_dict({k: torch.zeros(s) for k, s in zip(keys, shapes)})
    return {k: v.to(dtype).requires_grad_(True) for k, v in sd.items()}


def _module_case(g, tag, P, x, fn, seed):
    out = fn(x, P)
    check_summary(out, g.rec(f"{tag}.out"), 5e-7, f"{tag} out")
    grads = torch.autograd.grad((out * fill(out.shape, seed, 1.0).double()).sum(), [x] + list(P.values()))
    check_summary(grads[0], g.rec(f"{tag}.dx"), 5e-7, f"{tag} dx")
    for (n, _), gr in zip(P.items(), grads[1:]):
        check_summary(gr, g.rec(f"{tag}.grad.{n}"), 5e-7, f"{tag} {n}")


def test_twins_svt_s_inventory():
    g = Golden("g10_twins")
    P = twins_params(g)
    assert sum(v.numel() for v in P.values()) == int(g.arr("twins_svt_s.n_params"))
    keys = list(P.keys())
    # block layout of twins.py:327-345: [PatchEmbedding, layer, PEG, layer ...]; no position / mask buffers
    assert "block1.2.proj.weight" in keys and "block3.2.proj.weight" in keys and "block3.6.ff_global.3.bias" in keys
    assert not any("pos" in k or "mask" in k for k in keys)


def test_positional_encoding_generator_fp64():
    g = Golden("g10_twins")
    P = module_params({"proj.weight": (64, 1, 3, 3)})
    x = fill((2, 14, 14, 64), 81, 1.0).double().requires_grad_(True)
    _module_case(g, "peg", P, x, lambda t, p: R.twins_peg(t, p["proj.weight"]), 82)


def test_locally_grouped_attention_fp64():
    g = Golden("g10_twins")
    P = module_params({"weight.weight": (192, 64), "weight.bias": (192,), "linear.weight": (64, 64), "linear.bias": (64,)})
    x = fill((2, 14, 14, 64), 83, 1.0).double().requires_grad_(True)
    _module_case(g, "lsa", P, x, lambda t, p: R.twins_local_attention(t, p, 2, 32, 7), 84)


def test_global_subsampled_attention_fp64():
    g = Golden("g10_twins")
    P = module_params({"linear_q.weight": (128, 128), "linear_kv.weight": (256, 128), "linear.weight": (128, 128),
                       "linear.bias": (128,), "reduce_conv.weight": (128, 128, 7, 7), "reduce_conv.bias": (128,)})
    x = fill((2, 28, 28, 128), 85, 1.0).double().requires_grad_(True)
    _module_case(g, "gsa", P, x, lambda t, p: R.twins_global_attention(t, p, 4, 7), 86)


def test_patch_embedding_of_a_later_stage_fp64():
    g = Golden("g10_twins")
    P = module_params({"linear.weight": (128, 256), "linear.bias": (128,), "norm.weight": (128,), "norm.bias": (128,)})
    x = fill((2, 28, 28, 64), 87, 1.0).double().requires_grad_(True)
    _module_case(g, "patch_embed", P, x,
                 lambda t, p: R.twins_patch_embedding(t, p["linear.weight"], p["linear.bias"], p["norm.weight"], p["norm.bias"], 2),
                 88)


def test_twins_svt_s_full_model_fp64():
    g = Golden("g10_twins")
    P = twins_params(g, torch.float64)
    x = fill((2, 3, 224, 224), 21, 1.0).double()
    out = M.twins_forward(P, x, M.TWINS_SVT_S)
    check_summary(out, g.rec("twins_svt_s.train64.logits"), 5e-7, "twins fp64 logits")
    cot = fill(out.shape, name_seed("twins_svt_s.train64.cot"), 1.0, dtype=torch.float64)
    grads = torch.autograd.grad((out * cot).sum(), list(P.values()))
    check_model_grads(g, "twins_svt_s.train64", P, grads, 2e-6)


def test_twins_svt_s_full_model_fp32_and_drop_path_masks():
    g = Golden("g10_twins")
    P = twins_params(g)
    x = fill((2, 3, 224, 224), 21, 1.0)
    with torch.no_grad():
        out = M.twins_forward(P, x, M.TWINS_SVT_S)
    check_summary(out, g.rec("twins_svt_s.eval.logits"), 2e-5, "twins eval logits")
    check_summary(out, g.rec("twins_svt_s.train.logits"), 2e-5, "twins train logits")
    # the reference's own stochastic-depth draws (4 per layer, layer 0 has rate 0): same logits with the captured masks
    masks = torch.from_numpy(g.arr("twins_svt_s.dp.masks")).float()
    n_layers = sum(M.TWINS_SVT_S["depths"])
    assert masks.shape == (4 * (n_layers - 1), 4)
    per_layer = [(None,) * 4] + [tuple(masks[4 * i + k] for k in range(4)) for i in range(n_layers - 1)]
    xb = fill((4, 3, 224, 224), 22, 1.0)
    with torch.no_grad():
        outd = M.twins_forward(P, xb, M.TWINS_SVT_S, drop_masks=per_layer, drop_path=0.3)
    check_summary(outd, g.rec("twins_svt_s.dp.logits"), 2e-5, "twins drop-path logits")
