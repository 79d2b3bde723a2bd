"""Pin the whole-model CPU oracle (oracle/ref_models.py) against reference outputs G4/G5/G6.

fp32 oracle vs fp32 reference goldens: tolerance 2e-5 relative L2 on logits / sampled grads and
1e-4 on per-parameter grad norms (the reference's own fp32-vs-fp64 noise floor, also stored in the
golden as ``*.train64``, is 6e-7..1e-6 on logits and up to ~1e-5 on small grads).
"""
import numpy as np
import pytest
import torch

from golden_util import Golden
from oracle import ref_models as M
from oracle import ref_ops as R
from oracle.formula import check_summary, fill, fill_state_dict, name_seed

torch.set_num_threads(8)


def swin_state_dict(cfg):
    """Names/shapes of the reference SwinTransformer state_dict (floating tensors only)."""
    sd = {}
    d, h, ff = cfg["dims"], cfg["n_heads"], cfg["dim_ffs"]
    sd["patch_embedding.linear.weight"] = (d[0], 48)
    sd["patch_embedding.linear.bias"] = (d[0],)
    sd["patch_embedding.norm.weight"] = (d[0],)
    sd["patch_embedding.norm.bias"] = (d[0],)
    for s in range(4):
        j0 = 0
        if s > 0:
            sd[f"block{s+1}.0.norm.weight"] = (4 * d[s - 1],)
            sd[f"block{s+1}.0.norm.bias"] = (4 * d[s - 1],)
            sd[f"block{s+1}.0.linear.weight"] = (d[s], 4 * d[s - 1])
            j0 = 1
        for i in range(cfg["depths"][s]):
            p = f"block{s+1}.{j0+i}"
            hd = h[s] * cfg["dim_head"]
            for n, shp in [("norm_attn.weight", (d[s],)), ("norm_attn.bias", (d[s],)),
                           ("attn.weight.weight", (3 * hd, d[s])), ("attn.weight.bias", (3 * hd,)),
                           ("attn.linear.weight", (d[s], hd)), ("attn.linear.bias", (d[s],)),
                           ("attn.rel_pos.weight", ((2 * cfg["window_size"] - 1) ** 2, h[s])),
                           ("norm_ff.weight", (d[s],)), ("norm_ff.bias", (d[s],)),
                           ("ff.0.weight", (ff[s], d[s])), ("ff.0.bias", (ff[s],)),
                           ("ff.3.weight", (d[s], ff[s])), ("ff.3.bias", (d[s],))]:
                sd[f"{p}.{n}"] = shp
    sd["final_linear.0.weight"] = (d[3],)
    sd["final_linear.0.bias"] = (d[3],)
    sd["classifier.2.weight"] = (cfg["n_class"], d[3])
    sd["classifier.2.bias"] = (cfg["n_class"],)
    return sd


def vit_state_dict(cfg, n_class=1000):
    dim, ff = cfg["dim"], cfg["dim_ff"]
    n_patch = (cfg["image_size"] // cfg["window_size"]) ** 2
    sd = {"cls_token": (1, 1, dim), "pos_embed": (1, n_patch + 1, dim),
          "patch_embedding.linear.weight": (dim, 3, cfg["window_size"], cfg["window_size"]),
          "patch_embedding.linear.bias": (dim,)}
    for i in range(cfg["depth"]):
        p = f"layers.{i}"
        for n, shp in [("norm_attn.weight", (dim,)), ("norm_attn.bias", (dim,)),
                       ("attn.qkv.weight", (3 * dim, dim)), ("attn.qkv.bias", (3 * dim,)),
                       ("attn.linear.weight", (dim, dim)), ("attn.linear.bias", (dim,)),
                       ("norm_ff.weight", (dim,)), ("norm_ff.bias", (dim,)),
                       ("ff.0.weight", (ff, dim)), ("ff.0.bias", (ff,)),
                       ("ff.3.weight", (dim, ff)), ("ff.3.bias", (dim,))]:
            sd[f"{p}.{n}"] = shp
    sd["norm.weight"] = (dim,)
    sd["norm.bias"] = (dim,)
    if n_class:
        sd["head.weight"] = (n_class, dim)
        sd["head.bias"] = (n_class,)
    return sd


def make_params(shapes, dtype=torch.float32):
    sd = fill_state_dict({k: torch.zeros(s) for k, s in shapes.items()})
    return {k: v.to(dtype).requires_grad_(True) for k, v in sd.items()}


def check_model_grads(g, name, params, grads, norm_tol):
    names = [str(n) for n in g.arr(f"{name}.grad_names")]
    norms = g.arr(f"{name}.grad_norms")
    assert names == list(params.keys()), "parameter name/order differs from the reference's named_parameters()"
    worst = 0.0
    for n, ref, gr in zip(names, norms, grads):
        got = gr.double().norm().item()
        rel = abs(got - ref) / max(ref, 1e-12)
        worst = max(worst, rel)
        assert rel < norm_tol, f"{name} grad norm {n}: {got} vs {ref} (rel {rel:.2e})"
    for k in g.keys(f"{name}.grad."):
        pn = k[len(f"{name}.grad."):]
        check_summary(grads[names.index(pn)], g.rec(k), norm_tol, k)
    return worst


def test_swin_s_full_model_fp64():
    """Tight pin: fp64 oracle vs the reference run in fp64 (fixtures stored as fp32 -> 5e-7)."""
    g = Golden("g4_models")
    P = make_params(swin_state_dict(M.SWIN_S), torch.float64)
    x = fill((2, 3, 224, 224), 21, 1.0).double()
    out = M.swin_forward(P, x, M.SWIN_S)
    check_summary(out, g.rec("swin_s.train64.logits"), 5e-7, "swin fp64 logits")
    cot = fill(out.shape, name_seed("swin_s.train64.cot"), 1.0, dtype=torch.float64)
    grads = torch.autograd.grad((out * cot).sum(), list(P.values()))
    check_model_grads(g, "swin_s.train64", P, grads, 2e-6)


def test_swin_s_full_model_fp32():
    """fp32 oracle vs fp32 reference: both carry fp32 round-off through 24 layers, so the deepest
    gradients (patch embedding) agree only to ~1e-3; logits to 2e-5."""
    g = Golden("g4_models")
    P = make_params(swin_state_dict(M.SWIN_S))
    x = fill((2, 3, 224, 224), 21, 1.0)
    out = M.swin_forward(P, x, M.SWIN_S)
    check_summary(out, g.rec("swin_s.eval.logits"), 2e-5, "swin eval logits")
    check_summary(out, g.rec("swin_s.train.logits"), 2e-5, "swin train logits")
    check_summary(out, g.rec("swin_s.train64.logits"), 2e-5, "swin fp64 logits")
    cot = fill(out.shape, name_seed("swin_s.train.cot"), 1.0)
    grads = torch.autograd.grad((out * cot).sum(), list(P.values()))
    worst = check_model_grads(g, "swin_s.train", P, grads, 5e-3)
    print("worst fp32 grad-norm deviation", worst)


def test_swin_s_drop_path_masks():
    """Train mode with drop_path 0.3: masks are the reference's own captured bernoulli draws."""
    g = Golden("g4_models")
    masks = torch.from_numpy(g.arr("swin_s.dp.masks").astype(np.float32))   # (48, B)
    assert masks.shape[0] == 2 * sum(M.SWIN_S["depths"]) - 2   # layer 0 has p == 0 -> no draw
    # layer 0: rate 0 -> DropPath is the identity and draws nothing (layer.py:173-174)
    per_layer = [(None, None)] + [(masks[2 * i], masks[2 * i + 1]) for i in range(masks.shape[0] // 2)]
    P = make_params(swin_state_dict(M.SWIN_S))
    x = fill((2, 3, 224, 224), 21, 1.0)
    out = M.swin_forward(P, x, M.SWIN_S, drop_masks=per_layer, drop_path=0.3)
    check_summary(out, g.rec("swin_s.dp.logits"), 2e-5, "swin drop-path logits")
    cot = fill(out.shape, name_seed("swin_s.dp.cot"), 1.0)
    grads = torch.autograd.grad((out * cot).sum(), list(P.values()), allow_unused=True)
    grads = [gr if gr is not None else torch.zeros_like(p) for gr, p in zip(grads, P.values())]
    check_model_grads(g, "swin_s.dp", P, grads, 5e-3)


@pytest.mark.parametrize("dtype,tag,ltol,gtol", [(torch.float64, "train64", 5e-7, 2e-6),
                                                 (torch.float32, "train", 2e-5, 5e-3)])
def test_vit_s16_full_model(dtype, tag, ltol, gtol):
    g = Golden("g4_models")
    shapes = vit_state_dict(M.VIT_S16)
    P = make_params(shapes, dtype)
    x = fill((2, 3, 224, 224), 21, 1.0).to(dtype)
    head = lambda f: R.linear(f, P["head.weight"], P["head.bias"])
    out = M.vit_forward(P, x, M.VIT_S16, head=head)
    check_summary(out, g.rec(f"vit_s16.{tag}.logits"), ltol, "vit logits")
    if dtype == torch.float32:
        check_summary(out, g.rec("vit_s16.eval.logits"), ltol, "vit eval logits")
    cot = fill(out.shape, name_seed(f"vit_s16.{tag}.cot"), 1.0, dtype=dtype)
    grads = torch.autograd.grad((out * cot).sum(), list(P.values()))
    check_model_grads(g, f"vit_s16.{tag}", P, grads, gtol)


def test_vit_multicrop():
    g = Golden("g5_multicrop")
    P = make_params(vit_state_dict(M.VIT_S16))   # head params present but unused here
    crops = [fill((1, 3, 224, 224), 31, 1.0), fill((1, 3, 224, 224), 32, 1.0),
             fill((1, 3, 96, 96), 33, 1.0), fill((1, 3, 96, 96), 34, 1.0)]
    out = M.vit_forward(P, crops, M.VIT_S16)
    check_summary(out, g.rec("multicrop.out"), 2e-5, "multicrop out")
    check_summary(M.vit_interpolate_pos(P["pos_embed"], 36), g.rec("multicrop.pos36"), 1e-6, "pos36")
    cot = fill(out.shape, name_seed("multicrop.cot"), 1.0)
    gp, gc, gw = torch.autograd.grad((out * cot).sum(), [P["pos_embed"], P["cls_token"],
                                                         P["patch_embedding.linear.weight"]])
    check_summary(gp, g.rec("multicrop.d.pos_embed"), 1e-4, "d pos_embed")
    check_summary(gc, g.rec("multicrop.d.cls_token"), 1e-4, "d cls_token")
    check_summary(gw, g.rec("multicrop.d.patch_w"), 1e-4, "d patch w")


def test_mix_loss_value_and_grad():
    g = Golden("g6_train_step")
    lg = fill((4, 1000), 51, 3.0).requires_grad_(True)
    t1 = torch.tensor([1, 500, 999, 0]); t2 = torch.tensor([7, 500, 3, 998])
    r = torch.tensor([0.1, 0.5, 1.0, 0.0])
    lv = R.mix_loss(lg, t1, t2, r, 0.1)
    assert abs(lv.item() - float(g.arr("mixloss.value"))) < 1e-5 * abs(float(g.arr("mixloss.value")))
    (gr,) = torch.autograd.grad(lv, [lg])
    check_summary(gr, g.rec("mixloss.grad"), 1e-5, "mixloss grad")
    # closed form of SURVEY's gradient contract: (softmax - target)/B
    K = 1000
    on, off = 1 - 0.1 + 0.1 / K, 0.1 / K
    d1 = torch.full((4, K), off); d1[torch.arange(4), t1] = on
    d2 = torch.full((4, K), off); d2[torch.arange(4), t2] = on
    td = r[:, None] * d1 + (1 - r[:, None]) * d2
    assert torch.allclose(gr, (torch.softmax(lg.detach(), -1) - td) / 4, atol=1e-7)


def test_one_train_step():
    """A13 counterpart: fwd -> MixLoss -> bwd -> clip 5.0 -> AdamW(lr 1e-3, wd 0.05 w/ 'vit' skip)."""
    g = Golden("g6_train_step")
    P = make_params(swin_state_dict(M.SWIN_S))
    x = fill((2, 3, 224, 224), 41, 1.0)
    l1 = torch.tensor([3, 977]); l2 = torch.tensor([977, 3])
    ratio = torch.tensor([0.3, 0.85])
    out = M.swin_forward(P, x, M.SWIN_S)
    loss = R.mix_loss(out, l1, l2, ratio, 0.1)
    assert abs(loss.item() - float(g.arr("loss"))) < 2e-5 * abs(float(g.arr("loss")))
    grads = torch.autograd.grad(loss, list(P.values()))
    grads, total = R.clip_grad_norm(list(grads), 5.0)
    assert abs(total.item() - float(g.arr("total_norm"))) < 1e-4 * float(g.arr("total_norm"))
    names = [str(n) for n in g.arr("param_names")]
    assert names == list(P.keys())
    ref_norms = g.arr("param_norms_after")
    new = {}
    for (n, p), gr, rn in zip(P.items(), grads, ref_norms):
        skip = ("bias" in n or "cls" in n or "norm" in n or p.ndim == 1)      # factory.py:33-34
        pn, _, _ = R.adamw_step(p.detach(), gr, torch.zeros_like(p), torch.zeros_like(p), 1,
                                1e-3, 0.9, 0.999, 1e-8, 0.0 if skip else 0.05)
        new[n] = pn
        got = pn.double().norm().item()
        assert abs(got - rn) <= 2e-5 * max(rn, 1e-6), f"{n}: {got} vs {rn}"
    for k in ("classifier.2.bias", "block1.0.attn.rel_pos.weight", "patch_embedding.linear.weight"):
        check_summary(new[k], g.rec("p." + k), 1e-4, k)
