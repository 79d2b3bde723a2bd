"""-m gpu: the drop-in nn.Modules (HIP path) vs the reference's outputs (goldens G4/G5/G6) and vs the CPU oracle.

Tolerances (relative L2 unless noted):
  fp32 mode  logits 1e-4, sampled grads / per-parameter grad norms 2e-3  (north_star: 1e-3 on logits/grads;
             the reference's OWN fp32-vs-fp64 noise on the deepest gradients is ~6e-4, see
             tests/test_oracle_models.py, so grads are compared against the fp64 reference run)
  bf16 mode  checked against the fp32 CPU oracle on a seeded default-init model (logits 1.5e-2, gradients 2e-2 =
             at most 2 x what is measured, see _bf16_vs_oracle).  The formula-weight goldens are NOT used for bf16: that network is
             deliberately ill-conditioned and the reference's own CPU bf16-autocast run is 21-30 % off on its
             logits (measured in the authoring container), so it pins nothing at bf16 precision
"""
import numpy as np
import pytest
import torch

from golden_util import Golden
from gpu_util import check, dev, relerr, report
from oracle import ref_models as M
from oracle import ref_ops as R
from oracle.formula import check_summary, fill, fill_state_dict, name_seed

pytestmark = pytest.mark.gpu


def _load(model):
    model.load_state_dict(fill_state_dict(model.state_dict()))
    return model.to(dev())


def _grad_norm_check(g, name, model, tol, what):
    names = [str(n) for n in g.arr(f"{name}.grad_names")]
    norms = g.arr(f"{name}.grad_norms")
    got = dict(model.named_parameters())
    assert names == list(got.keys()), "named_parameters() order differs from the reference"
    worst, worst_n = 0.0, ""
    for n, ref in zip(names, norms):
        assert got[n].grad is not None, f"{n} has no grad"
        assert got[n].grad.dtype == torch.float32 and got[n].grad.shape == got[n].shape
        v = got[n].grad.double().norm().item()
        rel = abs(v - ref) / max(ref, 1e-12)
        if rel > worst:
            worst, worst_n = rel, n
    assert report(f"{what}: worst per-param grad-norm deviation ({worst_n})", worst, tol)
    for k in g.keys(f"{name}.grad."):
        pn = k[len(f"{name}.grad."):]
        e = check_summary(got[pn].grad, g.rec(k), tol, k)
        report(f"{what}: grad {pn}", e, tol)


def _swin():
    from models import SwinTransformer
    return _load(SwinTransformer(**M.SWIN_S, drop_path=0.0))


def _vit(head=True):
    from models import VisionTransformer
    from vtx.nn import Linear
    h = Linear(384, 1000) if head else None
    return _load(VisionTransformer(h, 224, 16, 12, 384, 6, 1536, 0.0, 0.0, 0.0, 0.0))


def test_swin_s_fp32_vs_reference():
    g = Golden("g4_models")
    model = _swin()
    x = fill((2, 3, 224, 224), 21, 1.0).to(dev())
    model.eval()
    with torch.no_grad():
        out = model(x)
    e = check_summary(out, g.rec("swin_s.eval.logits"), 1e-4, "swin fp32 eval logits")
    report("swin-s fp32 eval logits vs reference", e, 1e-4)
    model.train()
    out = model(x)
    assert out.dtype == torch.float32 and out.shape == (2, 1000)
    e = check_summary(out, g.rec("swin_s.train64.logits"), 1e-4, "swin fp32 train logits")
    report("swin-s fp32 train logits vs fp64 reference", e, 1e-4)
    cot = fill(out.shape, name_seed("swin_s.train64.cot"), 1.0).to(dev())
    (out * cot).sum().backward()
    _grad_norm_check(g, "swin_s.train64", model, 2e-3, "swin-s fp32")


def _seeded_init(model, seed):
    """Reference-style random init (normal std 0.02 / LN ones) with a fixed seed; rel_pos made non-zero."""
    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    for k, v in sd.items():
        if not torch.is_floating_point(v):
            continue
        if "norm" in k or k.startswith("final_linear"):
            sd[k] = (1.0 + 0.05 * torch.randn(v.shape, generator=g)) if k.endswith("weight") else 0.02 * torch.randn(v.shape, generator=g)
        else:
            sd[k] = 0.02 * torch.randn(v.shape, generator=g)
    model.load_state_dict(sd)
    return {k: v.clone() for k, v in sd.items() if torch.is_floating_point(v)}


def _bf16_vs_oracle(model, oracle_fwd, sd, x, what):
    """HIP path under bf16 autocast vs the fp32 CPU oracle on the SAME seeded weights / inputs.

    Whole-model bf16 tolerances: logits 1.5e-2 (measured 6.7e-3 Swin-S / 9.4e-3 ViT-S/16; the reference's own
    bf16-autocast-vs-fp64 band on a default-init model is 8e-3..9.5e-3); gradients: relative L2 of the
    concatenation of all parameter gradients <= 2e-2 (measured 7.7e-3 / 1.1e-2) and median per-parameter relative
    L2 <= 2e-2 -- a regression that doubles the bf16 error fails.
    """
    model.to(dev()).train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x.to(dev()))
    assert out.dtype == torch.bfloat16
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = oracle_fwd(P, x)
    check(f"{what} bf16 logits vs fp32 oracle", out.float(), ref, 1.5e-2)
    cot = torch.randn(ref.shape, generator=torch.Generator().manual_seed(7))
    (out.float() * cot.to(dev())).sum().backward()
    names = [n for n, _ in model.named_parameters()]
    rg = torch.autograd.grad((ref * cot).sum(), [P[n] for n in names])
    num = den = 0.0
    per, tiny = [], []
    norms = [r.double().norm().item() / max(r.numel(), 1) ** 0.5 for r in rg]           # RMS of every reference gradient
    typical = float(np.median([v for v in norms if v > 0]))
    for n, r, rms in zip(names, rg, norms):
        gp = dict(model.named_parameters())[n].grad.double().cpu()
        d = (gp - r.double()).norm().item()
        num += d * d
        den += r.double().norm().item() ** 2
        if rms < 1e-6 * typical:
            # a parameter whose TRUE gradient is (numerically) zero -- Twins-SVT's stage-4 linear_q with one sub-sampled key:
            # softmax over one key is constant -- receives bf16 rounding noise; a relative error is meaningless there (the old
            # informational line printed 4e24).  Absolute bound instead: its RMS stays below 1e-3 of a typical gradient's RMS.
            tiny.append((n, d / max(r.numel(), 1) ** 0.5 / typical))
        else:
            per.append(d / r.double().norm().item())
    assert report(f"{what} bf16 all-parameter gradient rel-L2 vs oracle", (num / den) ** 0.5, 2e-2)
    assert report(f"{what} bf16 median per-parameter gradient rel-L2", float(np.median(per)), 2e-2)
    # every single parameter: 3x the whole-model tolerance (small tensors -- a 169 x h rel_pos table, a bias -- are noisier than the
    # concatenation; round-4 runs: worst 1.3e-2 ViT-S/16, 1.8e-2 Swin-S, 2.0e-2 PVT-Small, gpurun_out/parity.log)
    assert report(f"{what} bf16 worst per-parameter gradient rel-L2", float(np.max(per)), 6e-2)
    for n, v in tiny:
        assert report(f"{what} bf16 zero-gradient parameter {n}: RMS / typical gradient RMS", v, 1e-3)


def test_swin_s_bf16_autocast_vs_oracle():
    from models import SwinTransformer
    model = SwinTransformer(**M.SWIN_S, drop_path=0.0)
    sd = _seeded_init(model, 0)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    _bf16_vs_oracle(model, lambda P, xx: M.swin_forward(P, xx, M.SWIN_S), sd, x, "swin-s")


def test_vit_s16_bf16_autocast_vs_oracle():
    from models import VisionTransformer
    from vtx.nn import Linear
    model = VisionTransformer(Linear(384, 1000), 224, 16, 12, 384, 6, 1536, 0.0, 0.0, 0.0, 0.0)
    sd = _seeded_init(model, 2)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(3))
    fwd = lambda P, xx: M.vit_forward(P, xx, M.VIT_S16, head=lambda f: R.linear(f, P["head.weight"], P["head.bias"]))
    _bf16_vs_oracle(model, fwd, sd, x, "vit-s/16")


def test_swin_s_drop_path_with_reference_masks(monkeypatch):
    """Train mode, drop_path 0.3, with the masks the reference itself drew (captured in the golden)."""
    import models.swin_transformer as S
    g = Golden("g4_models")
    masks = torch.from_numpy(g.arr("swin_s.dp.masks").astype(np.float32)).to(dev())
    model = _swin()
    model.set_dropout(None, 0.3)
    model.train()
    rates = M.swin_drop_path_rates(M.SWIN_S["depths"], 0.3)
    it = iter(range(masks.shape[0]))

    def fake_scale(p, training, batch, device):
        if not training or p == 0:
            return None
        return masks[next(it)] / (1.0 - p)

    monkeypatch.setattr(S, "drop_path_scale", fake_scale)
    x = fill((2, 3, 224, 224), 21, 1.0).to(dev())
    out = model(x)
    e = check_summary(out, g.rec("swin_s.dp.logits"), 1e-4, "swin drop-path logits")
    report("swin-s fp32 drop-path logits vs reference", e, 1e-4)
    cot = fill(out.shape, name_seed("swin_s.dp.cot"), 1.0).to(dev())
    (out * cot).sum().backward()
    _grad_norm_check(g, "swin_s.dp", model, 5e-3, "swin-s fp32 drop-path")
    assert rates[0] == 0.0 and abs(rates[-1] - 0.2875) < 1e-12


def test_vit_s16_fp32_vs_reference():
    g = Golden("g4_models")
    model = _vit()
    x = fill((2, 3, 224, 224), 21, 1.0).to(dev())
    model.train()
    out = model(x)
    e = check_summary(out, g.rec("vit_s16.train64.logits"), 1e-4, "vit fp32 logits")
    report("vit-s/16 fp32 logits vs fp64 reference", e, 1e-4)
    cot = fill(out.shape, name_seed("vit_s16.train64.cot"), 1.0).to(dev())
    (out * cot).sum().backward()
    _grad_norm_check(g, "vit_s16.train64", model, 2e-3, "vit-s/16 fp32")


def test_vit_multicrop_fp32():
    g = Golden("g5_multicrop")
    model = _vit(head=False)
    model.train()
    crops = [fill((1, 3, 224, 224), 31, 1.0), fill((1, 3, 224, 224), 32, 1.0),
             fill((1, 3, 96, 96), 33, 1.0), fill((1, 3, 96, 96), 34, 1.0)]
    out = model([c.to(dev()) for c in crops])
    e = check_summary(out, g.rec("multicrop.out"), 1e-4, "multicrop out")
    report("vit multi-crop (224+96) fp32 output vs reference", e, 1e-4)
    cot = fill(out.shape, name_seed("multicrop.cot"), 1.0).to(dev())
    (out * cot).sum().backward()
    for key, p in (("multicrop.d.pos_embed", model.pos_embed), ("multicrop.d.cls_token", model.cls_token),
                   ("multicrop.d.patch_w", model.patch_embedding.linear.weight)):
        e = check_summary(p.grad, g.rec(key), 2e-3, key)
        report(f"vit multi-crop {key}", e, 2e-3)


def test_one_train_step_fp32_vs_reference():
    """A13 counterpart on the HIP modules: MixLoss -> backward -> clip 5.0 -> AdamW step (train.py:273-299)."""
    from vtx.train_step import MixLoss, make_param_groups
    g = Golden("g6_train_step")
    model = _swin()
    model.train()
    x = fill((2, 3, 224, 224), 41, 1.0).to(dev())
    l1 = torch.tensor([3, 977], device=dev()); l2 = torch.tensor([977, 3], device=dev())
    ratio = torch.tensor([0.3, 0.85], device=dev())
    opt = torch.optim.AdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
    out = model(x)
    loss = MixLoss(eps=0.1)(out, l1, l2, ratio)
    loss.backward()
    total = torch.nn.utils.clip_grad_norm_(list(model.parameters()), 5.0)
    assert report("train step: loss", abs(loss.item() - float(g.arr("loss"))) / abs(float(g.arr("loss"))), 1e-4)
    assert report("train step: total grad norm", abs(total.item() - float(g.arr("total_norm"))) /
                  float(g.arr("total_norm")), 2e-3)
    opt.step()
    names = [str(n) for n in g.arr("param_names")]
    ref = g.arr("param_norms_after")
    worst = 0.0
    for (n, p), rn in zip(model.named_parameters(), ref):
        worst = max(worst, abs(p.detach().double().norm().item() - rn) / max(rn, 1e-6))
    assert names == [n for n, _ in model.named_parameters()]
    assert report("train step: worst param-norm deviation after AdamW", worst, 1e-3)
    for k in ("classifier.2.bias", "block1.0.attn.rel_pos.weight", "patch_embedding.linear.weight"):
        p = dict(model.named_parameters())[k]
        e = check_summary(p, g.rec("p." + k), 2e-2, k)   # Adam's sign-like first step amplifies tiny grad noise
        report(f"train step: param {k} after step", e, 2e-2)


@pytest.mark.parametrize("bf16", [False, True])
def test_train_step_grad_accum_2_equals_one_double_batch_and_the_oracle(bf16):
    """VERDICT r2 #1b: `train_step(grad_accum=2)` on one GPU.  Two micro-batches of 2 images with loss / 2 each accumulate
    the gradient of the 4-image mean loss: compared with ONE 4-image step (SGD lr 1, no clip: parameter delta = -gradient)
    and, in fp32, with the CPU oracle's gradient of the concatenated batch."""
    from vtx.train_step import MixLoss, train_step
    dt = torch.bfloat16 if bf16 else None
    cfg = dict(image_size=(224, 224), n_class=16, depths=(1, 1, 2, 1), dims=(32, 64, 128, 256), dim_head=32,
               n_heads=(1, 2, 4, 8), dim_ffs=(128, 256, 512, 1024), window_size=7)
    from models import SwinTransformer
    torch.manual_seed(11)
    base = SwinTransformer(**cfg, drop_path=0.0)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(4, 3, 224, 224, generator=gen)
    l1 = torch.randint(0, 16, (4,), generator=gen)
    l2, ratio = l1.roll(1), torch.rand(4, generator=gen)
    data = tuple(t.to(dev()) for t in (x, l1, l2, ratio))

    def run(accum):
        m = SwinTransformer(**cfg, drop_path=0.0)
        m.load_state_dict(sd)
        m.to(dev()).train()
        opt = torch.optim.SGD(m.parameters(), lr=1.0)
        n = 4 // accum
        for i in range(accum):
            if accum > 1 and i == 0:
                with pytest.raises(ValueError, match="micro_step"):     # a defaulted index never stepped
                    train_step(m, MixLoss(0.1), opt, tuple(t[:n] for t in data), 0.0, dt, grad_accum=accum)
            train_step(m, MixLoss(0.1), opt, tuple(t[i * n:(i + 1) * n] for t in data), clip_grad_norm=0.0,
                       autocast_dtype=dt, grad_accum=accum, micro_step=i)
            if i + 1 < accum:
                assert all(p.grad is not None for p in m.parameters()), "non-boundary micro-batch must keep its gradients"
        assert all(p.grad is None for p in m.parameters())
        return {k: (sd[k].to(dev()) - p.detach()) for k, p in m.named_parameters()}

    g2, g1 = run(2), run(1)
    num = sum(((g2[k].double() - g1[k].double()).norm() ** 2).item() for k in g1)
    den = sum((g1[k].double().norm() ** 2).item() for k in g1)
    assert report(f"grad_accum 2 vs one double batch ({'bf16' if bf16 else 'fp32'}): all-parameter gradient",
                  (num / den) ** 0.5, 8e-3 if bf16 else 2e-5)
    if not bf16:
        P = {k: v.clone().requires_grad_(True) for k, v in sd.items() if torch.is_floating_point(v)}
        keys = [k for k, _ in base.named_parameters()]
        loss = R.mix_loss(M.swin_forward(P, x, cfg), l1, l2, ratio, 0.1)
        gr = dict(zip(keys, torch.autograd.grad(loss, [P[k] for k in keys])))
        num = sum(((g2[k].double().cpu() - gr[k].double()).norm() ** 2).item() for k in keys)
        den = sum((gr[k].double().norm() ** 2).item() for k in keys)
        assert report("grad_accum 2 vs CPU oracle on the concatenated batch: all-parameter gradient", (num / den) ** 0.5, 1e-4)


def test_state_dict_round_trip_and_cpu_refusal():
    from models import SwinTransformer
    from vtx._lib import VtxError
    m = SwinTransformer(**M.SWIN_S)
    sd = fill_state_dict(m.state_dict())
    m.load_state_dict(sd)          # strict: keys / shapes identical to the reference layout
    with pytest.raises(VtxError):
        m(torch.zeros(1, 3, 224, 224))   # CPU tensors: the product path refuses, it never falls back


def test_grad_allreduce_on_rccl_single_rank():
    """The DDP machinery (buckets, post-accumulate hooks, foreach copy, async all_reduce(AVG) on the RCCL
    process group's side stream, finish()) on a real GPU with a 1-rank nccl group: gradients must equal the
    plain backward bit for bit (AVG over one rank), and a second step must re-arm correctly."""
    import os
    import torch.distributed as dist
    from models import SwinTransformer
    from vtx.ddp import GradAllReduce
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev())
    try:
        cfg = dict(M.SWIN_S, depths=(1, 1, 2, 1))
        torch.manual_seed(0)
        model = SwinTransformer(**cfg).to(dev()).train()
        for m in model.modules():
            if hasattr(m, "rel_pos"):
                torch.nn.init.normal_(m.rel_pos.weight, std=0.02)
        x = torch.randn(4, 3, 224, 224, device=dev())
        with torch.autocast("cuda", dtype=torch.bfloat16):
            model(x).float().square().sum().backward()
        ref = [p.grad.clone() for p in model.parameters()]
        model.zero_grad(set_to_none=True)
        ddp = GradAllReduce(model, bucket_bytes=8 << 20, first_bucket_bytes=1 << 20, force=True)
        assert ddp.active and len(ddp.buckets) >= 3
        for _ in range(2):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                model(x).float().square().sum().backward()
            ddp.finish()
            for p, r in zip(model.parameters(), ref):
                assert torch.equal(p.grad, r)
            model.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_bf16_weights_follow_optimizer_updates(fused):
    """The bf16 GEMM operands must reflect the CURRENT fp32 parameters on every forward (like the reference's
    per-call autocast casts): torch's fused AdamW does not bump tensor version counters, `.data` EMA updates do not
    either -- any cache keyed on them would train on stale weights.  After each optimizer step (and after a `.data`
    update) the model's logits must equal, bit for bit, those of a freshly built model loaded with its state_dict."""
    import copy
    from models import SwinTransformer
    from vtx.train_step import MixLoss, train_step
    cfg = dict(image_size=(224, 224), n_class=16, depths=(1, 1, 2, 1), dims=(32, 64, 128, 256), dim_head=32,
               n_heads=(1, 2, 4, 8), dim_ffs=(128, 256, 512, 1024), window_size=7)
    torch.manual_seed(3)
    model = SwinTransformer(**cfg, drop_path=0.0).to(dev()).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, fused=fused)
    g = torch.Generator(device=dev()).manual_seed(5)
    x = torch.randn(4, 3, 224, 224, device=dev(), generator=g)
    l1 = torch.randint(0, 10, (4,), device=dev(), generator=g)
    data = (x, l1, l1.roll(1), torch.rand(4, device=dev(), generator=g))

    def logits(m):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return m(x).float()

    def fresh():
        m = SwinTransformer(**cfg, drop_path=0.0).to(dev()).train()
        m.load_state_dict(copy.deepcopy(model.state_dict()))
        return m

    prev = logits(model)
    for step in range(2):
        train_step(model, MixLoss(0.1), opt, data, clip_grad_norm=5.0)
        cur = logits(model)
        assert not torch.equal(cur, prev), "the optimizer step did not change the bf16 forward at all"
        assert torch.equal(cur, logits(fresh())), f"stale bf16 weights after step {step} (fused={fused})"
        prev = cur
    with torch.no_grad():                                        # EMA-style update through .data (train_util.py:70-84)
        for p in model.parameters():
            p.data.mul_(0.5)
    assert torch.equal(logits(model), logits(fresh())), "stale bf16 weights after a .data update"


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["swin", "vit", "pvt"])
def test_training_overfits_a_fixed_batch(family):
    """End-to-end sanity of the whole training path in bf16 (forward, backward, per-forward weight casts, clip, fused AdamW):
    a small model must drive the loss on one fixed batch far below the uniform-prediction level ln(16) = 2.77."""
    from vtx.optim import FusedAdamW
    from vtx.train_step import MixLoss, make_param_groups, train_step
    torch.manual_seed(0)
    if family == "swin":
        from models import SwinTransformer
        model = SwinTransformer(image_size=(224, 224), n_class=16, depths=(1, 1, 2, 1), dims=(32, 64, 128, 256), dim_head=32,
                                n_heads=(1, 2, 4, 8), dim_ffs=(128, 256, 512, 1024), window_size=7, drop_path=0.05)
    elif family == "vit":
        from models import VisionTransformer
        from vtx.nn import Linear
        model = VisionTransformer(Linear(128, 16), 224, 16, 3, 128, 2, 512, 0.0, 0.0, 0.0, 0.05)
    else:
        from models.pvt import PyramidVisionTransformer
        model = PyramidVisionTransformer(224, 16, 3, (1, 1, 2, 1), (64, 128, 320, 512), (1, 2, 5, 8), (256, 512, 640, 1024),
                                         (8, 4, 2, 1), drop_path=0.05)
    model.to(dev()).train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(16, 3, 224, 224, generator=g).to(dev())
    l1 = torch.arange(16).to(dev())
    data = (x, l1, l1, torch.ones(16, device=dev()))
    opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=2e-3)
    crit = MixLoss(0.0)
    losses = [train_step(model, crit, opt, data, clip_grad_norm=5.0).item() for _ in range(60)]
    assert losses[0] > 2.0 and all(np.isfinite(losses))
    assert report(f"{family}: loss after 60 bf16 steps on a fixed batch (start {losses[0]:.2f})", min(losses[-5:]), 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["swin", "vit", "pvt", "dino"])
def test_no_grad_forward_is_bitwise_the_training_graph_forward(family):
    """Under torch.no_grad() (evaluation, the DINO teacher) the layer functions skip what only a backward would read
    (the pre-activation z of the MLP): the outputs must not change by a bit."""
    torch.manual_seed(0)
    if family == "swin":
        from models import SwinTransformer
        model = SwinTransformer(image_size=(224, 224), n_class=16, depths=(1, 1, 2, 1), dims=(32, 64, 128, 256), dim_head=32,
                                n_heads=(1, 2, 4, 8), dim_ffs=(128, 256, 512, 1024), window_size=7, drop_path=0.0)
    elif family == "vit":
        from models import VisionTransformer
        from vtx.nn import Linear
        model = VisionTransformer(Linear(128, 16), 224, 16, 3, 128, 2, 512, 0.0, 0.0, 0.0, 0.0)
    elif family == "pvt":
        from models.pvt import PyramidVisionTransformer
        model = PyramidVisionTransformer(224, 16, 3, (1, 1, 2, 1), (64, 128, 320, 512), (1, 2, 5, 8), (256, 512, 640, 1024),
                                         (8, 4, 2, 1), drop_path=0.0)
    else:
        from models.vit import dino
        model = dino(224, 16, 2, 128, 2, 512, 0.0, 0.0, 0.0, 0.0, 1024, depth_head=3, dim_head_ff=256, dim_head_bottleneck=64)
    model.to(dev()).eval()
    x = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(2)).to(dev())
    for ac in (None, torch.bfloat16):
        with torch.autocast("cuda", dtype=ac, enabled=ac is not None):
            ref = model(x)
            with torch.no_grad():
                out = model(x)
        assert ref.requires_grad and not out.requires_grad
        assert torch.equal(ref.detach(), out), f"{family} {ac}: no-grad forward differs from the training-graph forward"


def test_vit_at_384_runs_beyond_224_tokens_vs_oracle():
    """ViT-S/16 at 384 x 384 (577 tokens: the fine-tuning resolution of reference vit.py:153-175 -- bicubic position
    resize + attention over more tokens than the register-resident kernels hold): fp32 features and gradients vs the
    oracle, bf16 features within the whole-model band."""
    from models import VisionTransformer
    torch.manual_seed(11)
    vit = VisionTransformer(None, 224, 16, 2, 384, 6, 1536, 0.0, 0.0, 0.0, 0.0)
    sd = {k: v.detach().clone() for k, v in vit.state_dict().items()}
    vit.to(dev()).train()
    x = torch.randn(2, 3, 384, 384, generator=torch.Generator().manual_seed(12))
    cot = torch.randn(2, 384, generator=torch.Generator().manual_seed(13))
    f = vit(x.to(dev()))
    (f * cot.to(dev())).sum().backward()
    P = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ref = M.vit_forward(P, x.double(), dict(M.VIT_S16, depth=2))
    names = [n for n, _ in vit.named_parameters()]
    rg = torch.autograd.grad((ref * cot.double()).sum(), [P[n] for n in names])
    check("vit 384^2 (577 tokens) fp32 features", f, ref, 1e-4)
    num = den = 0.0
    for (n, p), r in zip(vit.named_parameters(), rg):
        num += (p.grad.double().cpu() - r).norm().item() ** 2
        den += r.norm().item() ** 2
    assert report("vit 384^2 fp32 all-parameter gradient rel-L2", (num / den) ** 0.5, 1e-3)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        fb = vit(x.to(dev()))
    check("vit 384^2 bf16 features", fb.float(), ref, 1.5e-2)


def test_swin_at_384_with_12x12_windows_vs_oracle():
    """Swin at 384 x 384 with window 12 (the fine-tuning geometry; reference models/swin_transformer.py:103-160 runs at any window):
    144 tokens per window with a 23 x 23 relative-position table and the shifted-window mask -- beyond the 7 x 7 fast path, on the
    generic window kernels (10 key tiles, bias gradient over 3 x 10 register tiles per wave).  fp32 logits and every parameter
    gradient vs the oracle (rel_pos tables made non-zero so that the bias and its gradient are exercised); bf16 within the band."""
    from models import SwinTransformer
    cfg = dict(image_size=(384, 384), n_class=10, depths=(1, 1, 2, 1), dims=(96, 192, 384, 768), dim_head=32,
               n_heads=(3, 6, 12, 24), dim_ffs=(384, 768, 1536, 3072), window_size=12)
    model = SwinTransformer(**cfg)
    sd = _seeded_init(model, 21)
    gen = torch.Generator().manual_seed(22)
    for k in sd:
        if k.endswith("rel_pos.weight"):
            sd[k] = 0.5 * torch.randn(sd[k].shape, generator=gen)
    model.load_state_dict(sd, strict=False)            # (sd: the floating tensors; pos / local_mask keep the constructor's tables)
    model.to(dev()).train()
    x = torch.randn(2, 3, 384, 384, generator=torch.Generator().manual_seed(23))
    out = model(x.to(dev()))
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items() if torch.is_floating_point(v)}
    ref = M.swin_forward(P, x, cfg)
    check("swin 384^2 window 12 fp32 logits vs oracle", out, ref, 1e-4)
    cot = torch.randn(ref.shape, generator=torch.Generator().manual_seed(24))
    (out * cot.to(dev())).sum().backward()
    names = [n for n, _ in model.named_parameters()]
    rg = torch.autograd.grad((ref * cot).sum(), [P[n] for n in names])
    got = dict(model.named_parameters())
    for n, r in zip(names, rg):
        check(f"swin 384^2 window 12 fp32 grad {n}", got[n].grad, r, 2e-3)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ob = model(x.to(dev()))
    check("swin 384^2 window 12 bf16 logits", ob.float(), ref, 2e-2)


@pytest.mark.parametrize("family", ["swin_s", "vit_s16", "pvt_small", "twins_svt_s"])
def test_full_size_bf16_step_is_deterministic_finite_and_matches_the_chunked_path(family):
    """VERDICT r2 (weak 2 / next 8): the composition that bench.py TIMES -- Swin-S B = 128 drop_path 0.3, ViT-S/16 B = 256,
    bf16, weight gradients on the side stream, tens of GB of live activations, persistent attention grids spanning many
    rounds -- as a property test (no oracle finishes at this size in seconds):
      * two seeded runs of two train steps end with bitwise-equal parameters (every reduction has a fixed order, the side
        stream only moves launches);
      * every gradient of a full-size backward is finite and no parameter's gradient is missing;
      * the full-batch forward equals the same model applied to chunks of 8 images -- the small-batch path that the golden
        / oracle tests pin -- to bf16 tolerance (logits), and the MixLoss values agree."""
    import bench
    from vtx import functional as VF
    from vtx.optim import FusedAdamW
    from vtx.train_step import MixLoss, make_param_groups, train_step
    B = bench.default_batch(family)
    dp = 0.3 if family == "swin_s" else 0.1
    assert VF._SIDE_ENABLED, "the timed configuration runs the weight gradients on the side stream"
    g = torch.Generator(device=dev()).manual_seed(77)
    x = torch.randn(B, 3, 224, 224, device=dev(), generator=g)
    l1 = torch.randint(0, 1000, (B,), device=dev(), generator=g)
    data = (x, l1, l1.roll(1), torch.rand(B, device=dev(), generator=g))

    def run():
        torch.manual_seed(5)
        model = bench.build_model(family, dp).to(dev()).train()
        opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
        torch.manual_seed(6)                                    # the DropPath draws of the two steps
        losses = [train_step(model, MixLoss(0.1), opt, data, clip_grad_norm=5.0).item() for _ in range(2)]
        return model, losses

    m1, loss1 = run()
    m2, loss2 = run()
    assert loss1 == loss2 and all(np.isfinite(loss1)), (loss1, loss2)
    for (n, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), f"{family}: parameter {n} differs between two seeded runs of the full-size step"
    del m2
    # full-size backward: all gradients present and finite
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = MixLoss(0.1)(m1(x), *data[1:])
    with VF.deferred_wgrad(True):
        loss.backward()
    bad = [n for n, p in m1.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all().item()]
    assert not bad, f"{family}: missing / non-finite gradients at full size: {bad[:6]}"
    # full batch vs chunks of 8 (eval: DropPath off -- the masks of a chunked run would be different draws)
    m1.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        full = m1(x).float()
        parts = torch.cat([m1(x[i:i + 8]).float() for i in range(0, B, 8)])
        lf = MixLoss(0.1)(full, *data[1:]).item()
        lp = np.mean([MixLoss(0.1)(parts[i:i + 8], *(t[i:i + 8] for t in data[1:])).item() for i in range(0, B, 8)])
    check(f"{family} B = {B} bf16 logits: full batch vs chunks of 8", full, parts, 4e-3)
    assert report(f"{family} B = {B} MixLoss: full batch vs mean over chunks of 8", abs(lf - lp) / abs(lp), 1e-3)


@pytest.mark.parametrize("family", ["swin_s", "vit_s16", "pvt_small"])
def test_full_size_step_is_bitwise_the_same_on_the_tiled_and_the_a_stationary_gemm(family):
    """Round 4: csrc/gemm_astat.hip computes every output element with the tiled kernel's expression and k order, so a full-size
    training step (B = 128 / 256, DropPath compaction on for Swin-S: row-mapped launches at whatever row counts the draws leave,
    copy-only rows, DropPath scales) must give the SAME BITS with GEMM_ASTAT = 0 and 1 -- logits, loss and every parameter after
    the optimizer step.  (This is the test that would have caught the scale-table indexing bug of the first integration: the
    shape-level tests only reached mapped launches whose row count was a multiple of 128.)"""
    import bench
    from vtx import options
    from vtx.optim import FusedAdamW
    from vtx.train_step import MixLoss, make_param_groups, train_step
    B = bench.default_batch(family)
    dp = 0.3 if family == "swin_s" else 0.1
    g = torch.Generator(device=dev()).manual_seed(78)
    x = torch.randn(B, 3, 224, 224, device=dev(), generator=g)
    l1 = torch.randint(0, 1000, (B,), device=dev(), generator=g)
    data = (x, l1, l1.roll(1), torch.rand(B, device=dev(), generator=g))

    def run(astat):
        with options.override(GEMM_ASTAT=astat):
            torch.manual_seed(5)
            model = bench.build_model(family, dp).to(dev()).train()
            opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
            torch.manual_seed(6)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                logits = model(x).float()
            torch.manual_seed(6)
            loss = train_step(model, MixLoss(0.1), opt, data, clip_grad_norm=5.0).item()
        return model, logits, loss

    m0, y0, l0 = run(0)
    m1, y1, l1_ = run(1)
    assert torch.equal(y0, y1), f"{family}: logits differ in {(y0 != y1).sum().item()} elements"
    assert l0 == l1_, (l0, l1_)
    for (n, a), (_, b) in zip(m0.named_parameters(), m1.named_parameters()):
        assert torch.equal(a, b), f"{family}: parameter {n} differs after one step"


def test_full_size_dino_step_is_deterministic_and_finite():
    """VERDICT r3 (weak 1): the DINO shape bench.py's `secondary` entry times -- 64 images x (2 x 224^2 + 8 x 96^2 crops), 65 536-way
    head, teacher on its own stream, shared-parameter gradient accumulation inside the reduce launches -- as a property test: two
    seeded runs of two steps end with bitwise-equal student AND teacher parameters and the same finite losses; every student
    gradient of a full-size backward is finite."""
    import bench
    from vtx.dino import DINOLoss, dino_train_step
    from vtx.optim import FusedAdamW
    from vtx.train_step import make_param_groups
    B = bench.default_batch("dino")
    g = torch.Generator(device=dev()).manual_seed(78)
    crops = [torch.randn(B, 3, 224, 224, device=dev(), generator=g) for _ in range(2)] + \
            [torch.randn(B, 3, 96, 96, device=dev(), generator=g) for _ in range(8)]

    def run():
        torch.manual_seed(5)
        student = bench.build_model("dino", 0.1).to(dev()).train()
        teacher = bench.build_model("dino", 0.0).to(dev()).train()
        teacher.load_state_dict(student.state_dict())
        for p in teacher.parameters():
            p.requires_grad = False
        crit = DINOLoss(65536, 10, 0.04, 0.07, 30, 300).to(dev())
        opt = FusedAdamW(make_param_groups(student.named_parameters(), 0.04, "dino"), lr=5e-4)
        torch.manual_seed(6)
        losses = [dino_train_step(student, teacher, crit, opt, crops, epoch=1, momentum=0.996, clip_grad_norm=3.0,
                                  freeze_last_layer=1, autocast_dtype=torch.bfloat16).item() for _ in range(2)]
        return student, teacher, crit, losses

    s1, t1, c1, loss1 = run()
    s2, t2, c2, loss2 = run()
    assert loss1 == loss2 and all(np.isfinite(loss1)), (loss1, loss2)
    for (n, a), (_, b) in zip(s1.named_parameters(), s2.named_parameters()):
        assert torch.equal(a, b), f"dino: student parameter {n} differs between two seeded runs of the full-size step"
    for (n, a), (_, b) in zip(t1.named_parameters(), t2.named_parameters()):
        assert torch.equal(a, b), f"dino: teacher parameter {n} differs between two seeded runs"
    assert torch.equal(c1.center, c2.center)
    assert all(torch.isfinite(p).all().item() for p in s1.parameters())


@pytest.mark.parametrize("family", ["vit", "swin"])
def test_residual_ff_and_positional_dropout_run_on_the_composed_path(family):
    """VERDICT r2 missing #3: dropout rates > 0 (vit.py:39,52-61, swin_transformer.py:144, layer.py:194) were accepted by the
    constructors and refused at run time.  Residual, feed-forward and positional dropout now run (HIP modules composed as the
    reference composes them, nn.Dropout on the device tensors in between): the train-mode step is finite and seed-
    deterministic, actually drops something, every parameter gets a gradient -- and in eval mode the model is bit-identical
    to the same weights with all rates 0 (the fused layers).  Attention-probability dropout stays refused (the
    probabilities never leave the fused kernel)."""
    torch.manual_seed(51)
    if family == "vit":
        from models import VisionTransformer
        from vtx.nn import Linear
        mk = lambda p: VisionTransformer(Linear(128, 16), 224, 16, 2, 128, 2, 512, p, 0.0, p, 0.1)
    else:
        from models import SwinTransformer
        mk = lambda p: SwinTransformer(image_size=(224, 224), n_class=16, depths=(1, 1, 2, 1), dims=(32, 64, 128, 256), dim_head=32,
                                       n_heads=(1, 2, 4, 8), dim_ffs=(128, 256, 512, 1024), window_size=7, drop_ff=p, drop_path=0.1)
    model, plain = mk(0.2).to(dev()), mk(0.0).to(dev())
    plain.load_state_dict(model.state_dict())
    x = torch.randn(4, 3, 224, 224, device=dev())

    def step(m, seed):
        m.train()
        m.zero_grad(set_to_none=True)
        torch.manual_seed(seed)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(x)
        out.float().square().mean().backward()
        return out.detach().float()

    a, b, c = step(model, 7), step(model, 7), step(model, 8)
    assert torch.isfinite(a).all() and torch.equal(a, b) and not torch.equal(a, c)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    assert not torch.equal(a, step(plain, 7)), "dropout 0.2 must change the training forward"
    model.eval(); plain.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        assert torch.equal(model(x), plain(x)), "eval mode: dropout is the identity, the fused layers run"
    if family == "vit":
        from models import VisionTransformer
        from vtx.nn import Linear
        # attention-probability dropout (vit.py:39) runs on the dropout variant of the attention kernels (tests/test_gpu_attn_dropout.py)
        da = VisionTransformer(Linear(128, 16), 224, 16, 1, 128, 2, 512, 0.0, 0.1, 0.0, 0.0).to(dev()).train()
        assert torch.isfinite(da(x).float()).all()
