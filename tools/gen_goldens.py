#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE on CPU.

Authoring-container only: imports /root/reference (read-only, never copied, never
shipped).  The fixtures hold the reference's OUTPUTS (and captured DropPath masks);
all inputs/weights are closed-form (oracle/formula.py), so nothing of the reference
travels.  Re-run:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_goldens.py

Golden ids follow SURVEY.md section 8(c): G1 pos tables, G2 local masks, G3 per-module
fwd+bwd vectors, G4 full-model logits + grad norms, G5 ViT multi-crop, G6 one train step, G7 PVT-Small (F1), G8 DINO head + loss (F2),
G9 mixup / cutmix / RandomErasing outputs (F4), G10 Twins-SVT (the row after F1-F4), G11 attention-probability dropout with an
injected keep mask (vit.py:39, swin_transformer.py:144, pvt.py:60, twins.py:88,147), G12 halo attention + a small HaloTransformer
(models/halo_transformer.py; SURVEY section 8 row F4's last sentence).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("VTX_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import numpy as np
import torch

# --- the reference only uses tensorfn.config.config_model as a registration decorator
tf = types.ModuleType("tensorfn")
tfc = types.ModuleType("tensorfn.config")
tfc.config_model = lambda *a, **k: (lambda f: f)
tf.config = tfc
sys.modules["tensorfn"] = tf
sys.modules["tensorfn.config"] = tfc

import warnings
warnings.filterwarnings("ignore")

from models import swin_transformer as ref_swin   # noqa: E402  (reference)
from models import vit as ref_vit                 # noqa: E402  (reference)
from models import layer as ref_layer             # noqa: E402  (reference)
from models import pvt as ref_pvt                 # noqa: E402  (reference)
from models import twins as ref_twins             # noqa: E402  (reference)
from models import halo_transformer as ref_halo   # noqa: E402  (reference)
import loss as ref_loss                           # noqa: E402  (reference)

from oracle.formula import fill, fill_state_dict, summarize, name_seed  # noqa: E402
from oracle.ref_models import HALO_TINY, PVT_SMALL, SWIN_S, TWINS_SVT_S, VIT_S16     # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def save(name, rec):
    flat = {}
    for k, v in rec.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}/{kk}"] = np.asarray(vv)
        else:
            flat[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(flat)} arrays)")


def load_formula(module, rel_pos_scale=0.5):
    module.load_state_dict(fill_state_dict(module.state_dict(), rel_pos_scale))
    return module


def grads_of(module):
    return {n: p.grad for n, p in module.named_parameters() if p.grad is not None}


# ------------------------------------------------------------------ G1 / G2
def gen_tables():
    rec = {}
    cases = [((56, 56), 7), ((28, 28), 7), ((14, 14), 7), ((7, 7), 7), ((8, 12), 4),
             ((12, 8), 4), ((6, 6), 3), ((10, 15), 5), ((16, 16), 8)]
    for (size, w) in cases:
        for shift in (False, True):
            m = ref_swin.MultiHeadedLocalAttention(32, 1, 32, size, w, shift)
            key = f"{size[0]}x{size[1]}_w{w}_s{int(shift)}"
            pos = m.pos.numpy()
            assert pos.min() >= 0 and pos.max() < 32767
            rec[f"pos_{key}"] = pos.astype(np.int16)
            if shift:
                lm = m.local_mask.numpy()
                rec[f"maskshape_{key}"] = np.array(lm.shape, dtype=np.int64)
                rec[f"mask_{key}"] = np.packbits(lm.reshape(-1))
    save("g1_g2_tables", rec)


# ------------------------------------------------------------------ G3
def run_module(mod, x, name, rec, extra_inputs_grad=True):
    x = x.clone().requires_grad_(extra_inputs_grad)
    out = mod(x)
    cot = fill(out.shape, name_seed(name + ".cot"), 1.0)
    (out * cot).sum().backward()
    rec[f"{name}.out"] = summarize(out)
    if extra_inputs_grad:
        rec[f"{name}.dx"] = summarize(x.grad)
    for n, g in grads_of(mod).items():
        rec[f"{name}.d.{n}"] = summarize(g)


def gen_modules():
    rec = {}
    # window attention: generic shifted / unshifted + the single-window wrap case (stage-4 quirk)
    for (size, shift, tag) in [((14, 14), True, "s1"), ((14, 14), False, "s0"), ((7, 7), True, "wrap")]:
        m = load_formula(ref_swin.MultiHeadedLocalAttention(96, 3, 32, size, 7, shift)).double()
        x = fill((2, size[0], size[1], 96), 11, 1.0, dtype=torch.float64)
        run_module(m, x, f"local_attn_{tag}", rec)
    # global attention L=197 and L=37
    for L in (197, 37):
        m = load_formula(ref_vit.MultiHeadedAttention(384, 6)).double()
        x = fill((2, L, 384), 12, 1.0, dtype=torch.float64)
        run_module(m, x, f"global_attn_L{L}", rec)
    m = load_formula(ref_layer.PositionwiseFeedForward(96, 384)).double()
    run_module(m, fill((2, 49, 96), 13, 1.0, dtype=torch.float64), "ffn", rec)
    m = load_formula(ref_vit.PatchEmbedding(3, 384, 16)).double()
    run_module(m, fill((2, 3, 224, 224), 14, 1.0, dtype=torch.float64), "vit_patch", rec, False)
    m = load_formula(ref_swin.PatchEmbedding(3, 96, 4)).double()
    xin = fill((2, 3, 224, 224), 14, 1.0, dtype=torch.float64).permute(0, 2, 3, 1).contiguous()
    run_module(m, xin, "swin_patch", rec, False)
    m = load_formula(ref_swin.PatchMerge(96, 192, 2)).double()
    run_module(m, fill((2, 14, 14, 96), 15, 1.0, dtype=torch.float64), "patch_merge", rec)
    for eps, tag in ((1e-6, "e6"), (1e-5, "e5")):
        m = torch.nn.LayerNorm(96, eps=eps)
        m.load_state_dict({"weight": fill((96,), 3, 0.1, 1.0), "bias": fill((96,), 4, 0.02)})
        run_module(m.double(), fill((2, 49, 96), 16, 2.0, 0.3, dtype=torch.float64), f"ln_{tag}", rec)
    # a full Swin TransformerLayer (LN + attn + residual + LN + ffn + residual)
    m = load_formula(ref_swin.TransformerLayer(96, 3, 32, 384, (14, 14), 7, True)).double()
    run_module(m, fill((2, 14, 14, 96), 17, 1.0, dtype=torch.float64), "swin_layer", rec)
    m = load_formula(ref_vit.TransformerLayer(384, 6, 1536, 0.0, 0.0, 0.0, 0.0)).double()
    run_module(m, fill((2, 197, 384), 18, 1.0, dtype=torch.float64), "vit_layer", rec)
    save("g3_modules", rec)


# ------------------------------------------------------------------ G4
def model_record(model, x, name, rec, train):
    model.train(train)
    model.zero_grad(set_to_none=True)
    out = model(x)
    rec[f"{name}.logits"] = summarize(out)
    if train:
        cot = fill(out.shape, name_seed(name + ".cot"), 1.0, dtype=out.dtype)
        (out * cot).sum().backward()
        names, norms = [], []
        for n, p in model.named_parameters():
            names.append(n)
            norms.append(p.grad.double().norm().item())
        rec[f"{name}.grad_names"] = np.array(names)
        rec[f"{name}.grad_norms"] = np.array(norms, dtype=np.float64)
        for n, p in model.named_parameters():
            if any(s in n for s in ("patch_embedding.linear.weight", "rel_pos", "cls_token",
                                    "pos_embed", "block3.5.attn.weight.weight",
                                    "layers.5.attn.qkv.weight", "classifier.2.weight",
                                    "block2.0.linear.weight", "head.weight")):
                rec[f"{name}.grad.{n}"] = summarize(p.grad)


def capture_bernoulli():
    """Record every DropPath mask the reference draws (layer.py:177)."""
    masks = []
    orig = torch.Tensor.bernoulli_

    def wrapped(self, *a, **k):
        r = orig(self, *a, **k)
        masks.append(r.detach().clone().reshape(-1))
        return r

    torch.Tensor.bernoulli_ = wrapped
    return masks, (lambda: setattr(torch.Tensor, "bernoulli_", orig))


def gen_models():
    rec = {}
    x = fill((2, 3, 224, 224), 21, 1.0)
    swin = load_formula(ref_swin.SwinTransformer(**SWIN_S, drop_path=0.0))
    model_record(swin, x, "swin_s.eval", rec, False)
    model_record(swin, x, "swin_s.train", rec, True)
    # fp64 run of the same model = the noise-floor reference for tolerances
    swin64 = load_formula(ref_swin.SwinTransformer(**SWIN_S, drop_path=0.0)).double()
    model_record(swin64, x.double(), "swin_s.train64", rec, True)
    # with DropPath (masks captured from the reference's own RNG draws)
    swin.set_dropout(None, 0.3)
    masks, restore = capture_bernoulli()
    torch.manual_seed(1234)
    try:
        model_record(swin, x, "swin_s.dp", rec, True)
    finally:
        restore()
    rec["swin_s.dp.masks"] = torch.stack(masks).numpy().astype(np.uint8)
    print("captured", len(masks), "drop-path masks")

    head = torch.nn.Linear(384, 1000)
    vit = ref_vit.VisionTransformer(head, VIT_S16["image_size"], VIT_S16["window_size"],
                                    VIT_S16["depth"], VIT_S16["dim"], VIT_S16["n_head"],
                                    VIT_S16["dim_ff"], 0.0, 0.0, 0.0, 0.0)
    load_formula(vit)
    model_record(vit, x, "vit_s16.eval", rec, False)
    model_record(vit, x, "vit_s16.train", rec, True)
    vit64 = ref_vit.VisionTransformer(torch.nn.Linear(384, 1000), 224, 16, 12, 384, 6, 1536,
                                      0.0, 0.0, 0.0, 0.0)
    load_formula(vit64).double()
    model_record(vit64, x.double(), "vit_s16.train64", rec, True)
    save("g4_models", rec)

    # ---------------------------------------------------------------- G5 multi-crop
    rec = {}
    vit.head = None
    crops = [fill((1, 3, 224, 224), 31, 1.0), fill((1, 3, 224, 224), 32, 1.0),
             fill((1, 3, 96, 96), 33, 1.0), fill((1, 3, 96, 96), 34, 1.0)]
    vit.train(True)
    vit.zero_grad(set_to_none=True)
    out = vit(crops)
    rec["multicrop.out"] = summarize(out)
    cot = fill(out.shape, name_seed("multicrop.cot"), 1.0)
    (out * cot).sum().backward()
    rec["multicrop.d.pos_embed"] = summarize(vit.pos_embed.grad)
    rec["multicrop.d.cls_token"] = summarize(vit.cls_token.grad)
    rec["multicrop.d.patch_w"] = summarize(vit.patch_embedding.linear.weight.grad)
    pe = vit.interpolate_pos_embedding(torch.zeros(1, 37, 384), vit.pos_embed)
    rec["multicrop.pos36"] = summarize(pe)
    save("g5_multicrop", rec)


# ------------------------------------------------------------------ G6
# ------------------------------------------------------------------ G7 (PVT-Small)
def gen_pvt():
    rec = {}
    # module level: spatial-reduction attention, dim 128 / 2 heads / reduction 4 on a 28 x 28 token grid
    att = load_formula(ref_pvt.MultiHeadedAttention(128, 2, reduction=4)).double()      # fp64: tight pins
    x = fill((2, 784, 128), 71, 1.0).double().requires_grad_(True)
    out, _ = att(x, 28, 28)
    cot = fill(out.shape, 72, 1.0).double()
    (out * cot).sum().backward()
    rec["sr_attn.out"] = summarize(out)
    rec["sr_attn.dx"] = summarize(x.grad)
    for n, p in att.named_parameters():
        rec[f"sr_attn.grad.{n}"] = summarize(p.grad)
    # stage-4 style: cls token, no reduction
    att1 = load_formula(ref_pvt.MultiHeadedAttention(512, 8, reduction=1)).double()
    x1 = fill((2, 50, 512), 73, 1.0).double().requires_grad_(True)
    out1, _ = att1(x1, 7, 7)
    (out1 * fill(out1.shape, 74, 1.0).double()).sum().backward()
    rec["attn_r1.out"] = summarize(out1)
    rec["attn_r1.dx"] = summarize(x1.grad)
    # patch embedding with cls token on token-major features (stage 4 geometry)
    pe = load_formula(ref_pvt.PatchEmbedding((14, 14), 320, 512, 2, cls_token=True)).double()
    xi = fill((2, 320, 14, 14), 75, 1.0).double().requires_grad_(True)
    po, hw = pe(xi)
    (po * fill(po.shape, 76, 1.0).double()).sum().backward()
    rec["patch_embed.out"] = summarize(po)
    rec["patch_embed.dx"] = summarize(xi.grad)
    for n, p in pe.named_parameters():
        rec[f"patch_embed.grad.{n}"] = summarize(p.grad)
    # full model
    x = fill((2, 3, 224, 224), 21, 1.0)
    pvt = load_formula(ref_pvt.PyramidVisionTransformer(**PVT_SMALL, drop_path=0.0))
    rec["pvt_small.n_params"] = np.array(sum(p.numel() for p in pvt.parameters()))
    rec["pvt_small.state_keys"] = np.array(list(pvt.state_dict().keys()))
    rec["pvt_small.state_shapes"] = np.array([str(tuple(v.shape)) for v in pvt.state_dict().values()])
    model_record(pvt, x, "pvt_small.eval", rec, False)
    model_record(pvt, x, "pvt_small.train", rec, True)
    for n, p in pvt.named_parameters():
        if any(s in n for s in ("patch_embedding.0.conv.weight", "patch_embedding.3.cls_token", "patch_embedding.2.pos",
                                "block2.1.attn.reduce_conv.weight", "block3.2.attn.linear_kv.weight",
                                "block1.0.attn.linear_q.weight", "classifier.weight")):
            rec[f"pvt_small.train.grad.{n}"] = summarize(p.grad)
    pvt64 = load_formula(ref_pvt.PyramidVisionTransformer(**PVT_SMALL, drop_path=0.0)).double()
    model_record(pvt64, x.double(), "pvt_small.train64", rec, True)
    save("g7_pvt", rec)


# ------------------------------------------------------------------ G8 (DINO head + loss)
def gen_dino():
    import torch.distributed as tdist
    rec = {}
    head = ref_vit.DINOHead(384, 4096, norm_last_layer=False)
    sd = fill_state_dict(head.state_dict())
    sd["last.weight_g"] = fill(sd["last.weight_g"].shape, name_seed("last.weight_g"), 0.3, 1.0)   # away from the all-ones init
    head.load_state_dict(sd)
    head.double()
    x = fill((6, 384), 81, 1.0).double().requires_grad_(True)
    out = head(x)
    (out * fill(out.shape, 82, 1.0).double()).sum().backward()
    rec["head.out"] = summarize(out)
    rec["head.dx"] = summarize(x.grad)
    for n, p in head.named_parameters():
        rec[f"head.grad.{n}"] = summarize(p.grad)
    rec["head.param_names"] = np.array([n for n, _ in head.named_parameters()])
    # DINOLoss: 4 crops (2 global + 2 local), B = 3, K = 4096; epoch 5 of a 30-epoch teacher-temperature warm-up
    if not tdist.is_initialized():
        tdist.init_process_group("gloo", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1)
    crit = ref_loss.DINOLoss(4096, 4, 0.04, 0.07, 30, 100).double()
    crit.center.copy_(fill((1, 4096), 83, 0.2).double())
    student = fill((12, 4096), 84, 2.0).double().requires_grad_(True)
    teacher = fill((6, 4096), 85, 2.0).double()
    loss = crit(student, teacher, 5)
    loss.backward()
    rec["loss.value"] = np.array(loss.item())
    rec["loss.teacher_temp"] = np.array(crit.teacher_temperature_schedule[5])
    rec["loss.dstudent"] = summarize(student.grad)
    rec["loss.center_after"] = summarize(crit.center)
    save("g8_dino", rec)


# ------------------------------------------------------------------ G9 (mixup / cutmix / erasing)
def gen_input_pipeline():
    import random
    tv = types.ModuleType("torchvision")          # transforms.py imports torchvision only for the PIL pipelines
    tvt = types.ModuleType("torchvision.transforms")

    class _Any:
        def __init__(self, *a, **k):
            pass

    def _ga(n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Any
    tvt.__getattr__ = _ga
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    import mix_dataset as ref_mix                 # (reference)
    import transforms as ref_tf                   # (reference)
    rec = {}
    n, h, w = 8, 16, 20                           # non-square: pins the (w, h) = (H, W) quirk of rand_bbox on tensors
    images = [fill((3, h, w), 900 + i, 0.5, 0.5) for i in range(n)]
    labels = list(range(10, 10 + n))
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    for tag, mixup, cutmix, seed in (("both", 0.2, 1, 5), ("beta_cutmix", 0.0, 0.5, 6), ("mixup_only", 0.8, 0, 7)):
        erase = ref_tf.RandomErasing(p=0.7, max_count=2, mode="const", device="cpu")
        class Fresh:                              # like a decoding dataset: a new tensor per access (MixDataset's cutmix
            def __len__(self):                    # writes into the tensor it was handed, mix_dataset.py:78)
                return n

            def __getitem__(self, i):
                return images[i].clone(), labels[i]
        md = ref_mix.MixDataset(Fresh(), lambda img: erase((img - mean) / std), mixup=mixup, cutmix=cutmix)
        random.seed(seed)
        outs, l1, l2, ratio = [], [], [], []
        for i in range(n):
            img, a, b, r = md[i]
            outs.append(img.numpy()); l1.append(a); l2.append(b); ratio.append(float(r))
        rec[f"{tag}.images"] = np.stack(outs)
        rec[f"{tag}.label1"] = np.array(l1)
        rec[f"{tag}.label2"] = np.array(l2)
        rec[f"{tag}.ratio"] = np.array(ratio, dtype=np.float64)
    save("g9_input_pipeline", rec)
    # G9b: the 'pixel' (what factory.py:177-181 configures) and 'rand' colour modes -- normal draws from torch's global
    # CPU generator, interleaved with the python `random` draws of the rectangles
    rec = {}
    for tag, mixup, cutmix, seed, mode in (("pixel", 0.2, 1, 8, "pixel"), ("rand", 0.2, 1, 9, "rand"),
                                           ("pixel_only", 0.0, 0, 10, "pixel")):
        erase = ref_tf.RandomErasing(p=0.8, max_count=2, mode=mode, device="cpu")
        class Fresh2:
            def __len__(self):
                return n

            def __getitem__(self, i):
                return images[i].clone(), labels[i]
        md = ref_mix.MixDataset(Fresh2(), lambda img: erase((img - mean) / std), mixup=mixup, cutmix=cutmix)
        random.seed(seed)
        torch.manual_seed(1000 + seed)
        outs, l2, ratio = [], [], []
        for i in range(n):
            img, a, b, r = md[i]
            outs.append(img.numpy()); l2.append(b); ratio.append(float(r))
        rec[f"{tag}.images"] = np.stack(outs)
        rec[f"{tag}.label2"] = np.array(l2)
        rec[f"{tag}.ratio"] = np.array(ratio, dtype=np.float64)
    save("g9b_erase_modes", rec)


def gen_train_step():
    rec = {}
    B = 2
    x = fill((B, 3, 224, 224), 41, 1.0)
    l1 = torch.tensor([3, 977])
    l2 = torch.tensor([977, 3])
    ratio = torch.tensor([0.3, 0.85], dtype=torch.float32)
    model = load_formula(ref_swin.SwinTransformer(**SWIN_S, drop_path=0.0))
    model.train()
    crit = ref_loss.MixLoss(eps=0.1)
    skip = lambda n, p: ("bias" in n or "cls" in n or "norm" in n or p.ndim == 1)  # factory.py:33-34
    nodecay = [p for n, p in model.named_parameters() if skip(n, p)]
    decay = [p for n, p in model.named_parameters() if not skip(n, p)]
    opt = torch.optim.AdamW([{"params": nodecay, "weight_decay": 0.0},
                             {"params": decay, "weight_decay": 0.05}], lr=1e-3)
    out = model(x)
    loss = crit(out, l1, l2, ratio)
    loss.backward()
    total = torch.nn.utils.clip_grad_norm_(list(model.parameters()), 5.0)
    rec["loss"] = np.float64(loss.item())
    rec["total_norm"] = np.float64(total.item())
    rec["dlogits_check"] = summarize(out)
    opt.step()
    names, norms = [], []
    for n, p in model.named_parameters():
        names.append(n)
        norms.append(p.detach().double().norm().item())
    rec["param_names"] = np.array(names)
    rec["param_norms_after"] = np.array(norms, dtype=np.float64)
    rec["p.classifier.2.bias"] = summarize(model.classifier[2].bias)
    rec["p.block1.0.attn.rel_pos.weight"] = summarize(model.block1[0].attn.rel_pos.weight)
    rec["p.patch_embedding.linear.weight"] = summarize(model.patch_embedding.linear.weight)
    # MixLoss alone (value + grad) on formula logits
    lg = fill((4, 1000), 51, 3.0).requires_grad_(True)
    t1 = torch.tensor([1, 500, 999, 0]); t2 = torch.tensor([7, 500, 3, 998])
    r = torch.tensor([0.1, 0.5, 1.0, 0.0])
    lv = crit(lg, t1, t2, r)
    lv.backward()
    rec["mixloss.value"] = np.float64(lv.item())
    rec["mixloss.grad"] = summarize(lg.grad)
    save("g6_train_step", rec)


# ------------------------------------------------------------------ G10 (Twins-SVT: the row after F1-F4)
def gen_twins():
    rec = {}
    # positional-encoding generator (depthwise 3 x 3 + residual) on a 14 x 14 map
    peg = load_formula(ref_twins.PositionalEncodingGenerator(64)).double()
    x = fill((2, 14, 14, 64), 81, 1.0).double().requires_grad_(True)
    out = peg(x)
    (out * fill(out.shape, 82, 1.0).double()).sum().backward()
    rec["peg.out"] = summarize(out)
    rec["peg.dx"] = summarize(x.grad)
    rec["peg.grad.proj.weight"] = summarize(peg.proj.weight.grad)
    # locally-grouped attention: dim 64, 2 heads x 32, window 7 on 14 x 14
    lsa = load_formula(ref_twins.MultiHeadedLocalAttention(64, 2, 32, 7)).double()
    x = fill((2, 14, 14, 64), 83, 1.0).double().requires_grad_(True)
    out = lsa(x)
    (out * fill(out.shape, 84, 1.0).double()).sum().backward()
    rec["lsa.out"] = summarize(out)
    rec["lsa.dx"] = summarize(x.grad)
    for n, p in lsa.named_parameters():
        rec[f"lsa.grad.{n}"] = summarize(p.grad)
    # global sub-sampled attention: dim 128, 4 heads (head dim 32), reduction 7 on 28 x 28 (16 keys)
    gsa = load_formula(ref_twins.MultiHeadedAttention(128, 4, reduction=7)).double()
    x = fill((2, 28, 28, 128), 85, 1.0).double().requires_grad_(True)
    out = gsa(x)
    (out * fill(out.shape, 86, 1.0).double()).sum().backward()
    rec["gsa.out"] = summarize(out)
    rec["gsa.dx"] = summarize(x.grad)
    for n, p in gsa.named_parameters():
        rec[f"gsa.grad.{n}"] = summarize(p.grad)
    # patch embedding of a later stage: 2 x 2 tokens of a 28 x 28 x 64 map
    pe = load_formula(ref_twins.PatchEmbedding(64, 128, 2)).double()
    x = fill((2, 28, 28, 64), 87, 1.0).double().requires_grad_(True)
    out = pe(x)
    (out * fill(out.shape, 88, 1.0).double()).sum().backward()
    rec["patch_embed.out"] = summarize(out)
    rec["patch_embed.dx"] = summarize(x.grad)
    for n, p in pe.named_parameters():
        rec[f"patch_embed.grad.{n}"] = summarize(p.grad)
    # full model
    x = fill((2, 3, 224, 224), 21, 1.0)
    tw = load_formula(ref_twins.TwinsSVT(**TWINS_SVT_S, drop_path=0.0))
    rec["twins_svt_s.n_params"] = np.array(sum(p.numel() for p in tw.parameters()))
    rec["twins_svt_s.state_keys"] = np.array(list(tw.state_dict().keys()))
    rec["twins_svt_s.state_shapes"] = np.array([str(tuple(v.shape)) for v in tw.state_dict().values()])
    model_record(tw, x, "twins_svt_s.eval", rec, False)
    model_record(tw, x, "twins_svt_s.train", rec, True)
    tw64 = load_formula(ref_twins.TwinsSVT(**TWINS_SVT_S, drop_path=0.0)).double()
    model_record(tw64, x.double(), "twins_svt_s.train64", rec, True)
    for n, p in tw64.named_parameters():
        if any(s in n for s in ("block1.0.linear.weight", "block1.2.proj.weight", "block3.1.attn_global.reduce_conv.weight",
                                "block2.1.attn_local.weight.weight", "block4.3.attn_global.linear_kv.weight",
                                "block3.4.ff_global.0.weight", "classifier.2.weight")):
            rec[f"twins_svt_s.train64.grad.{n}"] = summarize(p.grad)
    # stochastic depth: the masks the reference draws (4 per layer with p > 0) and the logits they give
    torch.manual_seed(7)
    twd = load_formula(ref_twins.TwinsSVT(**TWINS_SVT_S, drop_path=0.3))
    twd.train()
    xb = fill((4, 3, 224, 224), 22, 1.0)
    masks, restore = capture_bernoulli()
    try:
        outd = twd(xb)
    finally:
        restore()
    rec["twins_svt_s.dp.logits"] = summarize(outd)
    rec["twins_svt_s.dp.masks"] = torch.stack(masks).numpy().astype(np.uint8)
    print("captured", len(masks), "drop-path masks")
    save("g10_twins", rec)


# ------------------------------------------------------------------ G11: attention-probability dropout
def inject_dropout(masks, seed):
    """Replace F.dropout (also what nn.Dropout.forward calls) by ``x * keep / (1 - p)`` with a keep mask drawn HERE from a seeded
    numpy generator and recorded: the reference runs its own code path, only the Bernoulli draw is ours."""
    import torch.nn.functional as F
    orig = F.dropout
    rng = np.random.default_rng(seed)

    def fake(input, p=0.5, training=True, inplace=False):
        if not training or p == 0:
            return input
        keep = torch.from_numpy((rng.random(tuple(input.shape)) >= p).astype(np.uint8))
        masks.append(keep)
        return input * keep.to(input.dtype) / (1.0 - p)

    F.dropout = fake
    return lambda: setattr(F, "dropout", orig)


def gen_attn_dropout():
    rec = {}
    P = 0.25

    def run(name, mod, x, call, seed):
        mod = load_formula(mod).double().train()
        masks = []
        restore = inject_dropout(masks, seed)
        try:
            xx = x.clone().requires_grad_(True)
            out = call(mod, xx)
            cot = fill(out.shape, name_seed(name + ".cot"), 1.0)
            (out * cot).sum().backward()
        finally:
            restore()
        assert len(masks) == 1, (name, len(masks))
        rec[f"{name}.keepshape"] = np.array(masks[0].shape, dtype=np.int64)
        rec[f"{name}.keep"] = np.packbits(masks[0].numpy().reshape(-1))
        rec[f"{name}.out"] = summarize(out)
        rec[f"{name}.dx"] = summarize(xx.grad)
        for n, g in grads_of(mod).items():
            rec[f"{name}.d.{n}"] = summarize(g)
        print(name, "keep fraction", float(masks[0].double().mean()))

    for L in (37, 197):
        run(f"vit_L{L}", ref_vit.MultiHeadedAttention(128, 2, dropout=P), fill((2, L, 128), 21, 1.0, dtype=torch.float64),
            lambda m, x: m(x), 100 + L)
    for shift, tag in ((True, "s1"), (False, "s0")):
        run(f"swin_{tag}", ref_swin.MultiHeadedLocalAttention(64, 2, 32, (14, 14), 7, shift, dropout=P),
            fill((2, 14, 14, 64), 22, 1.0, dtype=torch.float64), lambda m, x: m(x), 300 + int(shift))
    run("pvt_r2", ref_pvt.MultiHeadedAttention(128, 2, reduction=2, dropout=P), fill((2, 64, 128), 23, 1.0, dtype=torch.float64),
        lambda m, x: m(x, 8, 8)[0], 400)
    run("pvt_r1_cls", ref_pvt.MultiHeadedAttention(128, 2, reduction=1, dropout=P), fill((2, 17, 128), 24, 1.0, dtype=torch.float64),
        lambda m, x: m(x, 4, 4)[0], 401)
    run("twins_local", ref_twins.MultiHeadedLocalAttention(64, 2, 32, 7, dropout=P), fill((2, 14, 14, 64), 25, 1.0, dtype=torch.float64),
        lambda m, x: m(x), 500)
    run("twins_global", ref_twins.MultiHeadedAttention(64, 2, reduction=7, dropout=P), fill((2, 14, 14, 64), 26, 1.0, dtype=torch.float64),
        lambda m, x: m(x), 501)
    run("halo_w7a3", ref_halo.MultiHeadedHaloAttention(64, 2, 32, 7, 3, dropout=P), fill((2, 14, 14, 64), 27, 1.0, dtype=torch.float64),
        lambda m, x: m(x), 600)
    save("g11_attn_dropout", rec)


# ------------------------------------------------------------------ G12: halo attention (models/halo_transformer.py)
def gen_halo():
    rec = {}
    for tag, (dim, nh, dh, w, a, hw) in {"w7a3": (64, 2, 32, 7, 3, (14, 14)), "w4a1": (64, 2, 32, 4, 1, (8, 12)),
                                         "w8a3d64": (128, 2, 64, 8, 3, (16, 16))}.items():
        m = load_formula(ref_halo.MultiHeadedHaloAttention(dim, nh, dh, w, a)).double()
        rec[f"halo_{tag}.pos"] = m.pos.numpy().astype(np.int32)
        rec[f"halo_{tag}.ntab"] = np.array(m.rel_pos.num_embeddings)
        x = fill((2, hw[0], hw[1], dim), 91, 1.0, dtype=torch.float64).requires_grad_(True)
        out = m(x)
        (out * fill(out.shape, name_seed(f"halo_{tag}.cot"), 1.0).double()).sum().backward()
        rec[f"halo_{tag}.out"] = summarize(out)
        rec[f"halo_{tag}.dx"] = summarize(x.grad)
        for n, p in m.named_parameters():
            rec[f"halo_{tag}.d.{n}"] = summarize(p.grad)
    x = fill((2, 3, 224, 224), 21, 1.0)
    hm = load_formula(ref_halo.HaloTransformer(**HALO_TINY))
    rec["halo_tiny.n_params"] = np.array(sum(p.numel() for p in hm.parameters()))
    rec["halo_tiny.state_keys"] = np.array(list(hm.state_dict().keys()))
    rec["halo_tiny.state_shapes"] = np.array([str(tuple(v.shape)) for v in hm.state_dict().values()])
    rec["halo_tiny.param_names"] = np.array([n for n, _ in hm.named_parameters()])
    # Forward only: the reference's TransformerLayer adds IN PLACE (halo_transformer.py:150-151: ``input += ...``), which
    # invalidates what LayerNorm saved for its backward -- ``backward()`` of the reference model raises ("modified by an inplace
    # operation") on this torch.  Gradients are pinned per module above (the attention module has no such add).
    model_record(hm, x, "halo_tiny.eval", rec, False)
    with torch.no_grad():
        hm.train()
        rec["halo_tiny.train_fwd.logits"] = summarize(hm(x))            # (drop_path 0: equals eval; recorded as evidence)
    hm64 = load_formula(ref_halo.HaloTransformer(**HALO_TINY)).double().eval()
    with torch.no_grad():
        rec["halo_tiny.eval64.logits"] = summarize(hm64(x.double()))
    save("g12_halo", rec)


if __name__ == "__main__":
    which = sys.argv[1:] or ["tables", "modules", "models", "step", "pvt", "dino", "input", "twins", "attn_dropout", "halo"]
    if "halo" in which:
        gen_halo()
    if "attn_dropout" in which:
        gen_attn_dropout()
    if "tables" in which:
        gen_tables()
    if "modules" in which:
        gen_modules()
    if "models" in which:
        gen_models()
    if "step" in which:
        gen_train_step()
    if "pvt" in which:
        gen_pvt()
    if "dino" in which:
        gen_dino()
    if "input" in which:
        gen_input_pipeline()
    if "twins" in which:
        gen_twins()
    # make sure nothing was written into the reference tree
    assert not os.path.exists(os.path.join(REF, "models", "__pycache__")), "pycache leaked into reference"
