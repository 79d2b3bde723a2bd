"""Whole-model CPU restatement: SwinTransformer.forward and VisionTransformer.forward.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional forwards driven by a
reference-compatible ``state_dict`` (same keys as SURVEY.md section 8(b)); the
backward comes from torch autograd over these functions.
"""
import math

import torch
import torch.nn.functional as F

from . import ref_ops as R

SWIN_S = dict(image_size=(224, 224), n_class=1000, depths=(2, 2, 18, 2),
              dims=(96, 192, 384, 768), dim_head=32, n_heads=(3, 6, 12, 24),
              dim_ffs=(384, 768, 1536, 3072), window_size=7)
VIT_S16 = dict(image_size=224, window_size=16, depth=12, dim=384, n_head=6, dim_ff=1536)


def swin_drop_path_rates(depths, drop_path):
    """dp_rate[i] = drop_path * i / n_blocks over transformer layers (swin:286-288)."""
    n = sum(depths)
    return [drop_path * float(i) / n for i in range(n)]


def swin_forward(sd, x_nchw, cfg, drop_masks=None, drop_path=0.0, q=None):
    """SwinTransformer.forward (swin:370-379).

    ``drop_masks``: optional list (one entry per transformer layer, network
    order) of (mask_attn, mask_ff) per-sample keep masks in {0,1}; rates follow
    ``swin_drop_path_rates``.  None = eval / drop_path 0.
    """
    depths, dims = cfg["depths"], cfg["dims"]
    w = cfg["window_size"]
    rates = swin_drop_path_rates(depths, drop_path)
    x = R.swin_patch_embedding(x_nchw, sd["patch_embedding.linear.weight"],
                               sd["patch_embedding.linear.bias"],
                               sd["patch_embedding.norm.weight"],
                               sd["patch_embedding.norm.bias"], q=q)
    li = 0
    for s in range(4):
        blk = f"block{s + 1}"
        j0 = 0
        if s > 0:
            x = R.patch_merge(x, sd[f"{blk}.0.norm.weight"], sd[f"{blk}.0.norm.bias"],
                              sd[f"{blk}.0.linear.weight"], q=q)
            j0 = 1
        for i in range(depths[s]):
            p = f"{blk}.{j0 + i}"
            shift = i % 2 == 0                                   # swin:362
            h = R._q(R.layer_norm(x, sd[f"{p}.norm_attn.weight"], sd[f"{p}.norm_attn.bias"], 1e-6), q)
            a = R.window_attention(h, sd[f"{p}.attn.weight.weight"], sd[f"{p}.attn.weight.bias"],
                                   sd[f"{p}.attn.linear.weight"], sd[f"{p}.attn.linear.bias"],
                                   sd[f"{p}.attn.rel_pos.weight"], cfg["n_heads"][s],
                                   cfg["dim_head"], w, shift, q=q)
            ma, mf = (drop_masks[li] if drop_masks is not None else (None, None))
            x = R._q(x + R.drop_path_apply(a, ma, rates[li]), q)  # swin:194
            h = R._q(R.layer_norm(x, sd[f"{p}.norm_ff.weight"], sd[f"{p}.norm_ff.bias"], 1e-6), q)
            f = R.feed_forward(h, sd[f"{p}.ff.0.weight"], sd[f"{p}.ff.0.bias"],
                               sd[f"{p}.ff.3.weight"], sd[f"{p}.ff.3.bias"], q=q)
            x = R._q(x + R.drop_path_apply(f, mf, rates[li]), q)  # swin:195
            li += 1
    x = R.layer_norm(x, sd["final_linear.0.weight"], sd["final_linear.0.bias"], 1e-5)
    pooled = R._q(x.mean(dim=(1, 2)), q)                        # AdaptiveAvgPool2d(1)+Flatten, swin:281
    return R.linear(pooled, sd["classifier.2.weight"], sd["classifier.2.bias"])


def vit_interpolate_pos(pos_embed, n_patch):
    """interpolate_pos_embedding (vit.py:153-175)."""
    n_pos = pos_embed.shape[1] - 1
    if n_patch == n_pos:
        return pos_embed
    dim = pos_embed.shape[-1]
    side = int(math.sqrt(n_pos))
    grid = pos_embed[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=math.sqrt(n_patch / n_pos), mode="bicubic",
                         align_corners=False, recompute_scale_factor=False)
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((pos_embed[:, :1], grid), 1)


def vit_forward_feature(sd, x_nchw, cfg, drop_masks=None, drop_path=0.0, q=None):
    """VisionTransformer.forward_feature (vit.py:139-151) -> (B, dim) cls feature."""
    depth, n_head = cfg["depth"], cfg["n_head"]
    rates = torch.linspace(0, drop_path, depth).tolist()         # vit.py:104
    t = R._q(R.vit_patch_embedding(x_nchw, sd["patch_embedding.linear.weight"],
                                   sd["patch_embedding.linear.bias"], cfg["window_size"]), q)
    B = t.shape[0]
    x = torch.cat((sd["cls_token"].expand(B, -1, -1), t), 1)
    x = R._q(x + vit_interpolate_pos(sd["pos_embed"], t.shape[1]), q)
    for i in range(depth):
        p = f"layers.{i}"
        h = R._q(R.layer_norm(x, sd[f"{p}.norm_attn.weight"], sd[f"{p}.norm_attn.bias"], 1e-6), q)
        a = R.global_attention(h, sd[f"{p}.attn.qkv.weight"], sd[f"{p}.attn.qkv.bias"],
                               sd[f"{p}.attn.linear.weight"], sd[f"{p}.attn.linear.bias"], n_head, q=q)
        ma, mf = (drop_masks[i] if drop_masks is not None else (None, None))
        x = R._q(x + R.drop_path_apply(a, ma, rates[i]), q)       # vit.py:60
        h = R._q(R.layer_norm(x, sd[f"{p}.norm_ff.weight"], sd[f"{p}.norm_ff.bias"], 1e-6), q)
        f = R.feed_forward(h, sd[f"{p}.ff.0.weight"], sd[f"{p}.ff.0.bias"],
                           sd[f"{p}.ff.3.weight"], sd[f"{p}.ff.3.bias"], q=q)
        x = R._q(x + R.drop_path_apply(f, mf, rates[i]), q)       # vit.py:61
    x = R.layer_norm(x[:, 0], sd["norm.weight"], sd["norm.bias"], 1e-6)   # norm then [:,0] == [:,0] then norm
    return x


def vit_forward(sd, inputs, cfg, head=None, **kw):
    """VisionTransformer.forward (vit.py:177-203): group consecutive same-size crops."""
    if not isinstance(inputs, (list, tuple)):
        inputs = [inputs]
    outs, start = [], 0
    sizes = [int(i.shape[-1]) for i in inputs]
    while start < len(inputs):
        end = start
        while end < len(inputs) and sizes[end] == sizes[start]:
            end += 1
        outs.append(vit_forward_feature(sd, torch.cat(inputs[start:end]), cfg, **kw))
        start = end
    out = torch.cat(outs)
    return head(out) if head is not None else out


# ------------------------------------------------------------------------------------------ PVT (models/pvt.py)
# PVT-Small hyper-parameters (not in the reference repo -- from the PVT paper)
PVT_SMALL = dict(image_size=224, n_class=1000, in_dim=3, depths=(3, 4, 6, 3), patch_embed_dims=(64, 128, 320, 512),
                 n_heads=(1, 2, 5, 8), dim_ffs=(512, 1024, 1280, 2048), reductions=(8, 4, 2, 1))
PVT_PATCH = (4, 2, 2, 2)                                                # pvt.py:171


def pvt_forward(sd, x_nchw, cfg, drop_masks=None, drop_path=0.0, q=None):
    """PyramidVisionTransformer.forward (pvt.py:262-286).  ``drop_masks`` as in swin_forward; the rates follow
    set_drop_path (pvt.py:209-231): linspace(0, drop_path, sum(depths)) in network order."""
    depths = cfg["depths"]
    rates = torch.linspace(0, drop_path, sum(depths)).tolist()
    out, li = x_nchw, 0
    B = x_nchw.shape[0]
    for s in range(4):
        pe = f"patch_embedding.{s}."
        if s > 0:                                                       # pvt.py:269, 274, 279: tokens -> NCHW
            out = out.reshape(B, height, width, -1).permute(0, 3, 1, 2)
        out, (height, width) = R.pvt_patch_embedding(out, sd[pe + "conv.weight"], sd[pe + "conv.bias"],
                                                     sd[pe + "norm.weight"], sd[pe + "norm.bias"], sd[pe + "pos"],
                                                     sd.get(pe + "cls_token"), PVT_PATCH[s])
        out = R._q(out, q)
        for j in range(depths[s]):
            pre = f"block{s + 1}.{j}."
            p = {k[len(pre + "attn."):]: v for k, v in sd.items() if k.startswith(pre + "attn.")}
            ma, mf = drop_masks[li] if drop_masks is not None else (None, None)
            a = R.pvt_attention(R._q(R.layer_norm(out, sd[pre + "norm_attn.weight"], sd[pre + "norm_attn.bias"], 1e-6), q),
                                height, width, p, cfg["n_heads"][s], cfg["reductions"][s], q)
            out = R._q(out + R.drop_path_apply(a, ma, rates[li]), q)
            f = R.feed_forward(R._q(R.layer_norm(out, sd[pre + "norm_ff.weight"], sd[pre + "norm_ff.bias"], 1e-6), q),
                               sd[pre + "ff.0.weight"], sd[pre + "ff.0.bias"], sd[pre + "ff.3.weight"],
                               sd[pre + "ff.3.bias"], q)
            out = R._q(out + R.drop_path_apply(f, mf, rates[li]), q)
            li += 1
    out = R.layer_norm(out[:, 0], sd["norm.weight"], sd["norm.bias"], 1e-6)   # pvt.py:283
    return R.linear(R._q(out, q), sd["classifier.weight"], sd["classifier.bias"])


# Twins-SVT-S of the paper (Chu et al. 2021, table 2) in the reference's formulation: one TransformerLayer = one locally-grouped
# + one globally sub-sampled block (twins.py:154-203), so depths (1, 1, 5, 2) are the paper's (2, 2, 10, 4) blocks.  The
# hyper-parameters are not in the reference repository (no .conf for the class).
TWINS_SVT_S = dict(n_class=1000, depths=(1, 1, 5, 2), dims=(64, 128, 256, 512), dim_head=32, n_heads=(2, 4, 8, 16),
                   dim_ffs=(256, 512, 1024, 2048), window_size=7)
TWINS_PATCH = (4, 2, 2, 2)                                              # twins.py:263-266


def twins_forward(sd, x_nchw, cfg, drop_masks=None, drop_path=0.0, q=None):
    """TwinsSVT.forward (twins.py:347-356).  ``drop_masks``: one entry per transformer layer (network order) of FOUR per-sample
    keep masks (local attention, local MLP, global attention, global MLP: twins.py:198-201); rates dp * i / n_layers
    (twins.py:275-277).  Block layout (twins.py:327-345): [PatchEmbedding, layer 0, PEG, layer 1, ...]."""
    depths, w = cfg["depths"], cfg["window_size"]
    rates = swin_drop_path_rates(depths, drop_path)
    x = x_nchw.permute(0, 2, 3, 1)
    li = 0
    for s in range(4):
        blk = f"block{s + 1}."
        x = R._q(R.twins_patch_embedding(x, sd[blk + "0.linear.weight"], sd[blk + "0.linear.bias"], sd[blk + "0.norm.weight"],
                                         sd[blk + "0.norm.bias"], TWINS_PATCH[s], q), q)
        j = 1
        for i in range(depths[s]):
            pre = f"{blk}{j}."
            sub = lambda name: {k[len(pre + name + "."):]: v for k, v in sd.items() if k.startswith(pre + name + ".")}
            ln = lambda name, t: R._q(R.layer_norm(t, sd[pre + name + ".weight"], sd[pre + name + ".bias"], 1e-6), q)
            ff = lambda name, t: R.feed_forward(t, sd[pre + name + ".0.weight"], sd[pre + name + ".0.bias"],
                                                sd[pre + name + ".3.weight"], sd[pre + name + ".3.bias"], q)
            m = drop_masks[li] if drop_masks is not None else (None,) * 4
            a = R.twins_local_attention(ln("norm_attn_local", x), sub("attn_local"), cfg["n_heads"][s], cfg["dim_head"], w, q)
            x = R._q(x + R.drop_path_apply(a, m[0], rates[li]), q)
            x = R._q(x + R.drop_path_apply(ff("ff_local", ln("norm_ff_local", x)), m[1], rates[li]), q)
            a = R.twins_global_attention(ln("norm_attn_global", x), sub("attn_global"), cfg["n_heads"][s], w, q)
            x = R._q(x + R.drop_path_apply(a, m[2], rates[li]), q)
            x = R._q(x + R.drop_path_apply(ff("ff_global", ln("norm_ff_global", x)), m[3], rates[li]), q)
            li += 1
            j += 1
            if i == 0:                                                  # twins.py:342-343
                x = R.twins_peg(x, sd[f"{blk}{j}.proj.weight"], q)
                j += 1
    x = R.layer_norm(x, sd["final_linear.0.weight"], sd["final_linear.0.bias"], 1e-5)
    pooled = R._q(x.mean(dim=(1, 2)), q)
    return R.linear(pooled, sd["classifier.2.weight"], sd["classifier.2.bias"])


# ------------------------------------------------------------------------------------------ Halo (models/halo_transformer.py)
HALO_TINY = dict(image_size=(224, 224), n_class=10, depths=(1, 1, 2, 1), dims=(64, 128, 256, 512), dim_head=32, n_heads=(2, 4, 8, 16),
                 dim_ffs=(128, 256, 512, 1024), window_size=7, halo_size=3)


def halo_forward(sd, x_nchw, cfg, drop_masks=None, drop_path=0.0, q=None):
    """HaloTransformer.forward (halo_transformer.py:272-280).  Block layout (262-270): [PatchEmbedding, layer, layer, ...] per stage,
    patch sizes 4, 2, 2, 2 (215-218); every layer has the SAME drop_path (no schedule); head: LayerNorm(eps 1e-5) -> Linear(C, 2C) ->
    LayerNorm -> SiLU per token, mean over the map, Linear (220-229).  ``drop_masks``: per layer (network order) two per-sample keep
    masks."""
    x = x_nchw.permute(0, 2, 3, 1)
    li = 0
    for k, patch in enumerate((4, 2, 2, 2)):
        pre = f"block{k + 1}."
        x = R.twins_patch_embedding(x, sd[pre + "0.linear.weight"], sd[pre + "0.linear.bias"], sd[pre + "0.norm.weight"],
                                    sd[pre + "0.norm.bias"], patch, q)
        for d in range(cfg["depths"][k]):
            lp = f"{pre}{d + 1}."
            p = {n[len(lp + "attn."):]: v for n, v in sd.items() if n.startswith(lp + "attn.")}
            a = R.halo_attention(R.layer_norm(x, sd[lp + "norm_attn.weight"], sd[lp + "norm_attn.bias"], 1e-6), p, cfg["n_heads"][k],
                                 cfg["dim_head"], cfg["window_size"], cfg["halo_size"], q)
            m = drop_masks[li] if drop_masks is not None else (None, None)
            x = x + R.drop_path_apply(a, m[0], drop_path)
            f = R.feed_forward(R.layer_norm(x, sd[lp + "norm_ff.weight"], sd[lp + "norm_ff.bias"], 1e-6), sd[lp + "ff.0.weight"],
                               sd[lp + "ff.0.bias"], sd[lp + "ff.3.weight"], sd[lp + "ff.3.bias"], q)
            x = x + R.drop_path_apply(f, m[1], drop_path)
            li += 1
    t = R.layer_norm(x, sd["final_linear.0.weight"], sd["final_linear.0.bias"], 1e-5)
    t = R.linear(t, sd["final_linear.1.weight"], sd["final_linear.1.bias"])
    t = R.silu(R.layer_norm(t, sd["final_linear.2.weight"], sd["final_linear.2.bias"], 1e-5))
    return R.linear(t.mean(dim=(1, 2)), sd["classifier.2.weight"], sd["classifier.2.bias"])
