"""Twins-SVT on the MI355X-native kernels -- drop-in for the reference's models/twins.py (the row
after F1-F4): same class names, constructor signatures, forward contracts and state_dict keys / shapes.

  patchify                      reference models/twins.py:15-22
  PositionalEncodingGenerator   reference models/twins.py:25-37     (depthwise 3x3 convolution + residual)
  MultiHeadedAttention          reference models/twins.py:40-93     (global sub-sampled attention, GSA)
  MultiHeadedLocalAttention     reference models/twins.py:96-151    (locally-grouped attention, LSA)
  TransformerLayer              reference models/twins.py:154-203   (LSA block + GSA block, four residual branches)
  PatchEmbedding                reference models/twins.py:206-220
  TwinsSVT                      reference models/twins.py:227-356

Nothing here is new arithmetic: the local half of a layer is a Swin block without shift, bias table or mask (the window
attention kernels of csrc/attention_win.hip with a zero bias table), the global half is a PVT block whose reduction conv
(kernel = stride = window_size) is not followed by a LayerNorm (csrc/attention_sr.hip at head dim 32) and reads its operand
the way the reference reshapes its 4-D input (twins.py:69-70: a fixed permutation of the map, csrc/twins_misc.hip), and the positional
encoding generator is one channels-last depthwise-convolution kernel (csrc/twins_misc.hip).  Features stay NHWC
(B, H, W, C) like the reference.
"""
from typing import Tuple

import torch
from torch import nn

try:  # registration decorator of the reference's config system (identity when tensorfn is absent)
    from tensorfn.config import config_model
except Exception:  # pragma: no cover
    def config_model(*args, **kwargs):
        return lambda f: f

try:
    from pydantic import StrictFloat, StrictInt
except Exception:  # pragma: no cover
    StrictInt, StrictFloat = int, float

from vtx import functional as VF
from vtx import tables
from vtx.nn import LayerNorm as _LayerNorm
from vtx.nn import Linear, drop_path_scale, drop_path_scope, reset_transformer_parameters, stochastic_depth_rates

from .layer import DropPath, PositionwiseFeedForward
from .swin_transformer import patchify, reduce_size  # noqa: F401  (same helpers, re-exported under the reference's names)

LayerNorm = lambda x: _LayerNorm(x, eps=1e-6)


class PositionalEncodingGenerator(nn.Module):
    def __init__(self, dim):
        super().__init__()
        # parameter container with the reference's Conv2d layout / default init; runs as one channels-last kernel
        self.proj = nn.Conv2d(dim, dim, 3, padding=1, bias=False, groups=dim)

    def forward(self, input):
        T = VF.compute_dtype(input)
        return VF.PegFn.apply(input.to(T), self.proj.weight)


class MultiHeadedAttention(nn.Module):
    """Global sub-sampled attention: queries = all tokens, keys / values = the reduction x reduction sub-sampled map."""

    def __init__(self, dim, n_head, reduction=1, dropout=0):
        super().__init__()
        self.dim_head = dim // n_head
        self.n_head = n_head
        self.linear_q = Linear(dim, dim, bias=False)
        self.linear_kv = Linear(dim, dim * 2, bias=False)
        self.linear = Linear(dim, dim)
        self.dropout = dropout
        self.reduction = reduction
        if self.reduction > 1:
            self.reduce_conv = nn.Conv2d(dim, dim, self.reduction, stride=self.reduction)   # runs as gather + GEMM

    def check(self, height, width):
        r = self.reduction
        if self.dim_head not in (32, 64):
            raise NotImplementedError("vtx: the sub-sampled attention kernel is built for head dim 32 or 64")
        if r <= 1:
            # reference twins.py:74-77 would chunk its 4-D (B, H, W, 2 dim) projection along dim 2 (the width) in that case;
            # TransformerLayer never builds the module that way (reduction = window_size, twins.py:182)
            raise NotImplementedError("vtx: twins.MultiHeadedAttention needs reduction > 1 (as TransformerLayer builds it)")
        if height % r or width % r:
            raise ValueError(f"feature map {(height, width)} is not a multiple of the reduction {r}")
        # (more than 64 sub-sampled keys, e.g. 448 x 448: the key-block kernels behind vtx_srattn_*)

    def drops(self):
        """True when F.dropout(attn, self.dropout, self.training) of the reference (twins.py:88) is active."""
        return self.training and self.dropout > 0

    def forward(self, input, keep=None):
        """The reference's standalone contract (models/twins.py:56-93): input (B, H, W, dim) -> (B, H, W, dim).
        keep: an explicit uint8 keep mask [B * heads, H * W, Lk] for the attention dropout (parity tests); else hashed."""
        B, H, W, C = input.shape
        self.check(H, W)
        T = VF.compute_dtype(input)
        x = input.to(T).reshape(B, H * W, C)
        r = self.reduction
        q = VF.LinearFn.apply(x, self.linear_q.weight, None)
        # the reduction conv's operand exactly as twins.py:69-70 builds it from the 4-D input (a fixed permutation of the map)
        patches = VF.TwinsSubsampleFn.apply(x.view(B, H, W, C), r)
        w_rows = self.reduce_conv.weight.view(C, -1)               # (c', py, px) columns: the gather's column order
        kvin = VF.LinearFn.apply(patches, w_rows, self.reduce_conv.bias)
        Lk = (H // r) * (W // r)
        kv = VF.LinearFn.apply(kvin.reshape(B * Lk, C), self.linear_kv.weight, None)
        out = VF.SrAttentionFn.apply(q.reshape(B * H * W, C), kv, B, H * W, Lk, self.n_head,
                                     VF.attn_drop(self.dropout, self.training, keep))
        return VF.LinearFn.apply(out.view(B, H, W, C), self.linear.weight, self.linear.bias)


class MultiHeadedLocalAttention(nn.Module):
    """Locally-grouped attention: plain softmax attention inside non-overlapping window_size x window_size windows."""

    def __init__(self, dim, n_head, dim_head, window_size, dropout=0):
        super().__init__()
        self.dim_head = dim_head
        self.n_head = n_head
        self.weight = Linear(dim, n_head * dim_head * 3, bias=True)
        self.linear = Linear(n_head * dim_head, dim)
        self.window_size = window_size
        self.dropout = dropout
        self._meta = {}              # (H, W, device) -> AttentionMeta: the window kernels' tables, built on first use

    def meta(self, height, width, device, eps=1e-6):
        """Geometry + integer tables of the window-attention kernels for this feature-map size.  The reference module has no
        position bias and no mask: the kernels get an all-zero bias table (S + 0 = S exactly) and the un-shifted geometry."""
        key = (height, width, str(device))
        m = self._meta.get(key)
        if m is None:
            w = self.window_size
            if height % w or width % w:
                raise ValueError(f"feature map {(height, width)} is not a multiple of the window size {w}")
            pos, _ = tables.make_pos_mask((height, width), w, False)
            ntab = (2 * w - 1) ** 2
            order, offsets = tables.pos_csr(pos, ntab)
            zero_bias = torch.zeros(ntab, self.n_head, dtype=torch.float32, device=device)
            m = self._meta[key] = (VF.AttentionMeta(self.n_head, self.dim_head, w * w, eps=eps, swin=(height, width, w, False),
                                                    pos=pos.to(device), mask=None,
                                                    csr=(order.to(device), offsets.to(device)), ntab=ntab, region=None,
                                                    fast=True), zero_bias)
        if m[0].eps != eps:
            m[0].eps = eps
        return m

    def check(self):
        pass

    def drops(self):
        """True when F.dropout(attn, self.dropout, self.training) of the reference (twins.py:147) is active."""
        return self.training and self.dropout > 0

    def forward(self, input, keep=None):
        """keep: an explicit uint8 keep mask [B * windows * heads, L, L] for the attention dropout (parity tests); else hashed."""
        T = VF.compute_dtype(input)
        meta, zero_bias = self.meta(input.shape[1], input.shape[2], input.device)
        qkv = VF.LinearFn.apply(input.to(T), self.weight.weight, self.weight.bias)
        out = VF.AttentionCoreFn.apply(qkv, zero_bias, meta, VF.attn_drop(self.dropout, self.training, keep))
        return VF.LinearFn.apply(out, self.linear.weight, self.linear.bias)


class TransformerLayer(nn.Module):
    def __init__(self, dim, n_head, dim_head, dim_ff, window_size, activation=nn.SiLU, drop_ff=0, drop_attn=0, drop_path=0):
        super().__init__()
        self.norm_attn_local = LayerNorm(dim)
        self.attn_local = MultiHeadedLocalAttention(dim, n_head, dim_head, window_size, drop_attn)
        self.norm_ff_local = LayerNorm(dim)
        self.ff_local = PositionwiseFeedForward(dim, dim_ff, activation=activation, dropout=drop_ff)

        self.norm_attn_global = LayerNorm(dim)
        self.attn_global = MultiHeadedAttention(dim, n_head, window_size, drop_attn)
        self.norm_ff_global = LayerNorm(dim)
        self.ff_global = PositionwiseFeedForward(dim, dim_ff, activation=activation, dropout=drop_ff)

        self.drop_path = DropPath(drop_path)
        self.drop_path._vtx_draws = 4          # four residual branches share this module (vtx.nn.drop_path_scope)

    def set_drop_path(self, p):
        self.drop_path.p = p

    def forward(self, input):
        if not (self.ff_local.fused_ok() and self.ff_global.fused_ok()) or self.attn_local.drops() or self.attn_global.drops():
            out = input + self.drop_path(self.attn_local(self.norm_attn_local(input)))
            out = out + self.drop_path(self.ff_local(self.norm_ff_local(out)))
            out = out + self.drop_path(self.attn_global(self.norm_attn_global(out)))
            return out + self.drop_path(self.ff_global(self.norm_ff_global(out)))
        T = VF.compute_dtype(input)
        B, H, W, C = input.shape
        al, ag, fl, fg = self.attn_local, self.attn_global, self.ff_local, self.ff_global
        al.check()
        ag.check(H, W)
        p = self.drop_path.p
        # four independent draws per layer in the reference's order (twins.py:198-201)
        s = [drop_path_scale(p, self.training, B, input.device) for _ in range(4)]
        dp_c = (1.0 / (1.0 - p)) if s[0] is not None else 0.0
        meta, zero_bias = al.meta(H, W, input.device, self.norm_attn_local.eps)
        # local half: a Swin block without shift / bias / mask
        out = VF.TransformerLayerFn.apply(
            input.to(T), self.norm_attn_local.weight, self.norm_attn_local.bias, al.weight.weight, al.weight.bias, zero_bias,
            al.linear.weight, al.linear.bias, self.norm_ff_local.weight, self.norm_ff_local.bias, fl[0].weight, fl[0].bias,
            fl[3].weight, fl[3].bias, s[0], s[1], dp_c, meta)
        # global half: a PVT block whose reduction conv is not followed by a LayerNorm
        red = ag.reduction > 1
        out = VF.PvtLayerFn.apply(
            out.reshape(B, H * W, C), self.norm_attn_global.weight, self.norm_attn_global.bias, ag.linear_q.weight,
            ag.linear_kv.weight, ag.reduce_conv.weight if red else None, ag.reduce_conv.bias if red else None, None, None,
            ag.linear.weight, ag.linear.bias, self.norm_ff_global.weight, self.norm_ff_global.bias, fg[0].weight, fg[0].bias,
            fg[3].weight, fg[3].bias, s[2], s[3], dp_c,
            VF.PvtMeta(ag.n_head, H, W, ag.reduction, 0, self.norm_attn_global.eps, twins=True))
        return out.view(B, H, W, C)


class PatchEmbedding(nn.Module):
    """Input: NHWC features (B, H, W, in_dim) as in the reference (its caller permutes the NCHW image first)."""

    def __init__(self, in_dim, out_dim, window_size):
        super().__init__()
        self.window_size = window_size
        self.linear = Linear(in_dim * window_size * window_size, out_dim)
        self.norm = _LayerNorm(out_dim)

    def forward_nchw(self, input_nchw):
        """The image stage: NCHW -> NHWC permute + patchify folded into one gather (the Swin stem's kernels)."""
        T = VF.compute_dtype(input_nchw)
        return VF.SwinPatchEmbedFn.apply(input_nchw, self.linear.weight, self.linear.bias, self.norm.weight, self.norm.bias,
                                         self.window_size, self.norm.eps, T)

    def forward(self, input):
        B, H, W, C = input.shape
        p = self.window_size
        if C == 3:                                        # NHWC view of the NCHW image maps back for free
            return self.forward_nchw(input.permute(0, 3, 1, 2))
        T = VF.compute_dtype(input)
        rows = VF.PatchifyFn.apply(input.to(T).reshape(B, H * W, C), H, W, p, 0)     # (B H/p W/p, p p C), columns (py, px, c)
        out = VF.LinearFn.apply(rows, self.linear.weight, self.linear.bias)
        out = VF.LayerNormFn.apply(out, self.norm.weight, self.norm.bias, self.norm.eps)
        return out.view(B, H // p, W // p, -1)


@config_model(name="twins_svt", namespace="model", use_type=True)
class TwinsSVT(nn.Module):
    def __init__(
        self,
        n_class: StrictInt,
        depths: Tuple[StrictInt, StrictInt, StrictInt, StrictInt],
        dims: Tuple[StrictInt, StrictInt, StrictInt, StrictInt],
        dim_head: StrictInt,
        n_heads: Tuple[StrictInt, StrictInt, StrictInt, StrictInt],
        dim_ffs: Tuple[StrictInt, StrictInt, StrictInt, StrictInt],
        window_size: StrictInt,
        drop_ff: StrictFloat = 0.0,
        drop_attn: StrictFloat = 0.0,
        drop_path: StrictFloat = 0.0,
    ):
        super().__init__()
        self.depths = depths
        width = 3
        for k, step in enumerate((4, 2, 2, 2)):            # 4 x 4 pixels first, then 2 x 2 tokens in front of every stage
            setattr(self, f"block{k + 1}", self.make_block(depths[k], width, dims[k], n_heads[k], dim_head, dim_ffs[k],
                                                           window_size, step, drop_ff, drop_attn))
            width = dims[k]
        self.final_linear = nn.Sequential(_LayerNorm(dims[-1]))
        self.classifier = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(1), Linear(dims[-1], n_class))
        self.apply(self.init_weights)
        self.set_dropout(None, drop_path)

    init_weights = staticmethod(reset_transformer_parameters)

    def stages(self):
        return (self.block1, self.block2, self.block3, self.block4)

    def set_dropout(self, dropout, drop_path):
        """Linear stochastic-depth schedule over the transformer layers (patch embeddings and positional-encoding generators
        are skipped); ``dropout`` is accepted and ignored like in the reference (twins.py:275-311)."""
        layers = [m for stage in self.stages() for m in stage if hasattr(m, "set_drop_path")]
        for layer, rate in zip(layers, stochastic_depth_rates(drop_path, sum(self.depths), endpoint=False)):
            layer.set_drop_path(rate)

    def make_block(self, depth, in_dim, dim, n_head, dim_head, dim_ff, window_size, reduction, drop_ff, drop_attn):
        block = [PatchEmbedding(in_dim, dim, reduction)]
        for k in range(depth):
            block.append(TransformerLayer(dim, n_head, dim_head, dim_ff, window_size, drop_ff=drop_ff, drop_attn=drop_attn))
            if k == 0:                                    # conditional position encoding after the first layer of a stage
                block.append(PositionalEncodingGenerator(dim))
        return nn.Sequential(*block)

    def forward(self, input):
        with VF.weight_scope(self, input), drop_path_scope(self, input.shape[0], input.device):   # one cast, one mask draw
            out = self.block1[0].forward_nchw(input)         # permute(0, 2, 3, 1) + patchify folded into the gather
            for k, stage in enumerate(self.stages()):
                for j, module in enumerate(stage):
                    if k == 0 and j == 0:
                        continue
                    out = module(out)
            norm = self.final_linear[0]
            out = VF.LayerNormFn.apply(out, norm.weight, norm.bias, norm.eps)
            out = VF.TokenMeanFn.apply(out)                   # AdaptiveAvgPool2d(1) + Flatten(1) on NHWC
            cls = self.classifier[2]
            return VF.LinearFn.apply(out, cls.weight, cls.bias)
