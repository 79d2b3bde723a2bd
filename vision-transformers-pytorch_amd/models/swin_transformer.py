"""Swin Transformer on the MI355X-native kernels -- drop-in for the reference's
models/swin_transformer.py: same class names, constructor keyword signatures, forward tensor contracts
and state_dict keys / shapes / dtypes; every forward/backward op on the
feature tensors runs in hand-written gfx950 HIP kernels (libvtx.so), not in PyTorch op compositions.

  patchify                    reference models/swin_transformer.py:15-22
  MultiHeadedLocalAttention   reference models/swin_transformer.py:25-160
  TransformerLayer            reference models/swin_transformer.py:163-197
  PatchEmbedding              reference models/swin_transformer.py:200-213
  PatchMerge                  reference models/swin_transformer.py:216-229
  SwinTransformer             reference models/swin_transformer.py:236-379

Features are NHWC (B, H, W, C) like the reference; roll, window partition and head split are address
arithmetic inside the attention kernel.
"""
import math
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

LayerNorm = lambda x: _LayerNorm(x, eps=1e-6)


def patchify(input, size):
    """(B,H,W,C) -> (B,H/size,W/size,size*size*C), flatten order (py, px, c).  Pure view/copy helper kept for
    API parity; the model itself folds this gather into the patch-embed / PatchMerge kernels."""
    rows, cols = input.shape[1] // size, input.shape[2] // size
    tiles = input.unflatten(2, (cols, size)).unflatten(1, (rows, size))        # (B, rows, py, cols, px, C)
    return tiles.transpose(2, 3).flatten(3)                                     # (B, rows, cols, py * px * C)


class MultiHeadedLocalAttention(nn.Module):
    def __init__(self, dim, n_head, dim_head, input_size, window_size, shift, dropout=0):
        super().__init__()
        self.dim_head = dim_head
        self.n_head = n_head
        self.weight = Linear(dim, n_head * dim_head * 3, bias=True)
        self.linear = Linear(n_head * dim_head, dim)
        self.input_size = tuple(input_size)
        self.window_size = window_size
        self.dropout = dropout
        self.shift = shift

        pos, local_mask = tables.make_pos_mask(self.input_size, window_size, shift)
        self.register_buffer("pos", pos)
        self.rel_pos = nn.Embedding((2 * window_size - 1) ** 2, n_head)
        self.rel_pos.weight.detach().zero_()
        if shift:
            self.register_buffer("local_mask", local_mask)
        order, offsets = tables.pos_csr(pos, (2 * window_size - 1) ** 2)
        self.register_buffer("_csr_order", order, persistent=False)
        self.register_buffer("_csr_offsets", offsets, persistent=False)
        self._region = None          # (local_mask version, device, region ids, structured?) -- see regions()

    def regions(self):
        """Region ids of the CURRENT local_mask buffer for the window-attention kernels (tables.mask_regions),
        re-derived whenever the buffer is replaced or written (load_state_dict, .to())."""
        if not self.shift:
            return None, True
        m = self.local_mask
        key = (m._version, m.device, m.data_ptr())
        if self._region is None or self._region[0] != key:
            region, ok = tables.mask_regions(m)
            self._region = (key, region, ok)
        return self._region[1], self._region[2]

    def meta(self, eps=1e-6):
        w = self.window_size
        region, fast = self.regions()
        return VF.AttentionMeta(
            self.n_head, self.dim_head, w * w, eps=eps,
            swin=(self.input_size[0], self.input_size[1], w, self.shift), pos=self.pos,
            mask=self.local_mask if self.shift else None, csr=(self._csr_order, self._csr_offsets),
            ntab=(2 * w - 1) ** 2, region=region, fast=fast)

    def check_input(self, input):
        if tuple(input.shape[1:3]) != self.input_size:
            raise ValueError(f"feature map {tuple(input.shape[1:3])} != input_size {self.input_size} this layer's "
                             "pos / local_mask tables were built for")

    def drops(self):
        """True when F.dropout(attn, self.dropout, self.training) of the reference (swin_transformer.py:144) is active."""
        return self.training and self.dropout > 0

    def forward(self, input, keep=None):
        """keep: an explicit uint8 keep mask [B * windows * heads, L, L] for the attention dropout (parity tests); else hashed."""
        self.check_input(input)
        T = VF.compute_dtype(input)
        qkv = VF.LinearFn.apply(input.to(T), self.weight.weight, self.weight.bias)
        out = VF.AttentionCoreFn.apply(qkv, self.rel_pos.weight, self.meta(), VF.attn_drop(self.dropout, self.training, keep))
        return VF.LinearFn.apply(out, self.linear.weight, self.linear.bias)


class TransformerLayer(nn.Module):
    def __init__(self, dim, n_head, dim_head, dim_ff, input_size, window_size, shift, activation=nn.SiLU,
                 drop_ff=0, drop_attn=0, drop_path=0):
        super().__init__()
        self.norm_attn = LayerNorm(dim)
        self.attn = MultiHeadedLocalAttention(dim, n_head, dim_head, input_size, window_size, shift, drop_attn)
        self.drop_path = DropPath(drop_path)
        self.norm_ff = LayerNorm(dim)
        self.ff = PositionwiseFeedForward(dim, dim_ff, activation=activation, dropout=drop_ff)

    def set_drop_path(self, p):
        self.drop_path.p = p

    def forward(self, input):
        self.attn.check_input(input)
        if not self.ff.fused_ok() or self.attn.drops():
            # call by call (swin_transformer.py:193-197): another activation / feed-forward dropout / attention dropout
            out = input + self.drop_path(self.attn(self.norm_attn(input)))
            return out + self.drop_path(self.ff(self.norm_ff(out)))
        T = VF.compute_dtype(input)
        B = input.shape[0]
        # two independent draws per layer, attention branch first (reference swin_transformer.py:194-195)
        s1 = drop_path_scale(self.drop_path.p, self.training, B, input.device)
        s2 = drop_path_scale(self.drop_path.p, self.training, B, input.device)
        a, f = self.attn, self.ff
        return VF.TransformerLayerFn.apply(
            input.to(T), self.norm_attn.weight, self.norm_attn.bias, a.weight.weight, a.weight.bias,
            a.rel_pos.weight, a.linear.weight, a.linear.bias, self.norm_ff.weight, self.norm_ff.bias,
            f[0].weight, f[0].bias, f[3].weight, f[3].bias, s1, s2,
            (1.0 / (1.0 - self.drop_path.p)) if s1 is not None else 0.0, a.meta(self.norm_attn.eps))


class PatchEmbedding(nn.Module):
    """Input: NHWC image (B, H, W, 3) as in the reference (its caller permutes NCHW -> NHWC first)."""

    def __init__(self, in_dim, out_dim, window_size):
        super().__init__()
        self.window_size = window_size
        self.linear = Linear(in_dim * window_size * window_size, out_dim)
        self.norm = _LayerNorm(out_dim)

    def forward_nchw(self, input_nchw):
        T = VF.compute_dtype(input_nchw)
        return VF.SwinPatchEmbedFn.apply(input_nchw, self.linear.weight, self.linear.bias, self.norm.weight,
                                         self.norm.bias, self.window_size, self.norm.eps, T)

    def forward(self, input):
        # NHWC view of an NCHW tensor (what SwinTransformer.forward passes) maps back for free
        return self.forward_nchw(input.permute(0, 3, 1, 2))


class PatchMerge(nn.Module):
    def __init__(self, in_dim, out_dim, window_size):
        super().__init__()
        if window_size != 2:
            raise NotImplementedError("vtx: PatchMerge kernel supports the 2x2 reduction used by SwinTransformer")
        self.window_size = window_size
        self.norm = _LayerNorm(in_dim * window_size * window_size)
        self.linear = Linear(in_dim * window_size * window_size, out_dim, bias=False)

    def forward(self, input):
        T = VF.compute_dtype(input)
        return VF.PatchMergeFn.apply(input.to(T), self.norm.weight, self.norm.bias, self.linear.weight, self.norm.eps)


def reduce_size(size, reduction):
    return tuple(side // reduction for side in size[:2])


@config_model(name="swin_transformer", namespace="model", use_type=True)
class SwinTransformer(nn.Module):
    def __init__(
        self,
        image_size: Tuple[StrictInt, StrictInt],
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
        self.patch_embedding = PatchEmbedding(3, dims[0], 4)
        # stage k (block1..block4): 4x4 patches first, then a 2x2 PatchMerge in front of every later stage
        size, width = reduce_size(image_size, 4), 3
        for k in range(4):
            merge = 1 if k == 0 else 2
            setattr(self, f"block{k + 1}", self.make_block(depths[k], width, dims[k], n_heads[k], dim_head, dim_ffs[k],
                                                           size, window_size, merge, drop_ff, drop_attn))
            size, width = reduce_size(size, merge), dims[k]
        self.final_linear = nn.Sequential(_LayerNorm(dims[-1]))
        self.classifier = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(1), Linear(dims[-1], n_class))
        self.apply(self.init_weights)
        self.set_dropout(None, drop_path)
        # vtx.nn.drop_path_scope draws this model's DropPath masks on the host, so that the layers whose GEMMs allow it
        # (dim and dim_ff multiples of 128: stages 3-4 of Swin-S) compute each branch for its kept samples only
        self._vtx_dp_compaction = any(d % 128 == 0 and f % 128 == 0 for d, f in zip(dims, dim_ffs))

    init_weights = staticmethod(reset_transformer_parameters)

    def stages(self):
        return (self.block1, self.block2, self.block3, self.block4)

    def set_dropout(self, dropout, drop_path):
        """Linear stochastic-depth schedule over all transformer layers (PatchMerge modules are skipped); ``dropout`` is
        accepted and ignored like in the reference (swin_transformer.py:307-319)."""
        layers = [m for stage in self.stages() for m in stage if hasattr(m, "set_drop_path")]
        for layer, rate in zip(layers, stochastic_depth_rates(drop_path, sum(self.depths), endpoint=False)):
            layer.set_drop_path(rate)

    def make_block(self, depth, in_dim, dim, n_head, dim_head, dim_ff, input_size, window_size, reduction, drop_ff,
                   drop_attn):
        grid = reduce_size(input_size, reduction)
        head = [PatchMerge(in_dim, dim, reduction)] if reduction > 1 else []
        return nn.Sequential(*head, *(TransformerLayer(dim, n_head, dim_head, dim_ff, grid, window_size,
                                                       shift=k % 2 == 0, drop_ff=drop_ff, drop_attn=drop_attn)
                                      for k in range(depth)))

    def forward(self, input):
        with VF.weight_scope(self, input), drop_path_scope(self, input.shape[0], input.device):   # one cast, one mask draw
            out = self.patch_embedding.forward_nchw(input)   # permute(0,2,3,1) + patchify folded into the gather
            for stage in self.stages():
                out = stage(out)
            norm = self.final_linear[0]
            out = VF.LayerNormFn.apply(out, norm.weight, norm.bias, norm.eps)
            out = VF.TokenMeanFn.apply(out)                   # AdaptiveAvgPool2d(1) + Flatten(1) on NHWC
            cls = self.classifier[2]
            return VF.LinearFn.apply(out, cls.weight, cls.bias)
