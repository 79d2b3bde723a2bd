"""MI355X-native drop-in for the reference's models/halo_transformer.py (blocked local attention with halos): same class names,
constructor arguments, state_dict keys and forward contracts; every op runs as a HIP kernel of libvtx.so.

  patchify                     reference models/halo_transformer.py:12-19
  MultiHeadedHaloAttention     reference models/halo_transformer.py:22-115 (pos table 41-57; forward 58-115)
  TransformerLayer             reference models/halo_transformer.py:118-154
  PatchEmbedding               reference models/halo_transformer.py:157-170
  HaloTransformer              reference models/halo_transformer.py:177-280

SURVEY.md section 8 ranks this family last (row F4's last sentence); it is built from the window gather / scatter kernels of
csrc/halo.hip and the key-block cross-attention kernels of csrc/attention_long.hip (head dim 32 or 64), the layers run call by call.
"""
from typing import Tuple

import torch
from torch import nn

from vtx import functional as VF
from vtx import ops, tables
from vtx.nn import LayerNorm as _LayerNorm
from vtx.nn import Linear, drop_path_scope, reset_transformer_parameters

from .layer import DropPath, PositionwiseFeedForward
from .swin_transformer import patchify, reduce_size  # noqa: F401  (same helpers, the reference's names)
from .twins import PatchEmbedding  # noqa: F401  (patchify -> Linear -> LayerNorm(eps 1e-5): the same module in both reference files)

LayerNorm = lambda x: _LayerNorm(x, eps=1e-6)


class MultiHeadedHaloAttention(nn.Module):
    def __init__(self, dim, n_head, dim_head, window_size, halo_size, dropout=0):
        super().__init__()
        self.dim_head = dim_head
        self.n_head = n_head
        self.weight = Linear(dim, n_head * dim_head * 3, bias=False)
        self.linear = Linear(n_head * dim_head, dim)
        self.window_size = window_size
        self.halo_size = halo_size
        self.dropout = dropout
        if halo_size < 1:
            raise ValueError("halo_size must be >= 1 (the reference's index table is empty for 0: halo_transformer.py:44)")
        pos, n_table = tables.make_halo_pos(window_size, halo_size)
        self.register_buffer("pos", pos)
        self.rel_pos = nn.Embedding(n_table, n_head)
        self.rel_pos.weight.detach().zero_()
        order, offsets = ops.pos_csr(pos, n_table)
        self.register_buffer("_csr_order", order, persistent=False)
        self.register_buffer("_csr_offsets", offsets, persistent=False)

    def meta(self):
        return VF.HaloMeta(self.n_head, self.dim_head, self.window_size, self.halo_size, self.pos, (self._csr_order, self._csr_offsets),
                           self.rel_pos.num_embeddings)

    def forward(self, input, keep=None):
        """keep: an explicit uint8 keep mask [B * windows * heads, window^2, (window + 2 halo)^2] for the attention dropout (the
        reference's attention tensor is (B, heads, windows, ...): transpose it), else the hash mask."""
        B, H, W, _ = input.shape
        w = self.window_size
        if H % w or W % w:
            raise ValueError(f"feature map {(H, W)} is not a multiple of the window size {w}")
        if self.dim_head not in (32, 64):
            raise NotImplementedError("vtx: the halo attention kernels are built for head dim 32 or 64")
        T = VF.compute_dtype(input)
        qkv = VF.LinearFn.apply(input.to(T), self.weight.weight, None)
        out = VF.HaloAttentionFn.apply(qkv, self.rel_pos.weight, self.meta(), VF.attn_drop(self.dropout, self.training, keep))
        return VF.LinearFn.apply(out, self.linear.weight, self.linear.bias)


class TransformerLayer(nn.Module):
    def __init__(self, dim, n_head, dim_head, dim_ff, window_size, halo_size, activation=nn.SiLU, drop_ff=0, drop_attn=0,
                 drop_path=0):
        super().__init__()
        self.norm_attn = LayerNorm(dim)
        self.attn = MultiHeadedHaloAttention(dim, n_head, dim_head, window_size, halo_size, drop_attn)
        self.drop_path = DropPath(drop_path)
        self.norm_ff = LayerNorm(dim)
        self.ff = PositionwiseFeedForward(dim, dim_ff, activation=activation, dropout=drop_ff)

    def set_drop_path(self, p):
        self.drop_path.p = p

    def forward(self, input):
        # (the reference adds in place, halo_transformer.py:150-151: same values)
        out = input + self.drop_path(self.attn(self.norm_attn(input)))
        return out + self.drop_path(self.ff(self.norm_ff(out)))


class HaloTransformer(nn.Module):
    def __init__(self, image_size, n_class, depths, dims, dim_head, n_heads, dim_ffs, window_size, halo_size, drop_ff=0, drop_attn=0,
                 drop_path=0):
        super().__init__()
        self.depths = depths
        width = 3
        for k, step in enumerate((4, 2, 2, 2)):
            setattr(self, f"block{k + 1}", self.make_block(depths[k], width, dims[k], n_heads[k], dim_head, dim_ffs[k], window_size,
                                                           halo_size, step, drop_ff, drop_attn, drop_path))
            width = dims[k]
        # LayerNorm -> Linear -> LayerNorm -> SiLU on every token, then the mean over the map and the classifier
        self.final_linear = nn.Sequential(_LayerNorm(dims[-1]), Linear(dims[-1], dims[-1] * 2), _LayerNorm(dims[-1] * 2), nn.SiLU(inplace=True))
        linear = Linear(dims[-1] * 2, n_class)
        self.classifier = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(1), linear)
        self.apply(self.init_weights)
        nn.init.normal_(linear.weight, std=0.01)       # (the reference's apply() re-draws it with std 0.02 afterwards as well:
        nn.init.normal_(linear.weight, std=0.02)       #  halo_transformer.py:221-226 -- the last draw is what the model starts from)
        nn.init.zeros_(linear.bias)

    init_weights = staticmethod(reset_transformer_parameters)

    def make_block(self, depth, in_dim, dim, n_head, dim_head, dim_ff, window_size, halo_size, reduction, drop_ff, drop_attn, drop_path):
        block = [PatchEmbedding(in_dim, dim, reduction)]
        for _ in range(depth):
            # (every layer gets the same drop_path: the reference has no schedule here, halo_transformer.py:262-276)
            block.append(TransformerLayer(dim, n_head, dim_head, dim_ff, window_size, halo_size, drop_ff=drop_ff, drop_attn=drop_attn,
                                          drop_path=drop_path))
        return nn.Sequential(*block)

    def forward(self, input):
        with VF.weight_scope(self, input), drop_path_scope(self, input.shape[0], input.device):
            out = self.block1[0].forward_nchw(input)             # permute(0, 2, 3, 1) + patchify folded into the gather
            for j, module in enumerate(self.block1):
                if j:
                    out = module(out)
            for stage in (self.block2, self.block3, self.block4):
                out = stage(out)
            B, H, W, C = out.shape
            T = VF.compute_dtype(out)
            fl = self.final_linear
            t = VF.LayerNormFn.apply(out.to(T).reshape(B * H * W, C), fl[0].weight, fl[0].bias, fl[0].eps)
            t = VF.LinearFn.apply(t, fl[1].weight, fl[1].bias)
            t = VF.LayerNormFn.apply(t, fl[2].weight, fl[2].bias, fl[2].eps)
            t = torch.nn.functional.silu(t)
            pooled = VF.TokenMeanFn.apply(t.view(B, H * W, 2 * C))
            return VF.LinearFn.apply(pooled, self.classifier[2].weight, self.classifier[2].bias)
