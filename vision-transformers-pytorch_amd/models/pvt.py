"""Pyramid Vision Transformer on the MI355X-native kernels -- drop-in for the reference's models/pvt.py (SURVEY.md
section 8, row F1): same class names, constructor signatures, forward contracts and state_dict keys / shapes.

  MultiHeadedAttention       reference models/pvt.py:12-68   (spatial-reduction attention)
  TransformerLayer           reference models/pvt.py:71-103
  PatchEmbedding             reference models/pvt.py:106-140
  PyramidVisionTransformer   reference models/pvt.py:143-288

Features stay token-major (B, tokens, C) through the whole network: the reference's per-stage
``reshape(B, H, W, C).permute(0, 3, 1, 2)`` (pvt.py:269, 274, 279) only exists to feed nn.Conv2d; here the stride = kernel
convolutions (patch embeddings, spatial-reduction conv) are a token gather + GEMM on the same layout.
"""
import math

import torch
from torch import nn

from vtx import functional as VF
from vtx.nn import LayerNorm as _LayerNorm
from vtx.nn import Linear, drop_path_scale, drop_path_scope

from .layer import DropPath, PositionwiseFeedForward, tuple2

LayerNorm = lambda x: _LayerNorm(x, eps=1e-6)


class MultiHeadedAttention(nn.Module):
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
            # parameter container with the reference's Conv2d layout / default init; runs as gather + GEMM
            self.reduce_conv = nn.Conv2d(dim, dim, self.reduction, stride=self.reduction)
            self.reduce_norm = LayerNorm(dim)

    def check(self):
        if self.dim_head != 64:
            raise NotImplementedError("vtx: the PVT attention kernel is built for head dim 64 (all PVT configurations)")
        if self.training and self.dropout > 0:
            raise NotImplementedError("vtx: attention dropout > 0 is not supported by the fused HIP path")


class TransformerLayer(nn.Module):
    def __init__(self, dim, n_head, dim_ff, activation=nn.SiLU, reduction=1, drop_ff=0, drop_attn=0, drop_path=0):
        super().__init__()
        self.norm_attn = LayerNorm(dim)
        self.attn = MultiHeadedAttention(dim, n_head, reduction, drop_attn)
        self.drop_path = DropPath(drop_path)
        self.norm_ff = LayerNorm(dim)
        self.ff = PositionwiseFeedForward(dim, dim_ff, activation=activation, dropout=drop_ff)

    def set_drop_path(self, p):
        self.drop_path.p = p

    def forward(self, input, height, width):
        a, f = self.attn, self.ff
        a.check()
        if not f.fused_ok():
            raise NotImplementedError("vtx: PVT runs on the fused layer only (SiLU activation, dropout 0)")
        T = VF.compute_dtype(input)
        B, L = input.shape[0], input.shape[1]
        skip = L - height * width                                     # leading cls tokens (stage 4)
        if skip < 0 or (a.reduction > 1 and (height % a.reduction or width % a.reduction)):
            raise ValueError(f"token count {L} / grid {(height, width)} / reduction {a.reduction} do not fit")
        if a.reduction == 1 and L > 64 or a.reduction > 1 and (height // a.reduction) * (width // a.reduction) > 64:
            raise NotImplementedError("vtx: the PVT attention kernel holds at most 64 reduced key tokens")
        # two independent draws per layer, attention branch first (reference pvt.py:100-101)
        s1 = drop_path_scale(self.drop_path.p, self.training, B, input.device)
        s2 = drop_path_scale(self.drop_path.p, self.training, B, input.device)
        red = a.reduction > 1
        return VF.PvtLayerFn.apply(
            input.to(T), self.norm_attn.weight, self.norm_attn.bias, a.linear_q.weight, a.linear_kv.weight,
            a.reduce_conv.weight if red else None, a.reduce_conv.bias if red else None,
            a.reduce_norm.weight if red else None, a.reduce_norm.bias if red else None,
            a.linear.weight, a.linear.bias, self.norm_ff.weight, self.norm_ff.bias, f[0].weight, f[0].bias, f[3].weight,
            f[3].bias, s1, s2, (1.0 / (1.0 - self.drop_path.p)) if s1 is not None else 0.0,
            VF.PvtMeta(a.n_head, height, width, a.reduction, skip, self.norm_attn.eps))


class PatchEmbedding(nn.Module):
    def __init__(self, image_size, in_dim, dim, patch_size, cls_token=False, dropout=0):
        super().__init__()
        size = tuple2(patch_size)
        img_size = tuple2(image_size)
        if size[0] != size[1]:
            raise NotImplementedError("vtx: square patches only")
        self.conv = nn.Conv2d(in_dim, dim, size, stride=size)        # parameter container; runs as gather + GEMM
        self.norm = LayerNorm(dim)
        height, width = img_size[0] // size[0], img_size[1] // size[1]
        n_patch = height * width
        if cls_token:
            n_patch += 1
        self.pos = nn.Parameter(torch.randn(n_patch, dim) * 0.02)
        self.cls_token = None
        if cls_token:
            self.cls_token = nn.Parameter(torch.randn(dim) * 0.02)
        self.dim = dim
        self.patch = size[0]
        self.dropout = nn.Dropout(dropout)

    def forward(self, input, grid=None, skip=0):
        """input: the NCHW image / feature map as in the reference, or token-major features (B, skip + H*W, C) with
        ``grid = (H, W)`` (what PyramidVisionTransformer passes between stages).  Returns (tokens, (height, width))."""
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("vtx: dropout > 0 is not supported by the fused HIP path")
        T = VF.compute_dtype(input)
        if input.dim() == 4 and input.shape[1] != 3:                  # NCHW feature map: go through the token layout
            B, C, H, W = input.shape
            input, grid, skip = input.permute(0, 2, 3, 1).reshape(B, H * W, C).to(T), (H, W), 0
        if input.dim() == 4:
            height, width = input.shape[2] // self.patch, input.shape[3] // self.patch
        else:
            height, width = grid[0] // self.patch, grid[1] // self.patch
            input = input.to(T)
        out = VF.PvtPatchEmbedFn.apply(input, self.conv.weight, self.conv.bias, self.norm.weight, self.norm.bias,
                                       self.cls_token, self.pos, self.patch, grid, skip, self.norm.eps, T)
        return out, (height, width)


class PyramidVisionTransformer(nn.Module):
    def __init__(self, image_size, n_class, in_dim, depths, patch_embed_dims, n_heads, dim_ffs, reductions, drop_ff=0,
                 drop_attn=0, drop_path=0):
        super().__init__()
        self.depths = depths
        self.patch_embedding = nn.ModuleList()
        patch_embed_dims = list(patch_embed_dims)
        cls_token = False
        patch_sizes = (4, 2, 2, 2)
        img_size = tuple2(image_size)
        for i, (p_in, p_out, p_size) in enumerate(zip([in_dim] + patch_embed_dims[:-1], patch_embed_dims, patch_sizes)):
            if i == len(patch_embed_dims) - 1:
                cls_token = True
            self.patch_embedding.append(PatchEmbedding(img_size, p_in, p_out, p_size, cls_token=cls_token,
                                                       dropout=drop_ff))
            img_size = (img_size[0] // p_size, img_size[1] // p_size)

        def make_block(i):
            return self.make_block(depths[i], patch_embed_dims[i], n_heads[i], dim_ffs[i], reductions[i], drop_ff,
                                   drop_attn)

        self.block1 = make_block(0)
        self.block2 = make_block(1)
        self.block3 = make_block(2)
        self.block4 = make_block(3)
        self.norm = LayerNorm(patch_embed_dims[-1])
        self.classifier = Linear(patch_embed_dims[-1], n_class)
        self.apply(self.init_weights)
        self.set_drop_path(drop_path)

    def set_drop_path(self, drop_path):
        p = torch.linspace(0, drop_path, sum(self.depths)).tolist()
        i = 0
        for blocks in (self.block1, self.block2, self.block3, self.block4):
            for block in blocks:
                block.set_drop_path(p[i])
                i += 1

    def init_weights(self, module):
        if isinstance(module, nn.Linear):
            nn.init.normal_(module.weight, std=0.02)
            if module.bias is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, nn.LayerNorm):
            nn.init.ones_(module.weight)
            nn.init.zeros_(module.bias)

    def make_block(self, depth, dim, n_head, dim_ff, reduction, drop_ff, drop_attn):
        block = nn.ModuleList()
        for _ in range(depth):
            block.append(TransformerLayer(dim, n_head, dim_ff, reduction=reduction, drop_ff=drop_ff, drop_attn=drop_attn))
        return block

    def forward(self, input):
        with VF.weight_scope(self, input), drop_path_scope(self, input.shape[0], input.device):   # one cast, one mask draw
            out, (height, width) = self.patch_embedding[0](input)
            for block in self.block1:
                out = block(out, height, width)
            for embed, blocks in ((self.patch_embedding[1], self.block2), (self.patch_embedding[2], self.block3),
                                  (self.patch_embedding[3], self.block4)):
                out, (height, width) = embed(out, grid=(height, width))       # token-major in, no NCHW round trip
                for block in blocks:
                    out = block(out, height, width)
            out = self.norm(out[:, 0])
            return self.classifier(out)
