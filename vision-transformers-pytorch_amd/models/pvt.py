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
from vtx.nn import Linear, drop_path_scale, drop_path_scope, pair, reset_transformer_parameters, stochastic_depth_rates

from .layer import DropPath, PositionwiseFeedForward

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

    def drops(self):
        """True when F.dropout(attn, self.dropout, self.training) of the reference (pvt.py:60) is active."""
        return self.training and self.dropout > 0

    def forward(self, input, height, width, prev=None, keep=None):
        """The reference's standalone contract (models/pvt.py:31-69): input (B, L, dim) with the last height * width tokens
        on the grid (leading tokens -- a cls token -- attend but are not reduced); returns ``(out, score)`` with score =
        q k^T / sqrt(d) of shape (B, heads, L, Lk) before the softmax.  The PVT layers do not come through here (they run
        the fused PvtLayerFn and discard the score like the reference's TransformerLayer, pvt.py:100); every op is a HIP
        kernel.  ``prev`` (never passed inside the reference) is not supported; no gradient flows through ``score``."""
        self.check()
        if prev is not None:
            raise NotImplementedError("vtx: pvt.MultiHeadedAttention.forward(prev=...) is not supported")
        T = VF.compute_dtype(input)
        x = input.to(T)
        B, L, C = x.shape
        r, skip = self.reduction, L - height * width
        if skip < 0 or (r > 1 and (height % r or width % r)):
            raise ValueError("pvt.MultiHeadedAttention: tokens do not match the (height, width) grid / reduction")
        q = VF.LinearFn.apply(x, self.linear_q.weight, None)
        if r > 1:
            patches = VF.PatchifyFn.apply(x, height, width, r, skip)
            w_rows = self.reduce_conv.weight.permute(0, 2, 3, 1).reshape(C, -1)       # conv weight as (py, px, c) columns
            red = VF.LinearFn.apply(patches, w_rows, self.reduce_conv.bias)
            kvin = self.reduce_norm(red)
            Lk = (height // r) * (width // r)
        else:
            kvin, Lk = x, L
        kv = VF.LinearFn.apply(kvin.reshape(B * Lk, C), self.linear_kv.weight, None)
        q2 = q.reshape(B * L, C)
        out = VF.SrAttentionFn.apply(q2, kv, B, L, Lk, self.n_head, VF.attn_drop(self.dropout, self.training, keep))
        out = VF.LinearFn.apply(out.view(B, L, C), self.linear.weight, self.linear.bias)
        from vtx import ops
        with torch.no_grad():
            score = ops.srattn_scores(q2.detach().contiguous(), kv.detach().contiguous(), B, L, Lk, self.n_head)
        return out, score

    def check(self):
        if self.dim_head != 64:
            raise NotImplementedError("vtx: the PVT attention kernel is built for head dim 64 (all PVT configurations)")


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
        if a.drops():
            # attention dropout: call by call (pvt.py:99-103) through the standalone module (its score output is discarded)
            out = input + self.drop_path(a(self.norm_attn(input), height, width)[0])
            return out + self.drop_path(f(self.norm_ff(out)))
        if not f.fused_ok():
            raise NotImplementedError("vtx: PVT runs on the fused layer only (SiLU activation, dropout 0)")
        T = VF.compute_dtype(input)
        B, L = input.shape[0], input.shape[1]
        skip = L - height * width                                     # leading cls tokens (stage 4)
        if skip < 0 or (a.reduction > 1 and (height % a.reduction or width % a.reduction)):
            raise ValueError(f"token count {L} / grid {(height, width)} / reduction {a.reduction} do not fit")
        # (more than 64 reduced keys -- 256 x 256 stage 4, 384 x 384 everywhere -- run on the key-block kernels: vtx_srattn_*)
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
        (ph, pw), (ih, iw) = pair(patch_size), pair(image_size)
        if ph != pw:
            raise NotImplementedError("vtx: square patches only")
        self.dim, self.patch = dim, ph
        self.conv = nn.Conv2d(in_dim, dim, (ph, pw), stride=(ph, pw))   # parameter container; runs as gather + GEMM
        self.norm = LayerNorm(dim)
        # learned positions (N(0, 0.02)) for every patch, plus one row and a class vector in the last stage
        self.pos = nn.Parameter(0.02 * torch.randn((ih // ph) * (iw // pw) + int(bool(cls_token)), dim))
        self.cls_token = nn.Parameter(0.02 * torch.randn(dim)) if cls_token else None
        self.dropout = nn.Dropout(dropout)

    def forward(self, input, grid=None, skip=0):
        """input: the NCHW image / feature map as in the reference, or token-major features (B, skip + H*W, C) with
        ``grid = (H, W)`` (what PyramidVisionTransformer passes between stages).  Returns (tokens, (height, width))."""
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
        return self.dropout(out), (height, width)         # (identity at the configured rate 0; reference pvt.py:138)


class PyramidVisionTransformer(nn.Module):
    def __init__(self, image_size, n_class, in_dim, depths, patch_embed_dims, n_heads, dim_ffs, reductions, drop_ff=0,
                 drop_attn=0, drop_path=0):
        super().__init__()
        self.depths = depths
        widths = list(patch_embed_dims)
        last = len(widths) - 1
        # stage k: patches of 4 (then 2, 2, 2) pixels of the previous map; the class token joins in the last stage
        self.patch_embedding = nn.ModuleList()
        grid = pair(image_size)
        for k, (step, c_in, c_out) in enumerate(zip((4, 2, 2, 2), [in_dim] + widths[:-1], widths)):
            self.patch_embedding.append(PatchEmbedding(grid, c_in, c_out, step, cls_token=k == last, dropout=drop_ff))
            grid = (grid[0] // step, grid[1] // step)
        for k in range(4):
            setattr(self, f"block{k + 1}",
                    self.make_block(depths[k], widths[k], n_heads[k], dim_ffs[k], reductions[k], drop_ff, drop_attn))
        self.norm = LayerNorm(widths[-1])
        self.classifier = Linear(widths[-1], n_class)
        self.apply(self.init_weights)
        self.set_drop_path(drop_path)

    init_weights = staticmethod(reset_transformer_parameters)

    def stages(self):
        return (self.block1, self.block2, self.block3, self.block4)

    def set_drop_path(self, drop_path):
        layers = [layer for stage in self.stages() for layer in stage]
        for layer, rate in zip(layers, stochastic_depth_rates(drop_path, sum(self.depths), endpoint=True)):
            layer.set_drop_path(rate)

    def make_block(self, depth, dim, n_head, dim_ff, reduction, drop_ff, drop_attn):
        return nn.ModuleList(TransformerLayer(dim, n_head, dim_ff, reduction=reduction, drop_ff=drop_ff,
                                              drop_attn=drop_attn) for _ in range(depth))

    def forward(self, input):
        with VF.weight_scope(self, input), drop_path_scope(self, input.shape[0], input.device):   # one cast, one mask draw
            out, grid = input, None
            for embed, stage in zip(self.patch_embedding, self.stages()):
                out, grid = embed(out) if grid is None else embed(out, grid=grid)   # token-major between stages
                for layer in stage:
                    out = layer(out, *grid)
            out = self.norm(out[:, 0])
            return self.classifier(out)
