"""Vision Transformer (+ DINO head / factory) on the MI355X-native kernels -- drop-in for the reference's
models/vit.py: same class names, constructor signatures, forward contracts (single tensor or list of
multi-resolution crops) and state_dict keys / shapes / dtypes.

  MultiHeadedAttention   reference models/vit.py:16-45
  TransformerLayer       reference models/vit.py:48-66
  PatchEmbedding         reference models/vit.py:69-76
  VisionTransformer      reference models/vit.py:79-203
  DINOHead / dino        reference models/vit.py:206-307
"""
from typing import Tuple, Union

import torch
from torch import nn

try:
    from tensorfn.config import config_model
except Exception:  # pragma: no cover
    def config_model(*args, **kwargs):
        return lambda f: f

try:
    from pydantic import StrictBool, StrictFloat, StrictInt
except Exception:  # pragma: no cover
    StrictInt, StrictFloat, StrictBool = int, float, bool

from vtx import functional as VF
from vtx.nn import LayerNorm as _LayerNorm
from vtx.nn import (Linear, drop_path_scale, drop_path_scope, pair, projection_mlp, reset_transformer_parameters,
                    resize_position_grid, same_resolution_runs, stochastic_depth_rates)

from .layer import DropPath, PositionwiseFeedForward

LayerNorm = lambda x: _LayerNorm(x, eps=1e-6)


class MultiHeadedAttention(nn.Module):
    def __init__(self, dim, n_head, bias=True, dropout=0):
        super().__init__()
        self.dim_head = dim // n_head
        self.n_head = n_head
        self.qkv = Linear(dim, dim * 3, bias=bias)
        self.dropout = nn.Dropout(dropout)
        self.linear = Linear(dim, dim)

    def meta(self, length, eps=1e-6):
        return VF.AttentionMeta(self.n_head, self.dim_head, length, eps=eps)

    def drops(self):
        """True when F.dropout(attn, p) of the reference (vit.py:39) is active: the layer then runs call by call with the
        dropout variant of the attention kernels (keep mask regenerated in the backward from a counter-based hash)."""
        return self.training and self.dropout.p > 0

    def forward(self, input, keep=None):
        """keep: an explicit uint8 keep mask [B * heads, L, L] for the attention dropout (parity tests); else hashed."""
        T = VF.compute_dtype(input)
        qkv = VF.LinearFn.apply(input.to(T), self.qkv.weight, self.qkv.bias)
        out = VF.AttentionCoreFn.apply(qkv, None, self.meta(input.shape[1]), VF.attn_drop(self.dropout.p, self.training, keep))
        return VF.LinearFn.apply(out, self.linear.weight, self.linear.bias)


class TransformerLayer(nn.Module):
    def __init__(self, dim, n_head, dim_ff, dropout, drop_attn, drop_ff, drop_path):
        super().__init__()
        self.norm_attn = LayerNorm(dim)
        self.attn = MultiHeadedAttention(dim, n_head, dropout=drop_attn)
        self.norm_ff = LayerNorm(dim)
        self.ff = PositionwiseFeedForward(dim, dim_ff, dropout=drop_ff)
        self.dropout = nn.Dropout(dropout)
        self.drop_path = DropPath(drop_path)

    def forward(self, input):
        if self.attn.qkv.bias is None or not self.ff.fused_ok() or (self.training and self.dropout.p > 0) or self.attn.drops():
            # reference composition (vit.py:59-63) from the HIP modules: residual / feed-forward / attention dropout > 0, a
            # bias-free qkv or another activation -- none of the BASELINE configurations
            out = input + self.drop_path(self.dropout(self.attn(self.norm_attn(input))))
            return out + self.drop_path(self.dropout(self.ff(self.norm_ff(out))))
        T = VF.compute_dtype(input)
        B = input.shape[0]
        s1 = drop_path_scale(self.drop_path.p, self.training, B, input.device)   # reference vit.py:60
        s2 = drop_path_scale(self.drop_path.p, self.training, B, input.device)   # reference vit.py:61
        a, f = self.attn, self.ff
        return VF.TransformerLayerFn.apply(
            input.to(T), self.norm_attn.weight, self.norm_attn.bias, a.qkv.weight, a.qkv.bias, None,
            a.linear.weight, a.linear.bias, self.norm_ff.weight, self.norm_ff.bias, f[0].weight, f[0].bias,
            f[3].weight, f[3].bias, s1, s2,
            (1.0 / (1.0 - self.drop_path.p)) if s1 is not None else 0.0, a.meta(input.shape[1], self.norm_attn.eps))

    def set_drop_path(self, p):
        self.drop_path.p = p


class PatchEmbedding(nn.Module):
    def __init__(self, in_dim, out_dim, window_size):
        super().__init__()
        # parameter container with the reference's Conv2d layout / default init; the forward is an im2col GEMM
        self.linear = nn.Conv2d(in_dim, out_dim, window_size, stride=window_size)

    def forward(self, input):
        T = VF.compute_dtype(input)
        return VF.VitPatchEmbedFn.apply(input, self.linear.weight, self.linear.bias, T)


class VisionTransformer(nn.Module):
    def __init__(self, head, image_size, window_size, depth, dim, n_head, dim_ff, dropout, drop_attn, drop_ff,
                 drop_path):
        super().__init__()
        rows, cols = (side // window_size for side in pair(image_size))
        self.depth = depth
        self.patch_embedding = PatchEmbedding(3, dim, window_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, rows * cols + 1, dim))
        self.pos_drop = nn.Dropout(dropout)
        self.layers = nn.ModuleList(TransformerLayer(dim, n_head, dim_ff, dropout, drop_attn, drop_ff, rate)
                                    for rate in stochastic_depth_rates(drop_path, depth, endpoint=True))
        self.norm = LayerNorm(dim)
        self.apply(self.init_weights)
        for table in (self.pos_embed, self.cls_token):
            nn.init.normal_(table, std=0.02)
        self.head = head                                     # registered last (state_dict order), not re-initialised
        # Stochastic-depth compaction (host-drawn DropPath masks, each branch over its kept samples only: csrc/layer.hip) pays
        # where enough samples are dropped (measured: Swin-S at 0.3 -5.4 % per step, ViT-S/16 at its rates <= 0.1 +1.2 %): opt-in by rate
        self._vtx_dp_compaction = (dim % 128 == 0 and dim_ff % 128 == 0 and dim // n_head == 64 and drop_path >= 0.2)

    init_weights = staticmethod(reset_transformer_parameters)

    def set_drop_path(self, drop_path):
        for layer, rate in zip(self.layers, stochastic_depth_rates(drop_path, self.depth, endpoint=True)):
            layer.set_drop_path(rate)

    def forward_feature(self, input):
        tokens = self.patch_embedding(input)
        tokens = VF.VitAssembleFn.apply(tokens, self.cls_token, resize_position_grid(self.pos_embed, tokens.shape[1]))
        tokens = self.pos_drop(tokens)                       # (identity at the configured rate 0; reference vit.py:144)
        with drop_path_scope(self, tokens.shape[0], tokens.device):    # one mask draw per crop group
            for layer in self.layers:
                tokens = layer(tokens)
        # reference: norm(out)[:, 0]; LayerNorm is per token, so normalising only the cls rows is identical
        return self.norm(tokens[:, 0])

    def interpolate_pos_embedding(self, input, pos_embed):
        """Reference call style (vit.py:153-175): ``input`` is the (batch, 1 + n_patch, dim) token tensor."""
        return resize_position_grid(pos_embed, input.shape[1] - 1)

    def forward(self, input):
        crops = list(input) if isinstance(input, (list, tuple)) else [input]
        with VF.weight_scope(self, crops[0]):                # bf16: one multi-tensor cast of all weights per forward
            feats = [self.forward_feature(crops[a] if b - a == 1 else torch.cat(crops[a:b]))   # no copy of a lone batch
                     for a, b in same_resolution_runs(crops)]
            output = feats[0] if len(feats) == 1 else torch.cat(feats)
            return output if self.head is None else self.head(output)


class DINOHead(nn.Module):
    """Projection head of DINO (reference models/vit.py:206-262); Linear layers run on the HIP GEMM."""

    def __init__(self, in_dim, out_dim, use_bn=False, norm_last_layer=True, depth=3, dim_ff=2048,
                 dim_bottleneck=256):
        super().__init__()
        self.mlp = projection_mlp([in_dim] + [dim_ff] * (depth - 1) + [dim_bottleneck], use_bn)
        self.apply(self.init_weights)
        # weight-normalised output layer with unit gain, trainable only when norm_last_layer is off (vit.py:238-241)
        self.last = nn.utils.weight_norm(Linear(dim_bottleneck, out_dim, bias=False))
        self.last.weight_g.data.fill_(1)
        self.last.weight_g.requires_grad_(not norm_last_layer)

    init_weights = staticmethod(reset_transformer_parameters)     # no LayerNorm inside: Linear rule only

    def forward(self, input):
        linears = [self.mlp] if isinstance(self.mlp, nn.Linear) else list(self.mlp)
        T = VF.compute_dtype(input)
        if not all(isinstance(m, (nn.Linear, nn.GELU)) for m in linears):
            # use_bn=True puts BatchNorm1d between the projection layers (reference vit.py:226-229; the reference's own DINO
            # configuration runs without it): the HIP linears composed around torch's BatchNorm1d / GELU on the device tensors
            out = VF.L2NormFn.apply(self.mlp(input.to(T)), 1e-12)
            return self.last(out)
        # Linear / GELU chain with the activation in the GEMM epilogues, then the L2-normalisation kernel (no fallback:
        # CPU tensors raise inside the HIP ops)
        wb = [t for m in linears if isinstance(m, nn.Linear) for t in (m.weight, m.bias)]
        out = VF.MlpChainFn.apply(input.to(T), VF.ACT_GELU, *wb)
        out = VF.L2NormFn.apply(out, 1e-12)
        return self.last(out)


@config_model(name="dino", namespace="model", use_type=True)
def dino(
    image_size: Union[StrictInt, Tuple[StrictInt, StrictInt]],
    window_size: StrictInt,
    depth: StrictInt,
    dim: StrictInt,
    n_head: StrictInt,
    dim_ff: StrictInt,
    dropout: StrictFloat,
    drop_attn: StrictFloat,
    drop_ff: StrictFloat,
    drop_path: StrictFloat,
    dim_head_out: StrictInt,
    use_bn: StrictBool = False,
    norm_last_layer: StrictBool = True,
    depth_head: StrictInt = 3,
    dim_head_ff: StrictInt = 2048,
    dim_head_bottleneck: StrictInt = 256,
):
    head = DINOHead(dim, dim_head_out, use_bn, norm_last_layer, depth_head, dim_head_ff, dim_head_bottleneck)
    return VisionTransformer(head, image_size, window_size, depth, dim, n_head, dim_ff, dropout, drop_attn, drop_ff,
                             drop_path)
