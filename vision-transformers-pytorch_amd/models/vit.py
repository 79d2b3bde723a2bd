"""Vision Transformer (+ DINO head / factory) on the MI355X-native kernels -- drop-in for the reference's
models/vit.py: same class names, constructor signatures, forward contracts (single tensor or list of
multi-resolution crops) and state_dict keys / shapes / dtypes.

  MultiHeadedAttention   reference models/vit.py:16-45
  TransformerLayer       reference models/vit.py:48-66
  PatchEmbedding         reference models/vit.py:69-76
  VisionTransformer      reference models/vit.py:79-203
  DINOHead / dino        reference models/vit.py:206-307
"""
import math
from typing import Tuple, Union

import torch
from torch import nn
from torch.nn import functional as F

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
from vtx.nn import Linear, drop_path_scale, drop_path_scope

from .layer import DropPath, PositionwiseFeedForward, tuple2

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

    def check(self):
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("vtx: attention dropout > 0 is not supported by the fused HIP path")

    def forward(self, input):
        self.check()
        T = VF.compute_dtype(input)
        qkv = VF.LinearFn.apply(input.to(T), self.qkv.weight, self.qkv.bias)
        out = VF.AttentionCoreFn.apply(qkv, None, self.meta(input.shape[1]))
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
        self.attn.check()
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("vtx: residual dropout > 0 is not supported by the fused HIP path")
        if self.attn.qkv.bias is None or not self.ff.fused_ok():
            out = input + self.drop_path(self.attn(self.norm_attn(input)))
            return out + self.drop_path(self.ff(self.norm_ff(out)))
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
        image_size = tuple2(image_size)
        n_patch = (image_size[0] // window_size) * (image_size[1] // window_size)

        self.patch_embedding = PatchEmbedding(3, dim, window_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n_patch + 1, dim))
        self.pos_drop = nn.Dropout(dropout)

        drop_path_rate = torch.linspace(0, drop_path, depth).tolist()
        self.layers = nn.ModuleList(
            [TransformerLayer(dim, n_head, dim_ff, dropout, drop_attn, drop_ff, dpr) for dpr in drop_path_rate])
        self.norm = LayerNorm(dim)

        self.apply(self.init_weights)
        nn.init.normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=0.02)

        self.head = head
        self.depth = depth

    def set_drop_path(self, drop_path):
        drop_path_rate = torch.linspace(0, drop_path, self.depth).tolist()
        for layer, p in zip(self.layers, drop_path_rate):
            layer.set_drop_path(p)

    def init_weights(self, module):
        if isinstance(module, nn.Linear):
            nn.init.normal_(module.weight, std=0.02)
            if module.bias is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, nn.LayerNorm):
            nn.init.ones_(module.weight)
            nn.init.zeros_(module.bias)

    def forward_feature(self, input):
        if self.training and self.pos_drop.p > 0:
            raise NotImplementedError("vtx: positional dropout > 0 is not supported by the fused HIP path")
        out = self.patch_embedding(input)
        pos_embed = self.interpolate_pos_embedding(out.shape[1], out.shape[-1], self.pos_embed)
        out = VF.VitAssembleFn.apply(out, self.cls_token, pos_embed)
        with drop_path_scope(self, out.shape[0], out.device):    # one mask draw per crop group
            for layer in self.layers:
                out = layer(out)
        # reference: norm(out)[:, 0]; LayerNorm is per token, so normalising only the cls rows is identical
        return self.norm(out[:, 0])

    def interpolate_pos_embedding(self, n_patch, dim, pos_embed):
        """Bicubic resize of the patch position grid for non-default crop sizes (reference vit.py:153-175);
        host-side glue on fp32 parameters (a (1, n, dim) tensor), differentiable through torch."""
        if not isinstance(n_patch, int):   # reference call style: (input, pos_embed)
            n_patch, dim, pos_embed = n_patch.shape[1] - 1, n_patch.shape[-1], dim
        n_pos = pos_embed.shape[1] - 1
        if n_patch == n_pos:
            return pos_embed
        cls_embed = pos_embed[:, 0]
        grid = pos_embed[:, 1:]
        side = int(math.sqrt(n_pos))
        grid = F.interpolate(grid.reshape(1, side, side, dim).permute(0, 3, 1, 2),
                             scale_factor=math.sqrt(n_patch / n_pos), mode="bicubic", align_corners=False,
                             recompute_scale_factor=False)
        grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
        return torch.cat((cls_embed.unsqueeze(0), grid), 1)

    def forward(self, input):
        if not isinstance(input, (list, tuple)):
            input = [input]
        crops = torch.cumsum(
            torch.unique_consecutive(torch.tensor([i.shape[-1] for i in input]), return_counts=True)[1], 0)
        start = 0
        with VF.weight_scope(self, input[0]):                # bf16: one multi-tensor cast of all weights per forward
            for end in crops:
                out = self.forward_feature(torch.cat(input[start:end]))
                output = out if start == 0 else torch.cat((output, out))
                start = end
            if self.head is not None:
                output = self.head(output)
        return output


class DINOHead(nn.Module):
    """Projection head of DINO (reference models/vit.py:206-262); Linear layers run on the HIP GEMM."""

    def __init__(self, in_dim, out_dim, use_bn=False, norm_last_layer=True, depth=3, dim_ff=2048,
                 dim_bottleneck=256):
        super().__init__()
        if depth == 1:
            self.mlp = Linear(in_dim, dim_bottleneck)
        else:
            layers = [Linear(in_dim, dim_ff)]
            if use_bn:
                layers.append(nn.BatchNorm1d(dim_ff))
            layers.append(nn.GELU())
            for _ in range(depth - 2):
                layers.append(Linear(dim_ff, dim_ff))
                if use_bn:
                    layers.append(nn.BatchNorm1d(dim_ff))
                layers.append(nn.GELU())
            layers.append(Linear(dim_ff, dim_bottleneck))
            self.mlp = nn.Sequential(*layers)
        self.apply(self.init_weights)
        self.last = nn.utils.weight_norm(Linear(dim_bottleneck, out_dim, bias=False))
        self.last.weight_g.detach().fill_(1)
        if norm_last_layer:
            self.last.weight_g.requires_grad = False

    def init_weights(self, module):
        if isinstance(module, nn.Linear):
            nn.init.normal_(module.weight, std=0.02)
            if module.bias is not None:
                nn.init.zeros_(module.bias)

    def forward(self, input):
        linears = [self.mlp] if isinstance(self.mlp, nn.Linear) else list(self.mlp)
        if input.is_cuda and all(isinstance(m, (nn.Linear, nn.GELU)) for m in linears):
            # no BatchNorm: Linear / GELU chain with the activation in the GEMM epilogues, L2 normalisation kernel
            T = VF.compute_dtype(input)
            wb = [t for m in linears if isinstance(m, nn.Linear) for t in (m.weight, m.bias)]
            out = VF.MlpChainFn.apply(input.to(T), VF.ACT_GELU, *wb)
            out = VF.L2NormFn.apply(out, 1e-12)
        else:
            out = self.mlp(input)
            out = F.normalize(out.float(), dim=-1, p=2).to(out.dtype)
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
