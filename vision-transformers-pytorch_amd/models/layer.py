"""Shared blocks of the transformer families -- MI355X-native counterparts of the hot-path symbols of
the reference's models/layer.py (same names, constructor arguments, state_dict keys):

  tuple2 / ensure_tuple      reference models/layer.py:9-25
  DropPath                   reference models/layer.py:166-183
  PositionwiseFeedForward    reference models/layer.py:186-196

The CNN-only helpers of the reference file (ScaledActivation, WSConv2d, StochasticDepth,
SqueezeExcite, GlobalContext) are outside the ViT / Swin hot path and are not provided.
"""
import torch
from torch import nn

from vtx import functional as VF
from vtx.nn import Linear, _DropPathBase, pair


def ensure_tuple(x, n_item):
    """``x`` repeated ``n_item`` times, or ``x`` itself when it already is a sequence of that length."""
    if isinstance(x, (str, bytes)) or not hasattr(x, "__iter__"):
        return (x,) * n_item
    if hasattr(x, "__len__") and len(x) != n_item:
        raise ValueError(f"length of {x} (length: {len(x)}) does not match the expected length {n_item}")
    return x


tuple2 = pair


class DropPath(_DropPathBase):
    """Per-sample stochastic depth: identity in eval or p == 0, else x / (1-p) * Bernoulli(1-p) mask.

    Inside the fused TransformerLayer the mask is folded into the GEMM epilogue as a per-sample
    scale; this standalone forward exists for API parity.
    """

    def __init__(self, p=0):
        super().__init__()
        self.p = p

    def forward(self, input):
        if not self.training or self.p == 0:
            return input
        keep = 1 - self.p
        mask = input.new_empty([input.shape[0]] + [1] * (input.ndim - 1)).bernoulli_(keep)
        return input / keep * mask

    def __repr__(self):
        return f"{self.__class__.__name__}(p={self.p})"


class PositionwiseFeedForward(nn.Sequential):
    """Linear -> activation -> Dropout -> Linear with the reference's Sequential layout (keys 0.*, 3.*)."""

    def __init__(self, in_dim, dim=None, out_dim=None, activation=nn.SiLU, dropout=0):
        dim = in_dim if dim is None else dim
        out_dim = in_dim if out_dim is None else out_dim
        super().__init__(Linear(in_dim, dim), activation(), nn.Dropout(dropout), Linear(dim, out_dim))
        self._fused = activation is nn.SiLU

    def fused_ok(self):
        return self._fused and not (self.training and self[2].p > 0)

    def forward(self, input):
        if not self.fused_ok():
            # non-SiLU activation, or dropout > 0 while training: the HIP linears around torch's activation / nn.Dropout
            # (element-wise glue on device tensors; every BASELINE configuration has dropout 0 and takes the fused path)
            return super().forward(input)
        T = VF.compute_dtype(input)
        return VF.FeedForwardFn.apply(input.to(T), self[0].weight, self[0].bias, self[3].weight, self[3].bias)
