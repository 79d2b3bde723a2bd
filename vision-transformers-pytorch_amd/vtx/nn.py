"""nn.Module parameter containers whose forward runs on the HIP kernels.

``Linear`` / ``LayerNorm`` subclass the torch modules so that construction, initialisation
(``isinstance(m, nn.Linear)`` in the reference's init_weights), ``state_dict`` keys, shapes and
dtypes are exactly the reference's; only ``forward`` differs.
"""
import torch
from torch import nn

from . import functional as VF


class Linear(nn.Linear):
    def forward(self, input):
        T = VF.compute_dtype(input)
        return VF.LinearFn.apply(input.to(T), self.weight, self.bias)


class LayerNorm(nn.LayerNorm):
    def forward(self, input):
        T = VF.compute_dtype(input)
        return VF.LayerNormFn.apply(input.to(T), self.weight, self.bias, self.eps)


def drop_path_scale(module_p, training, batch, device):
    """Per-sample DropPath scale mask/(1-p) (reference models/layer.py:172-180) or None when inactive.

    Draws with ``Tensor.bernoulli_`` from torch's global generator like the reference.
    """
    if not training or module_p == 0:
        return None
    keep = 1.0 - module_p
    mask = torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep)
    return mask / keep
