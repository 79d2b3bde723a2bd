"""nn.Module parameter containers whose forward runs on the HIP kernels.

``Linear`` / ``LayerNorm`` subclass the torch modules so that construction, initialisation
(``isinstance(m, nn.Linear)`` in the reference's init_weights), ``state_dict`` keys, shapes and
dtypes are exactly the reference's; only ``forward`` differs.
"""
import contextlib
import threading

import torch
from torch import nn

from . import functional as VF


class _DropPathBase(nn.Module):
    """Marker base of models.layer.DropPath (lets drop_path_scope find the modules without importing models)."""


class Linear(nn.Linear):
    def forward(self, input):
        T = VF.compute_dtype(input)
        return VF.LinearFn.apply(input.to(T), self.weight, self.bias)


class LayerNorm(nn.LayerNorm):
    def forward(self, input):
        T = VF.compute_dtype(input)
        return VF.LayerNormFn.apply(input.to(T), self.weight, self.bias, self.eps)


_dp = threading.local()


@contextlib.contextmanager
def drop_path_scope(model, batch, device):
    """Draw the DropPath masks of ONE forward pass of ``model`` in one go.

    Every transformer layer draws two per-sample Bernoulli(1 - p) masks (attention branch, MLP branch; reference
    vit.py:60-61, swin_transformer.py:194-195, pvt.py:100-101, layer.py:172-180) -- per layer that is a random-number
    launch plus a divide, ~100 tiny launches per step for Swin-S.  Inside this scope they all come from a single uniform
    draw over (2 x layers, batch), compared against each layer's keep probability and scaled by 1 / keep (3 launches);
    ``drop_path_scale`` then hands out rows in call order.  Same distribution, independent masks; the stream of torch's
    global generator is consumed differently from the reference's per-layer ``bernoulli_`` calls (outside a scope --
    a layer used on its own -- the per-call draw below is used)."""
    if not model.training or getattr(_dp, "rows", None) is not None:
        yield
        return
    ps = [m.p for m in model.modules() if isinstance(m, _DropPathBase) and m.p > 0 for _ in range(2)]
    if not ps:
        yield
        return
    cache = model.__dict__.setdefault("_vtx_dp_keep", {})
    key = (tuple(ps), str(device))
    keep = cache.get(key)
    if keep is None:
        cache.clear()
        keep = cache[key] = (1.0 - torch.tensor(ps, dtype=torch.float32)).view(-1, 1).to(device)
    scale = (torch.rand(len(ps), batch, device=device) < keep).to(torch.float32) / keep
    _dp.rows, _dp.ps, _dp.next, _dp.batch = scale, ps, 0, batch
    try:
        yield
    finally:
        _dp.rows = None


def drop_path_scale(module_p, training, batch, device):
    """Per-sample DropPath scale mask/(1-p) (reference models/layer.py:172-180) or None when inactive.

    Inside ``drop_path_scope`` the next pre-drawn row is returned; otherwise the mask is drawn here with
    ``Tensor.bernoulli_`` from torch's global generator like the reference does."""
    if not training or module_p == 0:
        return None
    rows = getattr(_dp, "rows", None)
    if rows is not None and _dp.next < len(_dp.ps) and _dp.ps[_dp.next] == module_p and _dp.batch == batch:
        k = _dp.next
        _dp.next = k + 1
        return rows[k]
    keep = 1.0 - module_p
    mask = torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep)
    return mask / keep
