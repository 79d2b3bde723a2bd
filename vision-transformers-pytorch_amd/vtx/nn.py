"""nn.Module parameter containers whose forward runs on the HIP kernels.

``Linear`` / ``LayerNorm`` subclass the torch modules so that construction, initialisation
(``isinstance(m, nn.Linear)`` in the reference's init_weights), ``state_dict`` keys, shapes and
dtypes are exactly the reference's; only ``forward`` differs.
"""
import contextlib
import os
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
# 1 (default): on a GPU the DropPath masks of a forward are drawn on the host so that the layers can skip dropped branches
# (VTX_DP_COMPACT=0: drawn on the device as in rounds 1-2, every branch computed and scaled)
_DP_HOST_DRAW = os.environ.get("VTX_DP_COMPACT", "1") != "0"


@contextlib.contextmanager
def drop_path_scope(model, batch, device):
    """Draw the DropPath masks of ONE forward pass of ``model`` in one go.

    Every transformer layer draws two per-sample Bernoulli(1 - p) masks (attention branch, MLP branch; reference
    vit.py:60-61, swin_transformer.py:194-195, pvt.py:100-101, layer.py:172-180) -- per layer that is a random-number
    launch plus a divide, ~100 tiny launches per step for Swin-S.  Inside this scope they all come from a single uniform
    draw over (2 x layers, batch), compared against each layer's keep probability and scaled by 1 / keep (3 launches);
    ``drop_path_scale`` then hands out rows in call order.  Same distribution, independent masks; the stream of torch's
    global generator is consumed differently from the reference's per-layer ``bernoulli_`` calls (outside a scope --
    a layer used on its own -- the per-call draw below is used).  On a GPU the draw comes from torch's CPU generator
    (``torch.manual_seed`` seeds both): see the comment at the draw."""
    if not model.training or getattr(_dp, "rows", None) is not None:
        yield
        return
    # (the module walk is cached: 0.3 ms of host time per Swin-S forward; p itself is read every time -- set_drop_path)
    dps = model.__dict__.get("_vtx_dp_modules")
    if dps is None or dps[0] != len(model._modules):
        dps = model.__dict__["_vtx_dp_modules"] = (len(model._modules), [m for m in model.modules() if isinstance(m, _DropPathBase)])
    # (draws per DropPath module and forward: 2, or what the owning layer declares -- a Twins-SVT layer has 4 branches)
    ps = [m.p for m in dps[1] if m.p > 0 for _ in range(getattr(m, "_vtx_draws", 2))]
    if not ps:
        yield
        return
    if device.type == "cuda" and _DP_HOST_DRAW and getattr(model, "_vtx_dp_compaction", False):
        # Drawn on the HOST (torch's CPU generator): the host then knows which samples every branch keeps, and the layer
        # runs each branch over its kept samples only (stochastic-depth compaction, csrc/layer.hip) -- a dropped branch costs
        # nothing instead of being computed and multiplied by 0.  One (2 x layers, batch) uniform draw; the scales and the
        # per-branch sample orders (kept first) go up through pinned memory, asynchronously.
        keep = 1.0 - torch.tensor(ps, dtype=torch.float32).view(-1, 1)
        mask = torch.rand(len(ps), batch) < keep
        host = (mask.to(torch.float32) / keep).pin_memory()
        order = torch.argsort(~mask, dim=1, stable=True).to(torch.int32).pin_memory()     # kept samples first, ascending
        nkeep = mask.sum(1).tolist()
        scale = host.to(device, non_blocking=True)
        perm = order.to(device, non_blocking=True)
        rows = []
        for k in range(len(ps)):
            r = scale[k]
            if 0 < nkeep[k]:
                r._vtx_perm = (perm[k], int(nkeep[k]))
            rows.append(r)
        scale = rows
    else:
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


# ------------------------------------------------------------------------------- host-side pieces shared by the families
# One implementation each of the small host-side rules the transformer families have in common (the reference spells
# them out per model file: vit.py:108-128 / 153-203 / 215-248, swin_transformer.py:15-22 / 307-332, pvt.py:230-262).

def reset_transformer_parameters(module, std=0.02):
    """Initialisation rule of all three families, to be used with ``Module.apply``: Linear weights ~ N(0, std),
    Linear biases 0, LayerNorm affine (1, 0); every other module keeps its torch default."""
    if isinstance(module, nn.LayerNorm):
        nn.init.constant_(module.weight, 1.0)
        nn.init.constant_(module.bias, 0.0)
    elif isinstance(module, nn.Linear):
        module.weight.data.normal_(mean=0.0, std=std)
        if module.bias is not None:
            module.bias.data.zero_()


def pair(value):
    """``v -> (v, v)``; a 2-sequence passes through; any other length is an error."""
    if isinstance(value, (str, bytes)) or not hasattr(value, "__iter__"):
        return (value, value)
    if hasattr(value, "__len__") and len(value) != 2:
        raise ValueError(f"length of {value} (length: {len(value)}) does not match the expected length 2")
    return value


def stochastic_depth_rates(top, n_layer, endpoint):
    """Per-layer drop-path probabilities rising linearly from 0: ``endpoint`` -> the last layer gets ``top`` (ViT, PVT:
    linspace), otherwise layer i gets top * i / n (Swin)."""
    if endpoint:
        return torch.linspace(0, top, n_layer).tolist()
    return [top * float(i) / n_layer for i in range(n_layer)]


def same_resolution_runs(images):
    """[(start, end)) index runs of consecutive entries with equal width -- DINO multi-crop batching: every run is
    concatenated and sent through the backbone once."""
    runs, start = [], 0
    for i in range(1, len(images) + 1):
        if i == len(images) or images[i].shape[-1] != images[start].shape[-1]:
            runs.append((start, i))
            start = i
    return runs


_RESIZE_MAPS = {}


def _bicubic_map(side, n_patch, device):
    """(n_patch, side * side) fp32 matrix of the bicubic resize of a side x side grid by sqrt(n_patch) / side
    (align_corners False, scale not recomputed): the resize is linear, so it is applied to unit images once."""
    key = (side, n_patch, str(device))
    if key not in _RESIZE_MAPS:
        basis = torch.eye(side * side, dtype=torch.float32).reshape(side * side, 1, side, side)
        out = nn.functional.interpolate(basis, scale_factor=(n_patch / (side * side)) ** 0.5, mode="bicubic",
                                        align_corners=False, recompute_scale_factor=False)
        _RESIZE_MAPS[key] = out.reshape(side * side, -1).t().contiguous().to(device)
    return _RESIZE_MAPS[key]


def resize_position_grid(pos_embed, n_patch):
    """(1, 1 + n, dim) class + patch position table -> (1, 1 + n_patch, dim): the square patch grid is resampled
    bicubically by sqrt(n_patch / n) (align_corners False, scale not recomputed), the class row is kept.

    On the GPU the resize runs as one small fp32 GEMM of the cached resampling matrix with the table (and its transpose
    in the backward) instead of torch's bicubic kernels, whose backward alone costs 0.24 ms per DINO step for a
    6 x 6 grid; on the CPU (oracle checks, host tests) torch's interpolate is used directly -- same linear map."""
    n_grid = pos_embed.shape[1] - 1
    if n_grid == n_patch:
        return pos_embed
    dim, side = pos_embed.shape[-1], int(n_grid ** 0.5)
    if pos_embed.is_cuda:
        with torch.autocast("cuda", enabled=False):                      # fp32, like the reference's interpolate
            wmap = _bicubic_map(side, n_patch, pos_embed.device)
            grid_t = pos_embed[0, 1:].float().t()                         # (dim, n): the "weight" of x @ weight^T
            pad = (-n_grid) % 8                                           # the GEMM moves 8-element vectors
            if pad:
                wmap, grid_t = nn.functional.pad(wmap, (0, pad)), nn.functional.pad(grid_t, (0, pad))
            grid = VF.LinearFn.apply(wmap.contiguous(), grid_t.contiguous(), None)
        return torch.cat((pos_embed[:, :1], grid.unsqueeze(0).to(pos_embed.dtype)), 1)
    grid = pos_embed[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
    grid = nn.functional.interpolate(grid, scale_factor=(n_patch / n_grid) ** 0.5, mode="bicubic", align_corners=False,
                                     recompute_scale_factor=False)
    return torch.cat((pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), 1)


def projection_mlp(widths, batch_norm):
    """Linear(widths[0], widths[1]) [BatchNorm1d] GELU ... Linear(widths[-2], widths[-1]) as an nn.Sequential (a single
    Linear for two widths), index layout as in the reference's DINO head so that state_dict keys agree."""
    if len(widths) == 2:
        return Linear(widths[0], widths[1])
    seq = []
    for k, (a, b) in enumerate(zip(widths[:-1], widths[1:])):
        seq.append(Linear(a, b))
        if k < len(widths) - 2:
            if batch_norm:
                seq.append(nn.BatchNorm1d(b))
            seq.append(nn.GELU())
    return nn.Sequential(*seq)
