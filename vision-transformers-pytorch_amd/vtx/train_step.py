"""Counterpart of the reference's supervised train step (train.py:265-299) for the HIP modules.

The reference's train.py never ships; this is the harness the benchmark and the parity tests use.
Same op order: autocast forward -> MixLoss / grad_accum -> backward (DDP all-reduce overlapped) ->
clip_grad_norm_ -> optimizer step -> zero_grad(set_to_none).  With ``vtx.optim.FusedAdamW`` clipping + AdamW run as
two multi-tensor HIP kernels; any torch optimizer works too (clip_grad_norm_ + step()).
MixLoss is one fused HIP kernel (value + gradient) for every reduction the reference has; no CPU fallback.
"""
import torch
from torch import nn

from . import functional as VF
from .ops import VtxError
from .optim import FusedAdamW


class _MixLossFn(torch.autograd.Function):
    """Value + gradient of MixLoss in one sweep (csrc/dino.hip mix_loss_kernel)."""

    @staticmethod
    def forward(ctx, output, target1, target2, interpolation, eps, reduction):
        from . import ops
        loss, dl = ops.mix_loss(output.detach(), target1, target2, interpolation, eps, reduction)
        ctx.save_for_backward(dl)
        ctx.per_row = reduction == "none"
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        g = g.to(dl.dtype)
        # (not in place: a second backward must see dl unscaled)
        return dl * (g.unsqueeze(1) if ctx.per_row else g), None, None, None, None, None


class MixLoss(nn.Module):
    """Label-smoothed KL between log-softmax and a mix of two one-hot targets (reference loss.py:53-86): reductions
    'mean' (train.py's), 'none' (per-sample) and 'sum' (what the reference does for any other string) -- one HIP kernel
    for value and gradient.  Logits of other floating dtypes are computed in fp32 and the gradient cast back.  CPU logits
    are refused: this package has no CPU path by design (the CPU restatement lives in oracle/, outside the product)."""

    def __init__(self, eps=0, reduction="mean"):
        super().__init__()
        self.eps = eps
        self.reduction = reduction

    def forward(self, output, target1, target2, interpolation):
        if not output.is_cuda or output.dim() != 2 or not output.is_floating_point():
            raise VtxError("vtx: MixLoss needs (B, classes) floating-point logits on the GPU (no CPU fallback)")
        if output.dtype not in (torch.float32, torch.bfloat16):
            output = output.float()             # (autograd casts the gradient back to the caller's dtype)
        red = self.reduction if self.reduction in ("mean", "none") else "sum"
        return _MixLossFn.apply(output, target1, target2, interpolation, float(self.eps), red)


def wd_skip(skip_type):
    """Weight-decay skip predicate (reference factory.py:25-39)."""
    def check(name, param):
        if skip_type == "nfnet":
            return "bias" in name or "gain" in name
        if skip_type == "resnet":
            return "bias" in name or "bn" in name or param.ndim == 1
        if skip_type == "vit":
            return "bias" in name or "cls" in name or "norm" in name or param.ndim == 1
        if skip_type == "dino":
            return "bias" in name or param.ndim == 1
        raise ValueError(skip_type)
    return check


def make_param_groups(named_parameters, weight_decay, skip_type="vit"):
    """Two AdamW groups: no-decay first, decay second (reference train_util.py:87-111)."""
    check = wd_skip(skip_type)
    decay, no_decay = [], []
    for n, p in named_parameters:
        if not p.requires_grad:
            continue
        (no_decay if check(n, p) else decay).append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


def accumulation_boundary(grad_accum, micro_step, loader_len=None):
    """True when this micro-batch closes an accumulation window: ``(i + 1) % grad_accum == 0 or (i + 1) == len(loader)``
    with the loader index ``i`` (train.py:285 -- the epoch's last micro-batches step even when the window is not full).
    With grad_accum > 1 the caller MUST pass the index -- a defaulted 0 would never step; without ``loader_len`` only the
    first clause applies (an endless / synthetic stream has no epoch tail)."""
    if grad_accum <= 1:
        return True
    if micro_step is None:
        raise ValueError("vtx: grad_accum > 1 needs micro_step (the loader index i of train.py:285)")
    if loader_len is not None and micro_step >= loader_len:
        raise ValueError("vtx: micro_step is the index INSIDE the epoch (0 <= i < loader_len, train.py:265)")
    return (micro_step + 1) % grad_accum == 0 or (loader_len is not None and micro_step + 1 == loader_len)


def backward_ddp(loss, ddp, boundary, ddp_sync, fresh):
    """``loss.backward()`` with the data-parallel exchange of this micro-batch.

    ``ddp_sync="boundary"`` (default): non-boundary micro-batches accumulate locally (``GradAllReduce.no_sync``), the
    boundary backward all-reduces the accumulated sum, overlapped with itself.  ``"every"``: the reference's pattern --
    DDP without no_sync (train.py:102-107, 283-299) all-reduces on EVERY micro-batch: finish() after each backward, so the
    next micro-batch accumulates onto averaged gradients.  Both give mean_ranks(sum_micro g) (linearity of the mean);
    "boundary" moves 1 / grad_accum of the bytes over xGMI."""
    if ddp_sync not in ("boundary", "every"):
        raise ValueError("vtx: ddp_sync must be 'boundary' or 'every'")
    if ddp is None or not ddp.active:
        with VF.deferred_wgrad(fresh):
            loss.backward()
        return
    if boundary or ddp_sync == "every":
        with VF.deferred_wgrad(fresh):
            loss.backward()
        if not boundary:
            ddp.finish()        # "every": averaged gradients installed before the next micro-batch accumulates onto them
    else:
        with ddp.no_sync(), VF.deferred_wgrad(fresh):
            loss.backward()


def train_step(model, criterion, optimizer, batch, clip_grad_norm=5.0, autocast_dtype=torch.bfloat16,
               grad_accum=1, ddp=None, micro_step=None, ddp_sync="boundary", loader_len=None):
    """One micro-batch of the reference's loop body (train.py:273-299).  ``batch`` = (input NCHW fp32, label1, label2,
    ratio) on the device.  Like the reference, clip + optimizer step + zero_grad run only on accumulation boundaries:
    ``(micro_step + 1) % grad_accum == 0`` (``micro_step`` = the loader index ``i``, required when grad_accum > 1) or, with
    ``loader_len`` = ``len(loader)``, on the epoch's last micro-batch (train.py:285's second clause: a tail shorter than
    the window still steps, and the next epoch starts on clean gradients); in between, gradients accumulate.  The loss of
    every micro-batch is divided by ``grad_accum`` as in the reference (train.py:281), tail included.

    ``ddp`` (vtx.ddp.GradAllReduce) overlaps the gradient all-reduce with backward; its ``finish()`` is the
    only synchronisation point before clipping (``ddp_sync``: see backward_ddp).  Returns the (unsynchronised) loss tensor.
    """
    x, l1, l2, ratio = batch
    boundary = accumulation_boundary(grad_accum, micro_step, loader_len)
    with torch.autocast("cuda", dtype=autocast_dtype, enabled=autocast_dtype is not None):
        out = model(x)
        loss = criterion(out, l1, l2, ratio) / grad_accum
    # Side-stream weight gradients (functional.deferred_wgrad) need "one gradient per parameter, .grad None on entry":
    # true for the first micro-batch after zero_grad(set_to_none) of these single-pass models, not while accumulating.
    fresh = grad_accum == 1 or micro_step % grad_accum == 0
    try:
        backward_ddp(loss, ddp, boundary, ddp_sync, fresh)
        if not boundary:
            return loss
        if ddp is not None:
            ddp.finish()
    except BaseException:
        if ddp is not None:
            ddp.reset()            # an abandoned backward must not leave reduced buckets / handed-out sinks behind
        raise
    if isinstance(optimizer, FusedAdamW):       # clip + AdamW in two multi-tensor HIP kernels (csrc/optim.hip)
        optimizer.step(max_grad_norm=clip_grad_norm or 0.0)
    else:
        if clip_grad_norm and clip_grad_norm > 0:
            torch.nn.utils.clip_grad_norm_(ddp.parameters if ddp is not None else list(model.parameters()),
                                           clip_grad_norm)
        optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    return loss
