"""DINO self-distillation on the HIP path: the loss module, the momentum-teacher update
and the train step of the reference's train_dino.py:188-288, for the multi-crop ViT student / teacher of models.vit.

  DINOLoss                 drop-in for reference loss.py:89-152 (same constructor, forward(student, teacher, epoch),
                           `center` buffer, fp32 teacher-temperature schedule); one fused HIP sweep computes the value,
                           the gradient w.r.t. the student logits and the teacher column sums (csrc/dino.hip)
  momentum_update          teacher = m * teacher + (1 - m) * student, one multi-tensor kernel (train_dino.py:258-263)
  cancel_last_layer_grad   reference train_util.py:25-31
  dino_train_step          train_dino.py:229-263 for one batch of crops
"""
import os

import torch
import torch.distributed as dist
from torch import nn
from torch.autograd import Function

from . import ops

# The teacher's forward (no_grad, 2 global crops) does not depend on the student's: it runs on its own HIP stream next to the
# student's forward, whose small local-crop launches leave most of the chip idle.  VTX_DINO_TEACHER_STREAM=0: one stream.
_TEACHER_STREAM = os.environ.get("VTX_DINO_TEACHER_STREAM", "1") != "0"
_teacher_streams = {}


def _teacher_stream(device):
    st = _teacher_streams.get(device)
    if st is None:
        st = _teacher_streams[device] = torch.cuda.Stream(device=device)
    return st


class _DinoLossFn(Function):
    @staticmethod
    def forward(ctx, student, teacher, center, n_crop, student_temp, teacher_temp):
        s = student if student.is_contiguous() else student.contiguous()
        t = teacher.detach()
        t = t if t.is_contiguous() else t.contiguous()
        if t.dtype != s.dtype:
            t = t.to(s.dtype)
        loss, ds, bc = ops.dino_loss(s.detach(), t, center.detach().reshape(-1).contiguous(), n_crop, student_temp,
                                     teacher_temp)
        ctx.save_for_backward(ds)
        ctx.mark_non_differentiable(bc)
        return loss, bc

    @staticmethod
    def backward(ctx, gloss, _gbc):
        (ds,) = ctx.saved_tensors
        return ds * gloss.to(ds.dtype), None, None, None, None, None    # (not in place: see train_step._MixLossFn)


class DINOLoss(nn.Module):
    def __init__(self, out_dim, n_crop, warmup_teacher_temperature, teacher_temperature, warmup_teacher_epoch, n_epoch,
                 student_temperature=0.1, center_momentum=0.9):
        super().__init__()
        self.student_temperature = student_temperature
        self.center_momentum = center_momentum
        self.n_crop = n_crop
        self.register_buffer("center", torch.zeros(1, out_dim))
        self.teacher_temperature_schedule = torch.cat((
            torch.linspace(warmup_teacher_temperature, teacher_temperature, warmup_teacher_epoch),
            torch.ones(n_epoch - warmup_teacher_epoch) * teacher_temperature)).tolist()

    def forward(self, student_output, teacher_output, epoch):
        temperature = self.teacher_temperature_schedule[epoch]
        loss, batch_center = _DinoLossFn.apply(student_output, teacher_output, self.center, self.n_crop,
                                               self.student_temperature, temperature)
        self.update_center(batch_center, teacher_output.shape[0])
        return loss

    @torch.no_grad()
    def update_center(self, batch_center, n_rows):
        """batch_center: column sums of this rank's teacher logits (from the loss kernel); reference loss.py:146-152."""
        world = 1
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(batch_center)
            world = dist.get_world_size()
        batch_center = batch_center.view(1, -1) / (n_rows * world)
        self.center.mul_(self.center_momentum).add_(batch_center, alpha=1 - self.center_momentum)


@torch.no_grad()
def momentum_update(teacher, student, momentum):
    """param_k = m * param_k + (1 - m) * param_q over zip(student.parameters(), teacher.parameters())."""
    tp = [p.data for p in teacher.parameters()]
    sp = [p.data for p in student.parameters()]
    if len(tp) != len(sp):
        raise ValueError("teacher / student parameter lists differ")
    ops.ema_update(tp, sp, momentum)


def cancel_last_layer_grad(epoch, model, freeze):
    if epoch >= freeze:
        return
    for n, p in model.named_parameters():
        if "last" in n:
            p.grad = None


def dino_train_step(student, teacher, criterion, optimizer, crops, epoch, momentum, clip_grad_norm=3.0,
                    freeze_last_layer=1, autocast_dtype=torch.bfloat16, grad_accum=1, ddp=None, micro_step=None,
                    ddp_sync="boundary"):
    """One DINO micro-batch on a list of crops (2 global first, then the local ones) resident on the device.  Like the
    reference (train_dino.py:239-263) clip, cancel_last_layer_grad, the optimizer step, zero_grad and the momentum
    update run only when ``(micro_step + 1) % grad_accum == 0``."""
    from .optim import FusedAdamW
    from .train_step import accumulation_boundary, backward_ddp
    boundary = accumulation_boundary(grad_accum, micro_step)
    side = None
    if _TEACHER_STREAM and crops[0].is_cuda and not ops.timing():
        main = torch.cuda.current_stream(crops[0].device)
        side = _teacher_stream(crops[0].device)
        side.wait_stream(main)            # the crops and last step's momentum update of the teacher are main-stream work
    with torch.autocast("cuda", dtype=autocast_dtype, enabled=autocast_dtype is not None):
        if side is not None:
            with torch.cuda.stream(side), torch.no_grad():
                teacher_out = teacher(crops[:2])
        else:
            with torch.no_grad():
                teacher_out = teacher(crops[:2])
        student_out = student(crops)
        if side is not None:
            main.wait_stream(side)
            teacher_out.record_stream(main)
        loss = criterion(student_out, teacher_out, epoch) / grad_accum
    # (multi-crop: the backbone's parameters get one gradient per resolution -- no side stream)
    from . import functional as VF
    with VF.shared_param_backward():      # a layer's second gradient (other crop resolution) is added inside its reduce launch
        backward_ddp(loss, ddp, boundary, ddp_sync, fresh=False)
    if not boundary:
        return loss
    if ddp is not None:
        ddp.finish()
    params = ddp.parameters if ddp is not None else [p for p in student.parameters() if p.requires_grad]
    if isinstance(optimizer, FusedAdamW):
        if epoch < freeze_last_layer and clip_grad_norm and clip_grad_norm > 0:
            # reference order: clip over ALL gradients first, then the last layer's are dropped (train_dino.py:246-250)
            torch.nn.utils.clip_grad_norm_(params, clip_grad_norm)
            cancel_last_layer_grad(epoch, student, freeze_last_layer)
            optimizer.step()
        else:
            cancel_last_layer_grad(epoch, student, freeze_last_layer)
            optimizer.step(max_grad_norm=clip_grad_norm or 0.0)
    else:
        if clip_grad_norm and clip_grad_norm > 0:
            torch.nn.utils.clip_grad_norm_(params, clip_grad_norm)
        cancel_last_layer_grad(epoch, student, freeze_last_layer)
        optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    momentum_update(teacher, student, momentum)
    return loss
