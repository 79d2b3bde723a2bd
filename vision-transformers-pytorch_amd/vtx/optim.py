"""Fused optimizer tail for the HIP training path.

``FusedAdamW`` is a drop-in for ``torch.optim.AdamW`` (same constructor arguments, param groups, state keys
``step`` / ``exp_avg`` / ``exp_avg_sq``, so optimizer checkpoints are interchangeable) whose ``step`` runs the whole
tail of the reference's train step (train.py:285-299) --

    nn.utils.clip_grad_norm_(params, max_norm)   ->   optimizer.step()

-- as two multi-tensor HBM-bound kernels (csrc/optim.hip): a deterministic squared-norm reduction over all gradients
and one AdamW pass that applies the clip coefficient on the fly (gradients themselves are left untouched).
"""
import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._steps = {}

    def _tensors(self):
        out = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda:
                    raise ops.VtxError("FusedAdamW: dense fp32 parameters / gradients on the GPU only")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                out.append((p, g, st, group))
        return out

    def _step_of(self, st):
        """Step count of one parameter as a Python int.  state["step"] stays a tensor (torch.optim.AdamW's format, so
        optimizer checkpoints are interchangeable); reading ~330 of them with .item() and bumping each with a tensor add
        costs ~2 ms of host time per step, so the ints are mirrored here and re-read only when the tensor object changes
        (load_state_dict)."""
        t = st["step"]
        ent = self._steps.get(id(st))
        if ent is None or ent[0] is not t:
            ent = self._steps[id(st)] = [t, int(t)]
        return ent[1]

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=0.0):
        """One AdamW update of every parameter that has a gradient.  ``max_grad_norm > 0`` additionally applies
        ``clip_grad_norm_(all these parameters, max_grad_norm)`` semantics inside the update; returns the total gradient
        norm (device scalar tensor) in that case, else None."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        tensors = self._tensors()
        if not tensors:
            return loss
        ps = [p for p, _, _, _ in tensors]
        gs = [g for _, g, _, _ in tensors]
        norm = None
        if max_grad_norm and max_grad_norm > 0:
            norm = ops.grad_sqnorm(gs)                       # over ALL gradients, whatever their step counts
        # torch.optim.AdamW keeps a step count PER PARAMETER (a parameter that gets its first gradient late -- DINO's
        # last layer is frozen during epoch 0, train_dino.py:250 -- starts at step 1 then): one multi-tensor launch per
        # distinct (betas, eps, step) -- a single one in the steady state
        by_key = {}
        for i, (_, _, st, g) in enumerate(tensors):
            by_key.setdefault((g["betas"], g["eps"], self._step_of(st)), []).append(i)
        for (betas, eps, t0), idx in by_key.items():
            sel = [tensors[i] for i in idx]
            ops.adamw_step([ps[i] for i in idx], [gs[i] for i in idx], [st["exp_avg"] for _, _, st, _ in sel],
                           [st["exp_avg_sq"] for _, _, st, _ in sel], [float(g["lr"]) for _, _, _, g in sel],
                           [float(g["weight_decay"]) for _, _, _, g in sel], norm, float(max_grad_norm or 0.0),
                           betas[0], betas[1], eps, t0 + 1)
        torch._foreach_add_([st["step"] for _, _, st, _ in tensors], 1)        # one call for all step tensors
        for _, _, st, _ in tensors:
            self._steps[id(st)][1] += 1
        return norm[1] if norm is not None else loss
