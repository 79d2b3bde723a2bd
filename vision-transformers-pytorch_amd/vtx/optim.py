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
        self._recs = None           # static per-parameter records (see _records)
        self._plans = {}
        self._checked = set()       # plans validated in the current step

    # ---- host-side bookkeeping.  A step touches ~330 parameters; looking at each one's state dict, validating four
    # tensors per parameter, building four address arrays and bumping 330 CPU step tensors cost ~2.4 ms of host time per
    # step (tools/probe/host_profile.py).  Everything that does not change from step to step is recorded once:
    # (parameter, state, group) triples in group order, the address arrays of params / exp_avg / exp_avg_sq per launch,
    # and ALL ``state["step"]`` tensors as 0-dim views of ONE CPU tensor, so that a step is one ``add_`` (the state
    # layout stays torch.optim.AdamW's: ``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, checkpoints interchangeable).
    def _records(self):
        key = tuple(id(p) for g in self.param_groups for p in g["params"])
        if self._recs is not None and self._recs[0] == key and all(st["step"] is v for (_, st, _), v in zip(self._recs[1], self._recs[3])):
            return self._recs
        recs = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise ops.VtxError("FusedAdamW: dense contiguous fp32 parameters on the GPU only")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                recs.append((p, st, group))
        flat = torch.tensor([float(st["step"]) for _, st, _ in recs], dtype=torch.float32)
        views = []
        for i, (_, st, _) in enumerate(recs):
            st["step"] = flat[i]                       # 0-dim view: float(st["step"]) / state_dict() see the live count
            views.append(st["step"])
        self._recs = (key, recs, flat, views, [int(v) for v in flat.tolist()])
        self._plans = {}
        return self._recs

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._recs = None

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._recs = None

    def _launch_plan(self, idx, recs):
        """Address arrays of one multi-tensor launch over the parameters ``idx`` (a tuple), built once and VALIDATED once
        per step against the live storage addresses of every parameter and moment tensor: ``p.data = ...``, ``module.to()``
        / ``.float()`` after construction or a replaced ``exp_avg_sq`` change an address without changing any Python
        object identity (and ``id()`` values are reused after garbage collection) -- the kernel would read and write freed
        memory.  The plan also holds references to the moment tensors it addresses.  ~0.1 ms per step for 330 tensors."""
        pl = self._plans.get(idx)
        if pl is not None and idx not in self._checked:
            ps = [recs[i][0] for i in idx]
            ms = [recs[i][1]["exp_avg"] for i in idx]
            vs = [recs[i][1]["exp_avg_sq"] for i in idx]
            if pl[6] != tuple(t.data_ptr() for t in ps + ms + vs):
                pl = None                                # some storage moved: rebuild the address arrays
            else:
                self._checked.add(idx)
        if pl is None:
            import ctypes
            ps = [recs[i][0] for i in idx]
            ms = [recs[i][1]["exp_avg"] for i in idx]
            vs = [recs[i][1]["exp_avg_sq"] for i in idx]
            ops._dev(*ps, *ms, *vs)
            for p, m, v in zip(ps, ms, vs):
                if m.shape != p.shape or v.shape != p.shape or m.dtype != torch.float32 or v.dtype != torch.float32:
                    raise ops.VtxError("FusedAdamW: exp_avg / exp_avg_sq must be fp32 tensors of the parameter's shape")
            chunk = ops._lib.load().vtx_opt_chunk()
            numel = (ctypes.c_int64 * len(idx))(*[p.numel() for p in ps])
            pl = self._plans[idx] = (ops._ptr_array(ps), ops._ptr_array(ms), ops._ptr_array(vs), numel,
                                     sum(p.numel() for p in ps), sum((p.numel() + chunk - 1) // chunk for p in ps),
                                     tuple(t.data_ptr() for t in ps + ms + vs), (ms, vs))
            self._checked.add(idx)
        return pl

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=0.0):
        """One AdamW update of every parameter that has a gradient.  ``max_grad_norm > 0`` additionally applies
        ``clip_grad_norm_(all these parameters, max_grad_norm)`` semantics inside the update; returns the total gradient
        norm (device scalar tensor) in that case, else None."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        _, recs, flat, _, counts = self._records()
        self._checked = set()
        live, gs = [], []
        for i, (p, _, _) in enumerate(recs):
            g = p.grad
            if g is None:
                continue
            if g.dtype != torch.float32 or g.is_sparse or not g.is_cuda:
                raise ops.VtxError("FusedAdamW: dense fp32 gradients on the GPU only")
            live.append(i)
            gs.append(g if g.is_contiguous() else g.contiguous())
        if not live:
            return loss
        live = tuple(live)
        norm = None
        if max_grad_norm and max_grad_norm > 0:              # over ALL gradients, whatever their step counts
            pl = self._launch_plan(live, recs)
            norm = ops.grad_sqnorm(gs, static=(pl[3], pl[5]))
        # torch.optim.AdamW keeps a step count PER PARAMETER (a parameter that gets its first gradient late -- DINO's
        # last layer is frozen during epoch 0, train_dino.py:250 -- starts at step 1 then): one multi-tensor launch per
        # distinct (betas, eps, step) -- a single one in the steady state
        by_key = {}
        for j, i in enumerate(live):
            g = recs[i][2]
            by_key.setdefault((g["betas"], g["eps"], counts[i]), []).append(j)
        for (betas, eps, t0), js in by_key.items():
            idx = live if len(js) == len(live) else tuple(live[j] for j in js)
            pl = self._launch_plan(idx, recs)
            ops.adamw_step(None, gs if len(js) == len(live) else [gs[j] for j in js], None, None,
                           [float(recs[i][2]["lr"]) for i in idx], [float(recs[i][2]["weight_decay"]) for i in idx], norm,
                           float(max_grad_norm or 0.0), betas[0], betas[1], eps, t0 + 1, static=pl[:5])
        if len(live) == len(recs):
            flat.add_(1.0)                                   # every state["step"] is a view of this one tensor
        else:
            flat[list(live)] += 1.0
        for i in live:
            counts[i] += 1
        return norm[1] if norm is not None else loss
