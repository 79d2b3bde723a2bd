"""Data-parallel gradient all-reduce for the one-process-per-GPU training path.

Counterpart of ``nn.parallel.DistributedDataParallel(model, device_ids=[local_rank])`` in the
reference (train.py:102-107): parameters + buffers are broadcast from rank 0 at construction, and
during ``backward()`` the fp32 gradients are averaged across ranks in buckets, in reverse parameter
order, as soon as every gradient of a bucket has been accumulated -- overlapped with the rest of
backward.  The reduction itself is ``torch.distributed.all_reduce`` on the RCCL process group
(``backend="nccl"`` IS RCCL on ROCm), which enqueues on the process group's own side HIP stream after
waiting for the producing stream (event record / stream-wait, no host sync); ``finish()`` makes the
compute stream wait for the last bucket before clipping / the optimizer step.

Designed for MI355X xGMI (point-to-point links, ring collectives per-link bound): few, large buckets
(default 48 MiB; first bucket 8 MiB so communication starts early in backward, last bucket at most 8 MiB so little
is left exposed after backward ends), one flat fp32 buffer
per bucket filled by a single multi-tensor copy, averaged by the collective itself (ncclAvg).
World size 1 bypasses all of it.  Works on CPU tensors with the gloo backend (used by the tests).
"""
import contextlib
import os
from typing import List

import torch
import torch.distributed as dist

MIB = 1 << 20
SLOT_ALIGN = 32      # floats: every parameter's slot in a flat bucket starts on a 128-byte boundary (the weight-gradient
                     # kernels store 16-byte vectors straight into the slots; Swin's 507-float rel_pos tables would
                     # otherwise leave every following slot on an odd 4-byte offset)


class _Bucket:
    def __init__(self, index):
        self.index = index
        self.params: List[torch.nn.Parameter] = []
        self.names: List[str] = []
        self.numel = 0
        self.flat = None
        self.views = []
        self.pending = 0
        self.work = None
        self.handles = {}          # id(param) -> hook handle
        self.trigger = None        # the parameter whose gradient arrived LAST in the learning backward (bucket-level hook)
        self.learned = None        # candidate for `trigger` (set by the counting hooks, adopted in finish())
        self.offsets = []          # start of every parameter's slot in `flat` (floats, SLOT_ALIGN-aligned)
        self.flat_numel = 0


def assign_buckets(named_params, bucket_bytes, first_bucket_bytes, last_bucket_bytes=0):
    """Reverse registration order (gradients become ready roughly back to front), greedy fill."""
    buckets = [_Bucket(0)]
    cap = first_bucket_bytes
    for name, p in reversed(list(named_params)):
        b = buckets[-1]
        nbytes = p.numel() * 4
        if b.numel > 0 and (b.numel * 4 + nbytes) > cap:
            b = _Bucket(len(buckets))
            buckets.append(b)
            cap = bucket_bytes
        b.params.append(p)
        b.names.append(name)
        b.numel += p.numel()
    # The last bucket closes only when the very first layers' gradients arrive, i.e. at the end of backward: whatever it
    # holds is all-reduced with nothing left to overlap.  Keep that exposed tail small: split the last bucket so that its
    # final part (the earliest-registered parameters) is at most `last_bucket_bytes`.
    last = buckets[-1]
    if last_bucket_bytes and len(last.params) > 1 and last.numel * 4 > last_bucket_bytes:
        tail, acc = 0, 0
        for p in reversed(last.params):
            if tail > 0 and (acc + p.numel()) * 4 > last_bucket_bytes:
                break
            acc += p.numel()
            tail += 1
        if 0 < tail < len(last.params):
            nb = _Bucket(len(buckets))
            nb.params, nb.names = last.params[-tail:], last.names[-tail:]
            nb.numel = acc
            last.params, last.names = last.params[:-tail], last.names[:-tail]
            last.numel -= acc
            buckets.append(nb)
    return buckets


class GradAllReduce:
    def __init__(self, module, process_group=None, bucket_bytes=48 * MIB, first_bucket_bytes=8 * MIB,
                 broadcast=True, force=False, last_bucket_bytes=8 * MIB, bucket_hooks=True):
        """``force=True`` keeps the bucket / hook / collective machinery active even for a 1-rank group
        (used to exercise the RCCL path on a single GPU); by default world size 1 bypasses everything."""
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        self.parameters = [p for _, p in named]
        self.buckets = []
        self.bucket_hooks = bucket_hooks       # False: a counting hook per parameter in every backward (round-3 behaviour)
        self.active = self.world > 1 or (force and dist.is_available() and dist.is_initialized())
        if not self.active:
            return
        for _, p in named:
            if p.dtype != torch.float32:
                raise ValueError("GradAllReduce expects fp32 master parameters")
        if broadcast:
            self.broadcast_state()
        self.buckets = assign_buckets(named, bucket_bytes, first_bucket_bytes, last_bucket_bytes)
        self._avg = dist.ReduceOp.AVG if dist.get_backend(process_group) == "nccl" else None
        # A forced ONE-rank group (force=True: tests, bench.py --force-ddp): the mean over one rank is the identity, and RCCL runs AVG on one
        # rank as a `oneRankReduce<FuncPreMulSum>` copy kernel over the whole bucket (Swin-S: 189 MB through 6 launches = 0.5 ms per step,
        # profiles/round6_ddp_overhead_one_gpu.txt) -- a kernel the N > 1 run does not have (its ring kernel averages on the fly).  SUM in
        # place is what a one-rank collective degenerates to: the call, its stream / event bookkeeping and work.wait() stay live.
        if self.world == 1 and os.environ.get("VTX_DDP_ONE_RANK_AVG", "0") != "1":      # (=1: keep the copy kernel -- a stand-in for the N > 1
            self._avg = dist.ReduceOp.SUM                                                 #  ring kernel when looking at stream overlap on one GPU)
        self._slot = {}
        self._sunk = set()         # ids of the parameters whose bucket slot was handed out as a gradient sink since finish()
        self._sync = True          # False inside no_sync(): hooks do not count, nothing is reduced
        self._saw_sync_hook = False   # a hook ran outside no_sync() since the last finish() (i.e. a synced backward happened)
        # Bucket-level trigger hooks verify "every gradient of the bucket is there" by `p.grad is not None`, which only
        # means ARRIVED IN THIS BACKWARD when every .grad was None at its entry.  Gradients that exist on entry -- an
        # accumulation cycle (no_sync), finish() after every micro-batch, zero_grad(set_to_none=False) -- make the check
        # vacuous: a changed arrival order would then reduce a bucket before its last gradient was accumulated, silently.
        # Once such a cycle is seen the object stays on per-parameter counting hooks (exact in every case) for good.
        self._counting_only = not bucket_hooks
        self._pending_counting = False
        for b in self.buckets:
            dev = b.params[0].device
            off = 0
            for p in b.params:
                b.offsets.append(off)
                off += (p.numel() + SLOT_ALIGN - 1) // SLOT_ALIGN * SLOT_ALIGN
            b.flat_numel = off
            b.flat = torch.zeros(off, dtype=torch.float32, device=dev)     # (the pad floats stay zero: harmless in the sum)
            for p, o in zip(b.params, b.offsets):
                b.views.append(b.flat[o:o + p.numel()].view_as(p))
                self._slot[id(p)] = (b, o)
            b.pending = len(b.params)
            self._count_hooks(b)
        if self.active:
            from . import functional as VF
            VF.register_grad_sink_provider(self)

    # -- construction-time sync (DDP broadcasts parameters and buffers from rank 0, train.py:102-107)
    @torch.no_grad()
    def broadcast_state(self):
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            if t.dtype == torch.bool:   # gloo/nccl have no bool broadcast: round-trip through uint8
                u = t.to(torch.uint8)
                dist.broadcast(u, 0, group=self.group)
                t.copy_(u.to(torch.bool))
            else:
                dist.broadcast(t.data, 0, group=self.group)

    def grad_sink(self, p):
        """A FRESH view of ``p``'s slot in its bucket (functional.grad_sink): a kernel that writes the gradient there and
        returns this tensor from backward makes autograd adopt it as ``p.grad`` -- no packing copy before the all-reduce."""
        ent = self._slot.get(id(p)) if self.active else None
        if ent is None or id(p) in self._sunk:
            # At most ONE sink per parameter between two finish() calls: a weight shared by two graph nodes of one
            # backward (DINO's backbone runs once per crop resolution, train_dino.py:229-236) gets two gradients BEFORE
            # autograd's AccumulateGrad sets .grad -- handing both nodes the same slot would make the second kernel
            # overwrite the first node's gradient while it waits in the engine's input buffer.  The second node gets None:
            # a fresh tensor, summed by the engine as usual.
            return None
        self._sunk.add(id(p))
        b, off = ent
        return b.flat[off:off + p.numel()].view_as(p)

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside this context only ACCUMULATE local gradients (torch DDP's ``no_sync``): the bucket hooks do
        not count and nothing is reduced.  The next backward outside the context reduces the accumulated sum -- by linearity
        the mean over ranks of the summed micro-batch gradients, i.e. what the reference's all-reduce-per-micro-batch
        (train.py:283-299 under DDP, no no_sync) arrives at, with 1 / grad_accum of its xGMI traffic."""
        prev, self._sync = self._sync, False
        self._need_counting()       # gradients will exist at the entry of the boundary backward: triggers cannot verify arrival
        try:
            yield
        finally:
            self._sync = prev

    # -- hooks.  A `post_accumulate_grad` hook per parameter is 329 Python calls per Swin-S backward on the autograd thread
    # (the reference's DDP counts in C++, train.py:103).  The order in which a model's gradients arrive is a property of its
    # graph, so the FIRST synced backward counts every parameter of a bucket (``_count_hooks``) and remembers whose gradient
    # arrived last; from then on only that parameter keeps a hook (``_trigger_hook``: one Python call per BUCKET per
    # backward).  The trigger verifies that every gradient of its bucket is there before it launches; if one is missing (the
    # graph changed: a frozen / newly trained parameter, another execution order) the bucket goes back to counting, and
    # finish() launches what is still unreduced once backward has ended -- late, never wrong.
    def _need_counting(self):
        """Switch every bucket to per-parameter counting hooks, permanently (see ``_counting_only``)."""
        if self._counting_only:
            return
        self._counting_only = True
        self._log_counting_once()
        for b in self.buckets:
            if len(b.handles) != len(b.params):
                pend = b.pending
                self._count_hooks(b)
                if b.work is not None:          # already reduced in this cycle: stays reduced until finish()
                    b.pending = pend

    def _log_counting_once(self):
        """ the fallback is silent otherwise: correct, but every backward then runs one Python hook per parameter (329 for
        Swin-S) instead of one per bucket"""
        if not getattr(self, "_logged_counting", False):
            self._logged_counting = True
            import warnings
            warnings.warn("vtx.ddp.GradAllReduce: gradients existed at the entry of a backward (no_sync() accumulation, finish() per "
                          "micro-batch or zero_grad(set_to_none=False)): bucket-level trigger hooks cannot verify arrival there, the object "
                          "stays on per-parameter counting hooks from now on (exact; ~one Python hook call per parameter and backward)",
                          RuntimeWarning, stacklevel=3)

    def _count_hooks(self, b):
        for h in b.handles.values():
            h.remove()
        b.handles = {id(p): p.register_post_accumulate_grad_hook(self._make_hook(b)) for p in b.params}
        b.trigger = b.learned = None
        b.pending = len(b.params)

    def _make_hook(self, bucket):
        def hook(param):
            if not self._sync:
                return
            self._saw_sync_hook = True
            bucket.pending -= 1
            if bucket.pending == 0:
                bucket.learned = param
                self._launch(bucket)
            elif bucket.pending < 0:
                raise RuntimeError(
                    "GradAllReduce: a second backward() reached an already-reduced bucket before finish() -- with gradient "
                    "accumulation either call finish() after EVERY backward (the reference's all-reduce per micro-batch) "
                    "or wrap the non-boundary backwards in no_sync()")
        return hook

    def _trigger_hook(self, bucket):
        def hook(param):
            if not self._sync:
                return
            if bucket.work is not None:
                raise RuntimeError(
                    "GradAllReduce: a second backward() reached an already-reduced bucket before finish() -- with gradient "
                    "accumulation either call finish() after EVERY backward (the reference's all-reduce per micro-batch) "
                    "or wrap the non-boundary backwards in no_sync()")
            self._saw_sync_hook = True
            if any(p.grad is v for p, v in zip(bucket.params, bucket.views)):
                # a gradient installed by the last finish() is still there (finish() after every micro-batch, or
                # zero_grad(set_to_none=False)): gradients existed at the entry of this backward, arrival cannot be
                # verified -> this bucket is reduced by finish(), and every bucket counts per parameter from then on
                bucket.trigger = None
                self._pending_counting = True
            elif all(p.grad is not None for p in bucket.params):
                bucket.pending = 0
                self._launch(bucket)
            else:
                bucket.trigger = None      # order changed: finish() reduces this bucket late and re-installs the counting hooks
        return hook

    def _adopt_trigger(self, b):
        """After a fully counted backward: keep only the last-arriving parameter's hook."""
        if self._counting_only or b.trigger is not None or b.learned is None or len(b.params) == 1:
            return
        for h in b.handles.values():
            h.remove()
        b.trigger = b.learned
        b.handles = {id(b.trigger): b.trigger.register_post_accumulate_grad_hook(self._trigger_hook(b))}

    def reset(self):
        """Re-arm after a step that was ABANDONED between backward() and finish() (a skipped step on a non-finite loss, an
        exception, a bare zero_grad): waits for outstanding collectives, forgets handed-out gradient sinks and makes every
        bucket count again.  Gradients already installed / accumulated are left as they are -- call zero_grad() as well."""
        if not self.active:
            return
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
            b.work = None
            b.pending = len(b.params)
            if b.trigger is None and len(b.handles) != len(b.params):
                self._count_hooks(b)
        self._sunk.clear()
        self._sync = True
        self._saw_sync_hook = False

    def _launch(self, b):
        side = None
        if b.params[0].is_cuda:
            from . import functional as VF
            # weight gradients computed on the side stream (functional.deferred_wgrad): the bucket is packed and reduced
            # BEHIND them on that stream; the main stream only joins at the end of backward / in finish()
            side = VF.side_stream_after_current(b.params[0].device)
        ctx = torch.cuda.stream(side) if side is not None else contextlib.nullcontext()
        with ctx:
            dst, src = [], []
            for p, v in zip(b.params, b.views):           # gradients that already live in the bucket need no packing
                g = p.grad
                if g.data_ptr() != v.data_ptr() or g.stride() != v.stride():
                    dst.append(v)
                    src.append(g)
            if dst:
                torch._foreach_copy_(dst, src)            # one multi-tensor copy of the rest into the flat bucket
            if self._avg is not None:
                b.work = dist.all_reduce(b.flat, op=self._avg, group=self.group, async_op=True)
            else:
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Wait (stream-wise) for every bucket, install the averaged gradients, re-arm for the next backward."""
        if not self.active:
            return
        late = [b for b in self.buckets if b.work is None and b.handles and len(b.handles) != len(b.params)]
        if not self._saw_sync_hook and any(b.work is None for b in self.buckets):
            # nothing ran outside no_sync() since the last finish() (a second finish() in a row, only no_sync backwards):
            # raise BEFORE launching any collective -- ranks must never disagree on which collectives were issued
            for b in self.buckets:
                if b.work is not None:
                    b.work.wait()
                b.work, b.pending = None, len(b.params)
            self._sunk.clear()
            raise RuntimeError("GradAllReduce.finish(): not every bucket was reduced -- no backward() ran outside no_sync() "
                               "since the last finish()")
        if self._pending_counting:
            self._pending_counting = False
            self._counting_only = True
            self._log_counting_once()
        for b in late:
            # a bucket on its bucket-level hook that was not reduced during backward: the trigger's parameter was not the
            # last one this time (or received no gradient).  Backward is over: reduce now if every gradient is there, and
            # count per parameter again until the order has been re-learned
            if all(p.grad is not None for p in b.params):
                b.pending = 0
                self._launch(b)
            self._count_hooks(b)
            if b.work is not None:
                b.pending = 0
        if any(b.pending != 0 for b in self.buckets):
            for b in self.buckets:                      # leave the object usable for the caller's next attempt
                if b.work is not None:
                    b.work.wait()
                b.work, b.pending = None, len(b.params)
            self._sunk.clear()
            missing = [n for b in self.buckets for n, p in zip(b.names, b.params) if p.grad is None]
            if missing:
                raise RuntimeError("GradAllReduce.finish(): parameters received no gradient in this backward "
                                   f"(unused parameters are not supported): {missing[:8]}")
            raise RuntimeError("GradAllReduce.finish(): not every bucket was reduced -- no backward() ran outside no_sync() "
                               "since the last finish()")
        for b in self.buckets:
            b.work.wait()
            if self._avg is None and self.world > 1:
                b.flat.div_(self.world)
            for p, v in zip(b.params, b.views):
                p.grad = v
            b.work = None
            b.pending = len(b.params)
            if self._counting_only and len(b.handles) != len(b.params):
                self._count_hooks(b)
            self._adopt_trigger(b)
        self._sunk.clear()
        self._saw_sync_hook = False

    def remove(self):
        for b in self.buckets:
            for h in b.handles.values():
                h.remove()
            b.handles = {}
