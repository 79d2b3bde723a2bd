"""autograd.Functions that bind the HIP kernels into PyTorch's autograd graph.

One Function per fused block of the reference's forward; each backward is
the hand-written kernel sequence, so parameters stay fp32 leaf nn.Parameters that accumulate
``.grad`` through autograd exactly as in the reference (DDP hooks, clip_grad_norm_, any torch
optimizer keep working).  Activations are stored in the compute dtype T (fp32, or bf16 under
``torch.autocast(dtype=torch.bfloat16)``); statistics / parameter gradients are fp32.
"""
import contextlib
import ctypes
import os
import struct
import threading
import warnings

import torch
from torch.autograd import Function

from . import _lib, ops, options
from .ops import ACT_DGELU, ACT_DSILU, ACT_GELU, ACT_SILU, VtxError


def compute_dtype(x):
    """fp32, or bf16 inside torch.autocast(bf16) -- the only knob, same as the reference's train.py:273."""
    if torch.is_autocast_enabled("cuda"):
        dt = torch.get_autocast_dtype("cuda")
        if dt == torch.bfloat16:
            return torch.bfloat16
        raise VtxError(f"vtx: autocast dtype {dt} unsupported on the MI355X path (use torch.bfloat16)")
    if x.dtype in (torch.float32, torch.bfloat16):
        return x.dtype
    raise VtxError(f"vtx: unsupported activation dtype {x.dtype}")


# ------------------------------------------------------------------------------------------- bf16 weight operands
# Per-forward semantics, like the reference under autocast (weights are cast inside every linear call, cached only
# for the duration of one autocast region): the bf16 operands always reflect the CURRENT fp32 parameters.  No cache
# survives a forward pass -- tensor version counters cannot be trusted for that (torch's fused optimizers and
# `.data` EMA updates change parameters without bumping them).  A top-level model wraps its forward in
# ``weight_scope(self, input)``: ONE multi-tensor HIP kernel (csrc/cast.hip) casts every Linear / Conv weight, plain
# and transposed; autograd Functions stash the pair they used for their own backward.
_tls = threading.local()


class WeightPlan:
    """The Linear / Conv2d weights of one model and the device descriptor table of the multi-tensor cast."""

    def __init__(self, module):
        self.params = []
        for m in module.modules():                    # leaf parameters only (weight_norm'd layers recompute theirs)
            w = m._parameters.get("weight") if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d)) else None
            if w is not None and w.dtype == torch.float32 and w.is_cuda:
                self.params.append(w)
        self.key, self.desc, self.offs, self.total, self.ntiles = None, None, None, 0, 0

    def _prepare(self):
        key = tuple((p.data_ptr(), p.device) for p in self.params)
        if key == self.key:
            return
        offs, recs, off, tile0 = [], [], 0, 0
        for p in self.params:
            rows, cols = p.shape[0], p.numel() // p.shape[0]
            tr, tc = (rows + 63) // 64, (cols + 63) // 64
            recs.append(struct.pack("<QqiiiI", p.data_ptr(), off, rows, cols, tile0, tc))
            offs.append(off)
            off += (rows * cols + 63) // 64 * 64                  # keep every matrix 128-byte aligned
            tile0 += tr * tc
        assert struct.calcsize("<QqiiiI") == ops.cast_desc_bytes()
        dev = self.params[0].device
        self.desc = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev)
        self.key, self.offs, self.total, self.ntiles = key, offs, off, tile0

    def cast_all(self):
        """-> {id(param): (plain bf16 view [param shape], transposed bf16 view [in, out])}, fresh buffers."""
        if not self.params:
            return {}
        self._prepare()
        dev = self.params[0].device
        flat = torch.empty(self.total, dtype=torch.bfloat16, device=dev)
        flat_t = torch.empty(self.total, dtype=torch.bfloat16, device=dev)
        ops.cast_weights(self.desc, len(self.params), self.ntiles, flat, flat_t)
        out = {}
        for p, off in zip(self.params, self.offs):
            n, rows = p.numel(), p.shape[0]
            out[id(p)] = (flat[off:off + n].view(p.shape), flat_t[off:off + n].view(n // rows, rows))
        return out


@contextlib.contextmanager
def weight_scope(module, x):
    """Cast all of ``module``'s matrix weights to bf16 once for this forward (no-op in fp32 mode / when nested)."""
    if getattr(_tls, "scope", None) is not None or not x.is_cuda or compute_dtype(x) != torch.bfloat16:
        yield
        return
    plan = module.__dict__.get("_vtx_weight_plan")
    # keyed by the module objects AND their current weight Parameters (load_state_dict(assign=True) / a later
    # weight_norm replace Parameters without touching the module tree).  The Parameters are compared on every forward
    # over the cached module list (60 us); the recursive walk of the tree (0.7 ms for Swin-S) is repeated on every
    # 32nd forward only -- a submodule added after the first forward is picked up within 32 steps.
    if plan is not None:
        plan[3][0] += 1
        mods = plan[2] if plan[3][0] % 32 else list(module.modules())
    else:
        mods = list(module.modules())
    ids = [(id(m), id(m._parameters.get("weight"))) for m in mods]
    if plan is None or plan[0] != ids:
        plan = (ids, WeightPlan(module), mods, [0])
        module.__dict__["_vtx_weight_plan"] = plan
    _tls.scope = plan[1].cast_all()
    try:
        yield
    finally:
        _tls.scope = None


def wcast(p, dtype):
    """Weight operand pair (plain [out, in...], transposed [in, out] or None) of parameter ``p`` in the compute dtype.
    From the enclosing weight_scope if there is one, else cast now (the transposed copy then comes lazily, dgrad)."""
    if p.dtype == dtype:
        return (p.detach(), None)
    sc = getattr(_tls, "scope", None)
    if sc is not None:
        ent = sc.get(id(p))
        if ent is not None and ent[0].dtype == dtype:
            return ent
    return (p.detach().to(dtype), None)


_DGRAD_SPLITK = os.environ.get("VTX_DGRAD_SPLITK", "1") != "0"


def dgrad(dy, wp, T, **epi):
    """dx = epi(dy @ W) for a ``wcast`` pair.  bf16 where the LDS-DMA kernel applies to the transposed problem (K =
    out-features, N = in-features; see ops.glds_ok): forward-layout kernel on the transposed copy; otherwise the
    register-staged NN kernel on W itself."""
    w, wt = wp
    w2 = w.view(w.shape[0], -1)
    if (_DGRAD_SPLITK and T == torch.bfloat16 and not epi and dy.dim() == 2 and w2.shape[0] >= 16384 and w2.shape[1] % 8 == 0 and
            w2.shape[1] >= 64 and dy.shape[0] % 8 == 0 and dy.shape[0] >= 64 and
            ((dy.shape[0] + 63) // 64) * ((w2.shape[1] + 127) // 128) < 128):
        # A very long contraction with a handful of output tiles (DINO's 65 536-way output layer: dx [640, 256] over K = 65 536 is 20
        # tiles of 1 024 k-steps -- 571 us on 20 CUs): split-K.  Both operands are contracted over their ROWS once dy is transposed
        # (W [out, in] already is), which is the weight-gradient kernel's problem: fp32 slabs over the out-features, fixed-order
        # reduce (deterministic), then the cast.
        dx32, _ = ops.wgrad(dy.t().contiguous(), w2 if w2.is_contiguous() else w2.contiguous(), want_bias=False)
        return ops.bias_cast(dx32, None, T)
    if T == torch.bfloat16 and ops.glds_ok(w2.shape[1], w2.shape[0]):
        if wt is None:
            wt = w2.t().contiguous()
        return ops.gemm(dy, wt, 0, **epi)
    return ops.gemm(dy, w2, 1, **epi)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _img(x):
    """Image batch as the patch gathers take it: bf16 channels-last (device input pipeline) as is, else contiguous NCHW."""
    return x if ops.is_nhwc_bf16(x) else _c(x)


# ------------------------------------------------------------------------------------------- a layer's weight gradients
# The weight-gradient GEMMs of a layer's backward feed nothing downstream in that backward.  Two things follow:
#   * GROUPING: the four of a transformer layer (fc2, fc1, proj, qkv) run as ONE launch (ops.wgrad_group): split-K is
#     only there to fill the chip, and four problems together need a quarter of the slices of one -- a quarter of the
#     fp32 slab traffic, and ONE reduce launch sums all slabs of the group instead of 8;
#   * SIDE STREAM (opt-in per backward pass: ``deferred_wgrad()``): the grouped launch goes to a second HIP stream and
#     fills the CUs the activation-gradient chain of the NEXT layers leaves idle (consecutive kernels of one stream run
#     strictly one after the other: every launch pays its ramp-up and its partly filled last round).  fork = the side
#     stream waits for everything enqueued so far (its operands); join = ``side_join()`` ONCE, after backward.
#     That is only correct when nothing on the main stream touches the returned gradients before the join: every
#     parameter must receive exactly one gradient per backward (no accumulation into an existing .grad, no parameter
#     shared by two graph nodes) -- vtx.train_step's supervised step guarantees it and opts in; GradAllReduce joins
#     before it reduces a bucket.  Operands are kept alive until the join (the caching allocator would otherwise hand
#     their memory to the main stream while the side stream still reads it).  Bitwise identical either way.
_SIDE_ENABLED = os.environ.get("VTX_SIDE_WGRAD", "1") != "0"
_SIDE_FENCE_MODE = os.environ.get("VTX_SIDE_FENCE", "0")
_TWINS_SPLITK = os.environ.get("VTX_TWINS_SPLITK", "1") != "0"   # few-row / long-K reduction convs on the split-K launch
# one column-reduce launch per layer (LayerNorm dgamma / dbeta x 2 + rel_pos gradient) instead of three; 0: each kernel reduces its own
_DEFER_REDUCE = os.environ.get("VTX_DEFER_REDUCE", "1") != "0"
_deferred = False


_SIDE_PROBE = os.environ.get("VTX_SIDE_PROBE", "1") != "0"
side_stream_report = {}          # device -> what the concurrency probe saw (bench.py prints it)


def _concurrent_stream(device, candidates=8, spin_us=150):
    """A stream that the GPU really serves CONCURRENTLY with the current one.  HIP multiplexes its streams onto a handful of hardware
    queues (4 by default) in creation order, and torch hands out pool streams round robin: the N-th stream a process asks for may share
    the compute stream's queue -- and then every 'side-stream' weight gradient runs strictly behind the kernel before it, with a
    cross-stream hand-off gap on top (measured: a process that created its RCCL communicator first got such a stream; Swin-S step
    +0.4 ms, concurrency 1.00 in the kernel trace -- profiles/round6_side_stream_queue.md).  Probe: one idle wavefront that holds its
    queue for `spin_us` on each of the two streams (libvtx vtx_debug_spin); concurrent streams take one period, a shared queue two."""
    cur = torch.cuda.current_stream(device)
    lib = _lib.load()
    seen = []
    first = None
    # one spin alone on the compute stream: the yardstick (the kernel counts a nominal 100-MHz wall clock; what matters is the ratio)
    alone = float(spin_us)
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        _lib.check(lib.vtx_debug_spin(spin_us, cur.cuda_stream), "vtx_debug_spin")
        e1.record(cur)
        e1.synchronize()
        alone = e0.elapsed_time(e1) * 1e3
    for k in range(candidates):
        st = torch.cuda.Stream(device=device)
        first = first or st
        best = None
        for rep in range(2):                                  # (first pair: warm-up of the kernel and the streams)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            st.wait_event(e0)
            _lib.check(lib.vtx_debug_spin(spin_us, cur.cuda_stream), "vtx_debug_spin")
            _lib.check(lib.vtx_debug_spin(spin_us, st.cuda_stream), "vtx_debug_spin")
            cur.wait_stream(st)
            e1.record(cur)
            e1.synchronize()
            best = e0.elapsed_time(e1) * 1e3
        seen.append(round(best, 1))
        if best < 1.6 * alone:
            side_stream_report[str(device)] = dict(chosen=k, pair_us=seen, spin_us=spin_us, alone_us=round(alone, 1), concurrent=True)
            return st
    side_stream_report[str(device)] = dict(chosen=0, pair_us=seen, spin_us=spin_us, alone_us=round(alone, 1), concurrent=False)
    warnings.warn(f"vtx: none of {candidates} candidate streams runs concurrently with the compute stream (two {spin_us}-us spin kernels took "
                  f"{seen} us): the side-stream weight gradients will serialise", RuntimeWarning)
    return first


class _SideState:
    def __init__(self, device):
        self.stream = _concurrent_stream(device) if _SIDE_PROBE else torch.cuda.Stream(device=device)
        self.pending = False
        self.keep = []


_side_states = {}


@contextlib.contextmanager
def deferred_wgrad(enabled=True):
    """Within this context (wrap ``loss.backward()``) the layers' grouped weight-gradient launches run on a side stream
    and are joined once, on exit.  Caller's promise: every parameter gets exactly ONE gradient in this backward and has
    ``.grad is None`` on entry (see the section comment above)."""
    global _deferred
    prev, _deferred = _deferred, bool(enabled) and _SIDE_ENABLED
    try:
        yield
    finally:
        _deferred = prev
        side_join()


def side_stream_after_current(dev):
    """The side stream of ``dev`` if weight gradients are pending on it, after making it wait for everything enqueued on
    the current stream so far -- else None.  vtx.ddp launches a bucket's packing copy and all-reduce under it: the
    COLLECTIVE then waits for the side-stream weight gradients (and, through this wait, for the main-stream ones), while the
    main stream -- the dgrad chain, the critical path of backward -- never stalls on a bucket boundary."""
    st = _side_states.get(dev)
    if st is None or not st.pending:
        return None
    st.stream.wait_stream(torch.cuda.current_stream(dev))
    return st.stream


def side_fence(dev, merge=False):
    """With VTX_SIDE_FENCE=1 (or =merge for the PatchMerge LayerNorm backward only) the current stream waits for the
    side-stream weight gradients enqueued so far (they stay pending: no join).  OFF by default since round 4.

    History: round 3 found that the full-size Swin-S step was not bit-reproducible in 10-20 % of runs when the LayerNorm
    backward of a PatchMerge overlapped the preceding layer's side-stream weight gradient, and fenced every backward node
    outside the transformer layers.  Round 4 found the cause (profiles/round4_nondeterminism_root_cause.md): a gfx950
    hazard between a packed-fp32 instruction and the ds_bpermute_b32 that reads its result in the next issue slot -- the
    code hipcc generated for the kernel's two cross-lane sums -- exposed only when another kernel's waves share the CU.
    The fix is one wait state in csrc/vtx_common.h (shfl_xor_f); with it the unfenced step is clean over 2 000+ trials of
    tools/probe/determinism_stress.py, so the fence is a measurement switch now."""
    st = _side_states.get(dev)
    if st is not None and st.pending and (_SIDE_FENCE_MODE == "1" or (_SIDE_FENCE_MODE == "merge" and merge)):
        torch.cuda.current_stream(dev).wait_stream(st.stream)


def side_join():
    """The current stream of every device with outstanding side-stream weight gradients waits for them."""
    for dev, st in _side_states.items():
        if st.pending:
            torch.cuda.current_stream(dev).wait_stream(st.stream)
            st.pending = False
            st.keep.clear()


# A parameter used by TWO graph nodes of one backward (DINO's multi-crop backbone: one pass per crop resolution,
# train_dino.py:229-236) gets two gradients that autograd then adds -- one `add` launch per parameter, 152 per DINO step.
# Inside ``shared_param_backward()`` (vtx.dino wraps loss.backward() in it) the first TransformerLayerFn node of a layer
# registers where its parameter gradients live; the second node's reduce launch ADDS its results there (ops.wgrad_group
# accumulate: out + sum, the very addition autograd would make -- same bits) and returns None for them.  Addresses only: the
# engine's input buffer keeps the first node's tensors alive until AccumulateGrad runs (after both nodes), and a reference
# held here would make AccumulateGrad clone them instead of adopting them.
_shared_grads = None


@contextlib.contextmanager
def shared_param_backward(enabled=True):
    global _shared_grads
    prev, _shared_grads = _shared_grads, ({} if enabled else None)
    try:
        yield
    finally:
        _shared_grads = prev


_grad_sink_providers = []      # weak references to objects with .grad_sink(param) (vtx.ddp.GradAllReduce)


def register_grad_sink_provider(obj):
    import weakref
    _grad_sink_providers.append(weakref.ref(obj))


def grad_sink(param):
    """Destination the weight-gradient kernel should write ``param``'s gradient to, or None.  A data-parallel gradient
    bucket (vtx.ddp) hands out a fresh view of the parameter's slot in its flat buffer: the kernel's final sum lands in
    the bucket, autograd adopts the returned tensor as ``param.grad`` without a copy, and the all-reduce needs no packing
    pass.  Only for the first gradient of a step (``param.grad is None``): later ones accumulate into .grad as usual."""
    if not _grad_sink_providers or param is None or param.grad is not None:
        return None
    for ref in list(_grad_sink_providers):
        obj = ref()
        if obj is None:
            _grad_sink_providers.remove(ref)
            continue
        v = obj.grad_sink(param)
        if v is not None:
            return v
    return None


def layer_wgrads(jobs, rows_per_scale=1, scale_const=0.0, post=None, params=None, colparts=None, accumulate=None):
    """Weight (and bias) gradients of several linears over the same tokens: jobs = [(dy, x, want_bias, rowscale)].
    -> [(dW, db)] (+ ``post(result)`` evaluated on the same stream).  One grouped launch when every problem is eligible
    (bf16, LDS-DMA shapes), else one launch each.  ``params``: the weight Parameters, so that gradients can be written
    straight into a gradient bucket (grad_sink).  ``colparts``: the layer's deferred column reductions (ops.Partials:
    LayerNorm dgamma / dbeta, rel_pos gradient) -- they ride in the grouped launch's ONE reduce launch (else in one
    colreduce_multi launch); the result is then (gradients, [(out0, out1)]).  ``accumulate``: ops.wgrad_group's (the
    caller checked that the grouped launch applies); returns None."""
    outs = None
    if accumulate is None and params is not None and _grad_sink_providers:
        outs = [grad_sink(p) for p in params]
        if not any(o is not None for o in outs):
            outs = None

    def run():
        red = None
        if accumulate is not None:
            return ops.wgrad_group(jobs, rows_per_scale, scale_const, colparts=colparts, accumulate=accumulate)
        if len(jobs) > 1 and ops.wgrad_group_ok(jobs, rows_per_scale, scale_const):
            res = ops.wgrad_group(jobs, rows_per_scale, scale_const, outs=outs, colparts=colparts)
            if colparts is not None:
                res, red = res
        else:
            res = [ops.wgrad(dy, x, want_bias=wb, rowscale=rs, rows_per_scale=rows_per_scale,
                             scale_const=scale_const if rs is not None else 0.0,
                             out=None if outs is None else outs[i]) for i, (dy, x, wb, rs) in enumerate(jobs)]
            if colparts is not None:
                red = ops.colreduce_multi(colparts)
        res = post(res) if post is not None else res
        return (res, red) if colparts is not None else res

    dev = jobs[0][0].device
    # (bench.py's event-sampled steps stay single-stream: with two kernels sharing the chip per-kernel durations are
    #  not attributable)
    if not (_deferred and dev.type == "cuda" and not ops.timing()):
        return run()
    st = _side_states.get(dev)
    if st is None:
        st = _side_states[dev] = _SideState(dev)
    st.stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(st.stream):
        res = run()
    st.keep.append((jobs, colparts))              # (the partials' workspaces were allocated on the main stream too)
    st.pending = True
    return res



# ------------------------------------------------------------------------------------------- one C call per layer
# vtx_layer_fwd / vtx_layer_bwd (csrc/layer.hip) enqueue a transformer layer's launches from ONE descriptor: the same
# entry points in the same order as the call-by-call code below (bit-identical), but one ctypes call and one activation
# buffer per layer instead of ~19 calls and ~28 tensor allocations -- the Python host path was 13.9 ms per Swin-S step
# against 17.9 ms of GPU time (tools/probe/host_time.py).  Taken for the two layer kinds the benchmarks run (window
# attention fast path; global attention without bias / mask); everything else takes the call-by-call path.  In bench.py's
# event-sampled steps the launches of a layer call are timed inside the library (vtx_timer_*).  VTX_LAYER_CALL=0 disables.
_LAYER_CALL = os.environ.get("VTX_LAYER_CALL", "1") != "0"
# a layer's branches are compacted when at least this percentage of its (sample, branch) pairs is dropped (Swin-S B = 128,
# same box, tools/probe/ab_thresh.sh: 2 % 16.90 | 6 % 16.87 | 10 % 16.88 | 14 % 16.97 | never 17.87 ms per step)
_COMPACT_MIN_PCT = int(os.environ.get("VTX_DP_COMPACT_MIN", "8"))
_ALIGN = 256


def _layer_kind(x, rel_pos, meta):
    if not (_LAYER_CALL and _DEFER_REDUCE and x.is_cuda):
        return 0
    if _wattn_ok(rel_pos, meta):
        return _lib.ATTN_WINDOW
    if rel_pos is None and meta.swin is None and meta.mask is None and meta.csr is None:
        return _lib.ATTN_GLOBAL
    return 0


class _LayerPlan:
    """Shape-dependent part of the two descriptors of one layer: buffer layouts, workspace sizes, prefilled structures."""

    def __init__(self, kind, meta, B, M, C, ff, T, want_z, has_rs, rps, dp_c):
        lib = _lib.load()
        es = 2 if T == torch.bfloat16 else 4
        nH, L = meta.n_head, meta.L
        H, W, win, shift = meta.swin if kind == _lib.ATTN_WINDOW else (0, 0, 0, 0)
        n_lse = (M // L) * nH * L
        off = [0]

        def carve(nbytes):
            o = off[0]
            off[0] = (o + nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
            return o
        self.f_off = {k: carve(n * es) for k, n in (("ln1", M * C), ("qkv", 3 * M * C), ("o", M * C), ("x1", M * C),
                                                     ("ln2", M * C), ("h", M * ff))}
        self.f_off["z"] = carve(M * ff * es) if want_z else None
        for k, n in (("mean1", M), ("rstd1", M), ("mean2", M), ("rstd2", M), ("lse", n_lse)):
            self.f_off[k] = carve(4 * n)
        self.f_bytes = off[0]
        off[0] = 0
        self.ln_wsb = lib.vtx_layernorm_bwd_workspace(M, C)
        if kind == _lib.ATTN_WINDOW:
            # (compaction runs the kernel over Bk <= B images; its persistent grid -- and with it the partial rows -- is
            #  not monotonic in the image count)
            self.attn_wsb = max(lib.vtx_wattn_bwd_workspace(b, nH, H, W, win) for b in range(1, B + 1))
        else:
            self.attn_wsb = lib.vtx_attention_bwd_workspace(B, L, nH, 0, 0, 0, 1) if L > 224 else 0
        n4 = ctypes.c_int * 4
        Ns, Ks = n4(C, ff, C, 3 * C), n4(ff, C, C, C)
        self.wgrad_wsb = lib.vtx_wgrad_group_workspace(4, Ns, Ks, M)
        self.slices = lib.vtx_wgrad_group_slices(4, Ns, Ks, M)
        # the backward's ONE grouped weight-gradient launch must apply (bf16, LDS-DMA shapes); else: call-by-call path
        self.ok = bool(lib.vtx_wgrad_group_ok(ops.BF16 if T == torch.bfloat16 else ops.F32, 4, Ns, Ks, M, int(has_rs), int(rps),
                                              float(dp_c)))
        self.b_off = {k: carve(n * es) for k, n in (("dz", M * ff), ("dln2", M * C), ("dx1", M * C), ("dout", M * C),
                                                     ("dqkv", 3 * M * C), ("dln1", M * C))}
        for k, n in (("ln1_ws", self.ln_wsb), ("ln2_ws", self.ln_wsb), ("attn_ws", self.attn_wsb), ("wgrad_ws", self.wgrad_wsb)):
            self.b_off[k] = carve(n) if n else None
        self.b_bytes = off[0]
        common = dict(dtype=ops.BF16 if T == torch.bfloat16 else ops.F32, attn_kind=kind, M=M, C=C, ff=ff, nH=nH, L=L, B=B,
                      H=H, W=W, win=win, shift=int(bool(shift)))
        self.fwd = _lib.LayerFwd(eps=float(meta.eps), **common)
        self.bwd = _lib.LayerBwd(ln_ws_bytes=self.ln_wsb, attn_ws_bytes=self.attn_wsb, wgrad_ws_bytes=self.wgrad_wsb, **common)
        self.ntab = (2 * win - 1) ** 2 if kind == _lib.ATTN_WINDOW else 0


_layer_plans = {}      # by geometry only (a plan holds sizes, offsets and the geometry fields of the descriptors, no addresses)


def _layer_plan(kind, meta, B, M, C, ff, T, want_z, has_rs, rps, dp_c):
    key = (kind, meta.n_head, meta.L, meta.swin if kind == _lib.ATTN_WINDOW else None, meta.eps, B, M, C, ff, T, want_z,
           has_rs, rps, dp_c)
    pl = _layer_plans.get(key)
    if pl is None:
        pl = _layer_plans[key] = _LayerPlan(kind, meta, B, M, C, ff, T, want_z, has_rs, rps, dp_c)
    return pl


def _layer_perms(kind, T, C, ff, s1, s2, head_dim=64, L=0, rps=196):
    """((perm1, Bk1), (perm2, Bk2)) when this layer can run its two branches over their kept samples only (stochastic-depth
    compaction, csrc/layer.hip): the DropPath scales carry the host-drawn sample orders (vtx.nn.drop_path_scope), bf16,
    window attention or the bf16 global-attention fast path, and every GEMM of the layer on the wave-private LDS-DMA kernel
    (N % 128 == 0, K % 64 == 0)."""
    p1 = getattr(s1, "_vtx_perm", None) if s1 is not None else None
    p2 = getattr(s2, "_vtx_perm", None) if s2 is not None else None
    if p1 is None or p2 is None or T != torch.bfloat16 or C % 128 or ff % 128:
        return None
    if 3 * rps < 126:
        return None                      # (the mapped GEMM keeps a tile's sample order in four scalars: <= 4 samples per 128 rows)
    if kind == _lib.ATTN_GLOBAL and (head_dim != 64 or L > 224 or not options.get("SATTN")):
        return None                      # (the bf16 fast-path attention kernels take the sample order; the others do not)
    # the row map costs a little in every kernel (address arithmetic, copy-only tiles): only where enough samples are dropped
    B = p1[0].numel()
    if 100 * (2 * B - p1[1] - p2[1]) < _COMPACT_MIN_PCT * 2 * B:
        return None
    if options.get("GLDS_EPI") != 1 or not options.get("GEMM_GLDS"):
        return None
    return p1, p2


def _copy_desc(d):
    return type(d).from_buffer_copy(d)


def _dp(t):
    return None if t is None else t.data_ptr()


def _check_layer_inputs(x, T, *f32s):
    if not x.is_contiguous() or x.numel() == 0:
        raise VtxError("vtx: layer input must be a non-empty contiguous device tensor")
    for t in f32s:
        if t is not None and (t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous()):
            raise VtxError("vtx: layer parameters must be contiguous float32 device tensors")


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = _c(x)
        y, mean, rstd = ops.layernorm_fwd(x, weight.detach(), bias.detach(), eps)
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        side_fence(dy.device)
        x, weight, mean, rstd = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(_c(dy), x, mean, rstd, weight.detach())
        return dx, dg, db, None


class LinearFn(Function):
    """y = x W^T + b  (nn.Linear).  Output dtype = x dtype.

    The kernels move 16-byte vectors, i.e. need out-features % 8 == 0 (bf16; % 4 in fp32).  Other widths (a 10- or
    100-way classifier) run on a zero-padded weight / bias; the pad columns are sliced off the result and their
    gradients dropped, so the module's parameters keep the reference's shapes."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _c(x)
        wp = wcast(weight, x.dtype)
        N = weight.shape[0]
        pad = (-N) % 8
        b = None if bias is None else bias.detach()
        if pad:
            wp = (torch.nn.functional.pad(wp[0], (0, 0, 0, pad)), None)
            b = None if b is None else torch.nn.functional.pad(b, (0, pad))
        ctx.wp, ctx.pad = wp, pad
        y = ops.gemm(x, wp[0], 0, bias=b)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y[..., :N] if pad else y

    @staticmethod
    def backward(ctx, dy):
        side_fence(dy.device)
        x, weight = ctx.saved_tensors
        N = weight.shape[0]
        dy = torch.nn.functional.pad(dy, (0, ctx.pad)) if ctx.pad else _c(dy)
        dW, db = ops.wgrad(dy, x, want_bias=ctx.has_bias)
        dx = dgrad(dy, ctx.wp, x.dtype) if ctx.needs_input_grad[0] else None
        if ctx.pad:
            dW = dW[:N].contiguous()
            db = None if db is None else db[:N].contiguous()
        return dx, dW.view_as(weight), db


def _act_gemm(ctx, a, w, bias, act):
    """GEMM + fused activation; the pre-activation z is written only when a backward will read it (under
    ``torch.no_grad()`` -- the DINO teacher, evaluation -- nothing needs gradients and that HBM write is skipped)."""
    if any(ctx.needs_input_grad):
        return ops.gemm(a, w, 0, bias=bias, act=act, want_aux=True)
    return ops.gemm(a, w, 0, bias=bias, act=act), None


class FeedForwardFn(Function):
    """PositionwiseFeedForward (reference models/layer.py:186-196): Linear -> SiLU -> Linear, SiLU fused
    into the first GEMM's epilogue, silu' into the second GEMM's dgrad epilogue."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x = _c(x)
        T = x.dtype
        ctx.wp = (wcast(w1, T), wcast(w2, T))
        h, z = _act_gemm(ctx, x, ctx.wp[0][0], b1.detach(), ACT_SILU)
        y = ops.gemm(h, ctx.wp[1][0], 0, bias=b2.detach())
        ctx.save_for_backward(x, w1, w2, z, h)
        return y

    @staticmethod
    def backward(ctx, dy):
        side_fence(dy.device)
        x, w1, w2, z, h = ctx.saved_tensors
        T = x.dtype
        dy = _c(dy)
        dW2, db2 = ops.wgrad(dy, h)
        dz = dgrad(dy, ctx.wp[1], T, act=ACT_DSILU, aux_in=z)
        dW1, db1 = ops.wgrad(dz, x)
        dx = dgrad(dz, ctx.wp[0], T)
        return dx, dW1, db1, dW2, db2


class MlpChainFn(Function):
    """Linear -> act -> Linear -> act -> ... -> Linear with the activation (SiLU or exact GELU) fused into each GEMM's
    epilogue and its derivative into the following dgrad's epilogue: the projection MLP of the DINO head (reference
    models/vit.py:221-241 without BatchNorm).  args: x, act, w0, b0, w1, b1, ..."""

    @staticmethod
    def forward(ctx, x, act, *wb):
        x = _c(x)
        T = x.dtype
        n = len(wb) // 2
        wps = [wcast(wb[2 * i], T) for i in range(n)]
        hs, zs, h = [x], [], x
        for i in range(n):
            b = None if wb[2 * i + 1] is None else wb[2 * i + 1].detach()
            if i < n - 1:
                h, z = _act_gemm(ctx, h, wps[i][0], b, act)
                hs.append(h); zs.append(z)
            else:
                h = ops.gemm(h, wps[i][0], 0, bias=b)
        ctx.save_for_backward(*hs, *zs)
        ctx.wps, ctx.n, ctx.dact, ctx.has_bias = wps, n, (ACT_DSILU if act == ACT_SILU else ACT_DGELU), [wb[2 * i + 1] is not None for i in range(n)]
        return h

    @staticmethod
    def backward(ctx, dy):
        side_fence(dy.device)
        n = ctx.n
        saved = ctx.saved_tensors
        hs, zs = saved[:n], saved[n:]
        T = hs[0].dtype
        d = _c(dy)
        grads = [None] * (2 * n)
        for i in range(n - 1, -1, -1):
            dW, db = ops.wgrad(d, hs[i], want_bias=ctx.has_bias[i])
            grads[2 * i], grads[2 * i + 1] = dW, db
            if i > 0:
                d = dgrad(d, ctx.wps[i], T, act=ctx.dact, aux_in=zs[i - 1])
            elif ctx.needs_input_grad[0]:
                d = dgrad(d, ctx.wps[0], T)
            else:
                d = None
        return (d, None, *grads)


class L2NormFn(Function):
    """F.normalize(x, dim=-1, p=2) (reference models/vit.py:258)."""

    @staticmethod
    def forward(ctx, x, eps):
        y, nrm = ops.l2norm_fwd(_c(x), eps)
        ctx.save_for_backward(y, nrm)
        return y

    @staticmethod
    def backward(ctx, dy):
        side_fence(dy.device)
        y, nrm = ctx.saved_tensors
        return ops.l2norm_bwd(_c(dy), y, nrm), None


class AttentionMeta:
    """Static description of one attention module (geometry + integer tables on the device)."""

    def __init__(self, n_head, dim_head, L, eps=1e-6, swin=None, pos=None, mask=None, csr=None, ntab=0, region=None,
                 fast=True):
        self.n_head, self.dim_head, self.L, self.eps = n_head, dim_head, L, eps
        self.swin, self.pos, self.mask, self.csr, self.ntab = swin, pos, mask, csr, ntab
        # window-attention fast path: region ids of the mask (tables.mask_regions); fast=False when the mask buffer
        # does not have the region structure -> generic masked kernels
        self.region, self.fast = region, fast


def _wattn_ok(rel_pos, meta):
    return (rel_pos is not None and meta.swin is not None and meta.fast and
            ops.wattn_supported(meta.dim_head, meta.swin[2]) and (meta.mask is None or meta.region is not None))


def attn_drop(p, training, keep=None):
    """(p, seed, keep) of one attention-dropout call, None when the reference's F.dropout(attn, p, training) is the identity.
    The 64-bit seed of the counter-based mask hash is DRAWN FROM TORCH'S GENERATOR (one CPU `randint` per active call, ~5 us of host
    time; no device sync): ``torch.manual_seed``, ``torch.set_rng_state`` and a checkpointed RNG state govern the masks exactly as
    they govern the reference's ``F.dropout`` (the former process-global call counter survived a re-seed and was in no
    RNG state).  Data-parallel replicas seeded alike still draw different masks: the rank is mixed in.
    keep: an explicit uint8 keep mask [problems, Lq, Lk] (parity tests).  p >= 1 is refused by the kernels' wrappers (the reference
    yields zeros there; no configuration uses it)."""
    if not training or not p > 0:
        return None
    rank = 0
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank = torch.distributed.get_rank()
    draw = int(torch.randint(0, 1 << 62, (1,)).item())            # (CPU generator: torch.manual_seed seeds it together with the device's)
    seed = (draw * 0x9E3779B97F4A7C15 + rank * 0xA24BAED4963EE407) & 0xFFFFFFFFFFFFFFFF
    return (float(p), seed, keep)


def _attn_forward(qkv, rel_pos, meta, drop=None):
    """-> (o, lse, aux) where aux is what the matching backward needs (the bias tensor of the generic path)."""
    B = qkv.shape[0]
    if drop is None and _wattn_ok(rel_pos, meta):
        o, lse = ops.wattn_fwd(qkv, rel_pos.detach(), meta.pos, meta.region, B, meta.L, meta.n_head, meta.swin)
        return o, lse, None
    bias = ops.relpos_bias(rel_pos.detach(), meta.pos, meta.n_head) if rel_pos is not None else None
    o, lse = ops.attention_fwd(qkv, B, meta.L, meta.n_head, meta.dim_head, swin=meta.swin, bias=bias, mask=meta.mask, drop=drop)
    return o, lse, bias


def _attn_backward(qkv, o, do, lse, aux, meta, rel_pos=None, defer=False, drop=None):
    """-> dqkv, drel_pos (an ops.Partials with ``defer`` on the window-attention fast path: reduced later in one launch
    with the layer's LayerNorm partials)."""
    B = qkv.shape[0]
    if drop is None and _wattn_ok(rel_pos, meta):
        return ops.wattn_bwd(qkv, o, do, lse, rel_pos.detach(), meta.pos, meta.region, B, meta.L, meta.n_head,
                             meta.swin, meta.ntab, defer=defer)
    return ops.attention_bwd(qkv, o, do, lse, B, meta.L, meta.n_head, meta.dim_head, swin=meta.swin, bias=aux,
                             mask=meta.mask, csr=meta.csr, ntab=meta.ntab, drop=drop)


class AttentionCoreFn(Function):
    """softmax(q k^T / sqrt(d) [+ rel-pos bias, -inf mask]) v on the QKV projection output."""

    @staticmethod
    def forward(ctx, qkv, rel_pos, meta, drop=None):
        qkv = _c(qkv)
        o, lse, aux = _attn_forward(qkv, rel_pos, meta, drop)
        ctx.save_for_backward(qkv, o, lse, aux, rel_pos)
        ctx.meta, ctx.drop = meta, drop
        return o

    @staticmethod
    def backward(ctx, do):
        side_fence(do.device)
        qkv, o, lse, aux, rel_pos = ctx.saved_tensors
        dqkv, drel = _attn_backward(qkv, o, _c(do), lse, aux, ctx.meta, rel_pos, drop=ctx.drop)
        return dqkv, drel, None, None


class HaloMeta:
    """Static description of one halo-attention module: geometry + the integer tables on the device."""

    def __init__(self, n_head, dim_head, window, halo, pos, csr, ntab):
        self.n_head, self.dim_head, self.window, self.halo = n_head, dim_head, window, halo
        self.pos, self.csr, self.ntab = pos, csr, ntab


class HaloAttentionFn(Function):
    """Attention core of halo_transformer.MultiHeadedHaloAttention (reference models/halo_transformer.py:58-104) on the bias-free QKV
    projection output (B, H, W, 3 h D): window partition of the queries, (window + 2 halo)^2 neighbourhoods of the keys / values with
    zero rows outside the map, softmax(q k^T / sqrt(D) + rel_pos[pos]) v, inverse partition -> (B, H, W, h D)."""

    @staticmethod
    def forward(ctx, qkv, rel_pos, meta, drop=None):
        qkv = _c(qkv)
        B, H, W, C3 = qkv.shape
        hd = C3 // 3
        w, a, nH = meta.window, meta.halo, meta.n_head
        nW, Lq, Lk = (H // w) * (W // w), w * w, (w + 2 * a) ** 2
        q = ops.window_gather(qkv, B, H, W, 0, hd, w, 0)
        kv = ops.window_gather(qkv, B, H, W, hd, 2 * hd, w, a)
        bias = ops.table_bias(rel_pos.detach(), meta.pos, nH)
        o, lse = ops.xattn_fwd(q.view(B * nW * Lq, hd), kv.view(B * nW * Lk, 2 * hd), B * nW, Lq, Lk, nH, bias, drop=drop)
        out = torch.empty((B, H, W, hd), dtype=qkv.dtype, device=qkv.device)
        ops.window_scatter(o, out, B, H, W, 0, hd, w, 0)
        ctx.save_for_backward(q, kv, o, lse, bias)
        ctx.meta, ctx.geom, ctx.drop = meta, (B, H, W, hd, nW, Lq, Lk), drop
        return out

    @staticmethod
    def backward(ctx, dout):
        side_fence(dout.device)
        q, kv, o, lse, bias = ctx.saved_tensors
        meta = ctx.meta
        B, H, W, hd, nW, Lq, Lk = ctx.geom
        w, a, nH = meta.window, meta.halo, meta.n_head
        do = ops.window_gather(_c(dout), B, H, W, 0, hd, w, 0)
        dq, dkv, dbias = ops.xattn_bwd(q.view(B * nW * Lq, hd), kv.view(B * nW * Lk, 2 * hd), o, do.view(B * nW * Lq, hd), lse,
                                       B * nW, Lq, Lk, nH, bias, drop=ctx.drop)
        dqkv = torch.empty((B, H, W, 3 * hd), dtype=q.dtype, device=q.device)
        ops.window_scatter(dq, dqkv, B, H, W, 0, hd, w, 0)
        ops.window_scatter(dkv, dqkv, B, H, W, hd, 2 * hd, w, a)          # sums over the neighbourhoods that hold a token
        drel = ops.table_bias_bwd(dbias, meta.csr, meta.ntab, nH)
        return dqkv, drel, None, None


class TransformerLayerFn(Function):
    """One pre-LN transformer block (reference models/vit.py:59-63, models/swin_transformer.py:193-197):
         x1 = x  + s1 * proj(attn(qkv(LN1(x))))        y = x1 + s2 * fc2(silu(fc1(LN2(x1))))
    as 7 kernels forward (LN, GEMM, attention, GEMM+residual, LN, GEMM+SiLU, GEMM+residual); s1/s2 are
    the per-sample DropPath scales mask/(1-p) (models/layer.py:172-180) or None."""

    @staticmethod
    def forward(ctx, x, ln1_w, ln1_b, qkv_w, qkv_b, rel_pos, proj_w, proj_b, ln2_w, ln2_b, fc1_w, fc1_b, fc2_w,
                fc2_b, s1, s2, dp_c, meta):
        x = _c(x)
        T = x.dtype
        B, C = x.shape[0], x.shape[-1]
        rps = (x.numel() // C) // B
        # identities of the 12 parameters in the order backward returns their gradients (shared_param_backward)
        ctx.pids = tuple(map(id, (ln1_w, ln1_b, qkv_w, qkv_b, proj_w, proj_b, ln2_w, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b)))
        ctx.meta, ctx.rps, ctx.dp_c = meta, rps, float(dp_c)
        kind = _layer_kind(x, rel_pos, meta)
        if kind and (qkv_w.shape[0] != 3 * C or proj_w.shape[1] != C or meta.n_head * meta.dim_head != C or qkv_b is None or
                     proj_b is None or fc1_b is None or fc2_b is None):
            kind = 0                         # (the one-call path assumes heads x head dim == dim and biased linears)
        pl = None
        if kind:
            pl = _layer_plan(kind, meta, B, x.numel() // C, C, fc1_w.shape[0], T, any(ctx.needs_input_grad),
                             s1 is not None or s2 is not None, rps, float(dp_c))
            if not pl.ok:
                kind = 0
        ctx.kind = kind
        if kind:
            return TransformerLayerFn._forward_one_call(ctx, kind, pl, x, ln1_w, ln1_b, qkv_w, qkv_b, rel_pos, proj_w, proj_b,
                                                        ln2_w, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b, s1, s2, meta)
        ln1, mean1, rstd1 = ops.layernorm_fwd(x, ln1_w.detach(), ln1_b.detach(), meta.eps)
        wq, wo, w1, w2 = ctx.wp = (wcast(qkv_w, T), wcast(proj_w, T), wcast(fc1_w, T), wcast(fc2_w, T))
        qkv = ops.gemm(ln1, wq[0], 0, bias=qkv_b.detach())
        o, lse, bias = _attn_forward(qkv, rel_pos, meta)
        x1 = ops.gemm(o, wo[0], 0, bias=proj_b.detach(), resid=x, rowscale=s1, rows_per_scale=rps)
        ln2, mean2, rstd2 = ops.layernorm_fwd(x1, ln2_w.detach(), ln2_b.detach(), meta.eps)
        h, z = _act_gemm(ctx, ln2, w1[0], fc1_b.detach(), ACT_SILU)
        y = ops.gemm(h, w2[0], 0, bias=fc2_b.detach(), resid=x1, rowscale=s2, rows_per_scale=rps)
        ctx.save_for_backward(x, ln1_w, qkv_w, proj_w, ln2_w, fc1_w, fc2_w, mean1, rstd1, ln1, qkv, o, lse, x1,
                              mean2, rstd2, ln2, z, h, bias, s1, s2, rel_pos)
        return y

    # ---- the same layer through ONE C call (csrc/layer.hip)
    @staticmethod
    def _forward_one_call(ctx, kind, pl, x, ln1_w, ln1_b, qkv_w, qkv_b, rel_pos, proj_w, proj_b, ln2_w, ln2_b, fc1_w, fc1_b,
                          fc2_w, fc2_b, s1, s2, meta):
        T = x.dtype
        B, C = x.shape[0], x.shape[-1]
        M, ff = x.numel() // C, fc1_w.shape[0]
        _check_layer_inputs(x, T, ln1_w, ln1_b, qkv_b, proj_b, ln2_w, ln2_b, fc1_b, fc2_b, s1, s2)
        want_z = any(ctx.needs_input_grad)
        wq, wo, w1, w2 = ctx.wp = (wcast(qkv_w, T), wcast(proj_w, T), wcast(fc1_w, T), wcast(fc2_w, T))
        buf = torch.empty(pl.f_bytes, dtype=torch.uint8, device=x.device)
        y = torch.empty_like(x)
        base, fo = buf.data_ptr(), pl.f_off
        d = _copy_desc(pl.fwd)
        d.rows_per_scale = ctx.rps
        d.x, d.y = x.data_ptr(), y.data_ptr()
        d.ln1_w, d.ln1_b, d.ln2_w, d.ln2_b = ln1_w.data_ptr(), ln1_b.data_ptr(), ln2_w.data_ptr(), ln2_b.data_ptr()
        d.wq, d.wo, d.w1, d.w2 = wq[0].data_ptr(), wo[0].data_ptr(), w1[0].data_ptr(), w2[0].data_ptr()
        d.bq, d.bo, d.b1, d.b2 = qkv_b.data_ptr(), proj_b.data_ptr(), fc1_b.data_ptr(), fc2_b.data_ptr()
        if kind == _lib.ATTN_WINDOW:
            d.rel_pos, d.pos, d.region = rel_pos.data_ptr(), meta.pos.data_ptr(), _dp(meta.region)
        d.s1, d.s2 = _dp(s1), _dp(s2)
        ctx.perms = _layer_perms(kind, T, C, ff, s1, s2, meta.dim_head, meta.L, M // B)
        if ctx.perms is not None and any(w[1] is None for w in (wq, wo, w1, w2)):
            # the mapped backward (vtx_layer_bwd) multiplies by the TRANSPOSED bf16 weight copies; they only exist inside a
            # weight_scope (a top-level model's forward).  Called without one -- VisionTransformer.forward_feature() directly,
            # bf16 parameters -- the layer runs uncompacted instead of failing in backward
            ctx.perms = None
        if ctx.perms is not None:
            (p1, d.Bk1), (p2, d.Bk2) = ctx.perms
            d.perm1, d.perm2 = p1.data_ptr(), p2.data_ptr()
        d.ln1, d.qkv, d.o, d.x1, d.ln2, d.h = (base + fo["ln1"], base + fo["qkv"], base + fo["o"], base + fo["x1"],
                                               base + fo["ln2"], base + fo["h"])
        d.z = base + fo["z"] if want_z else None
        d.mean1, d.rstd1, d.mean2, d.rstd2, d.lse = (base + fo["mean1"], base + fo["rstd1"], base + fo["mean2"],
                                                     base + fo["rstd2"], base + fo["lse"])
        _lib.check(_lib.load().vtx_layer_fwd(ctypes.byref(d), ops._stream()), "vtx_layer_fwd")
        ctx.save_for_backward(x, buf, ln1_w, qkv_w, proj_w, ln2_w, fc1_w, fc2_w, s1, s2, rel_pos, fc1_b)
        ctx.plan = pl
        return y

    @staticmethod
    def _backward_one_call(ctx, dy):
        x, buf, ln1_w, qkv_w, proj_w, ln2_w, fc1_w, fc2_w, s1, s2, rel_pos, fc1_b = ctx.saved_tensors
        pl, m, kind = ctx.plan, ctx.meta, ctx.kind
        if pl.f_off["z"] is None:
            raise VtxError("vtx: this layer's forward ran without a graph (no pre-activation was kept)")
        dy = _c(dy)
        dev = x.device
        C = x.shape[-1]
        ff = fc1_w.shape[0]
        wq, wo, w1, w2 = ctx.wp
        scratch = torch.empty(pl.b_bytes, dtype=torch.uint8, device=dev)
        dx = torch.empty_like(x)
        f32 = dict(dtype=torch.float32, device=dev)
        shared = _shared_grads if rel_pos is None else None
        acc = None
        if shared is not None:
            first = [shared.get(pid) for pid in ctx.pids]
            if all(a is not None for a in first) and pl.slices >= 2:
                acc = first
        outs = None
        if acc is None:
            sinks = [grad_sink(p) for p in (qkv_w, proj_w, fc1_w, fc2_w)] if _grad_sink_providers else (None,) * 4
            dWq, dWo, dW1, dW2 = [sk if sk is not None else torch.empty(p.shape, **f32)
                                  for sk, p in zip(sinks, (qkv_w, proj_w, fc1_w, fc2_w))]
            # the nine small gradients: one allocation, one split (views of it go back to autograd)
            nrel = pl.ntab * m.n_head
            small = torch.empty(9 * C + ff + nrel, **f32).split((C, C, 3 * C, C, C, C, ff, C) + ((nrel,) if nrel else ()))
            dg1, dbe1, dbq, dbo, dg2, dbe2, db1, db2 = small[:8]
            drel = small[8].view(pl.ntab, m.n_head) if nrel else None
            outs = (dg1, dbe1, dWq, dbq, dWo, dbo, dg2, dbe2, dW1, db1, dW2, db2)
            gp = [t.data_ptr() for t in outs]
        else:
            gp, drel = acc, None
        base, fo, sb, bo = buf.data_ptr(), pl.f_off, scratch.data_ptr(), pl.b_off
        d = _copy_desc(pl.bwd)
        d.rows_per_scale, d.scale_const, d.accumulate = ctx.rps, ctx.dp_c, int(acc is not None)
        d.dy, d.x, d.dx = dy.data_ptr(), x.data_ptr(), dx.data_ptr()
        (d.ln1, d.qkv, d.o, d.x1, d.ln2, d.z, d.h) = (base + fo["ln1"], base + fo["qkv"], base + fo["o"], base + fo["x1"],
                                                      base + fo["ln2"], base + fo["z"], base + fo["h"])
        d.mean1, d.rstd1, d.mean2, d.rstd2, d.lse = (base + fo["mean1"], base + fo["rstd1"], base + fo["mean2"],
                                                     base + fo["rstd2"], base + fo["lse"])
        d.ln1_w, d.ln2_w = ln1_w.data_ptr(), ln2_w.data_ptr()
        d.b1 = fc1_b.data_ptr()                     # (the fused MLP's backward recomputes z)
        d.wq, d.wo, d.w1, d.w2 = (wq[0].data_ptr(), wo[0].data_ptr(), w1[0].data_ptr(), w2[0].data_ptr())
        d.wqt, d.wot, d.w1t, d.w2t = _dp(wq[1]), _dp(wo[1]), _dp(w1[1]), _dp(w2[1])
        if kind == _lib.ATTN_WINDOW:
            inv_cells, inv_count = ops._pos_inverse(m.pos, m.ntab)
            d.rel_pos, d.pos, d.region = rel_pos.data_ptr(), m.pos.data_ptr(), _dp(m.region)
            d.inv_cells, d.inv_count = inv_cells.data_ptr(), inv_count
            d.drel = drel.data_ptr()
        d.s1, d.s2 = _dp(s1), _dp(s2)
        if ctx.perms is not None:
            (p1, d.Bk1), (p2, d.Bk2) = ctx.perms
            d.perm1, d.perm2 = p1.data_ptr(), p2.data_ptr()
        d.dz, d.dln2, d.dx1, d.dout, d.dqkv, d.dln1 = (sb + bo["dz"], sb + bo["dln2"], sb + bo["dx1"], sb + bo["dout"],
                                                       sb + bo["dqkv"], sb + bo["dln1"])
        d.ln1_ws, d.ln2_ws, d.wgrad_ws = sb + bo["ln1_ws"], sb + bo["ln2_ws"], sb + bo["wgrad_ws"]
        d.attn_ws = sb + bo["attn_ws"] if bo["attn_ws"] is not None else None
        (d.dg1, d.dbe1, d.dWq, d.dbq, d.dWo, d.dbo, d.dg2, d.dbe2, d.dW1, d.db1, d.dW2, d.db2) = gp
        side = None
        if _deferred and not ops.timing():          # (bench.py's event-sampled steps stay single-stream: attributable durations)
            st = _side_states.get(dev)
            if st is None:
                st = _side_states[dev] = _SideState(dev)
            side = st.stream.cuda_stream
            st.keep.append((scratch, buf, dy, x, s1, s2, ctx.wp, ctx.perms))     # what the side stream still reads after this returns
            st.pending = True
        _lib.check(_lib.load().vtx_layer_bwd(ctypes.byref(d), ops._stream(), side), "vtx_layer_bwd")
        if acc is not None:
            return (dx,) + (None,) * 17
        if shared is not None and not any(pid in shared for pid in ctx.pids):
            for pid, a in zip(ctx.pids, gp):
                shared[pid] = a
        return (dx, dg1, dbe1, dWq, dbq, drel, dWo, dbo, dg2, dbe2, dW1, db1, dW2, db2, None, None, None, None)

    @staticmethod
    def backward(ctx, dy):
        if ctx.kind:
            return TransformerLayerFn._backward_one_call(ctx, dy)
        (x, ln1_w, qkv_w, proj_w, ln2_w, fc1_w, fc2_w, mean1, rstd1, ln1, qkv, o, lse, x1, mean2, rstd2, ln2, z, h,
         bias, s1, s2, rel_pos) = ctx.saved_tensors
        m, rps, dp_c = ctx.meta, ctx.rps, ctx.dp_c
        wq, wo, w1, w2 = ctx.wp
        T = x.dtype
        dy = _c(dy)
        B = x.shape[0]
        # ---- MLP branch
        dz = dgrad(dy, w2, T, act=ACT_DSILU, aux_in=z, rowscale=s2, rows_per_scale=rps)
        dln2 = dgrad(dz, w1, T)
        if _DEFER_REDUCE:
            dx1, part2 = ops.layernorm_bwd(dln2, x1, mean2, rstd2, ln2_w.detach(), dres=dy, defer=True)
        else:
            dx1, dg2, dbe2 = ops.layernorm_bwd(dln2, x1, mean2, rstd2, ln2_w.detach(), dres=dy)
        # ---- attention branch
        do = dgrad(dx1, wo, T, rowscale=s1, rows_per_scale=rps)
        dqkv, drel = _attn_backward(qkv, o, do, lse, bias, m, rel_pos, defer=_DEFER_REDUCE)
        dln1 = dgrad(dqkv, wq, T)
        if _DEFER_REDUCE:
            dx, part1 = ops.layernorm_bwd(dln1, x, mean1, rstd1, ln1_w.detach(), dres=dx1, defer=True)
            # ---- the layer's small column reductions (LayerNorm dgamma / dbeta twice, rel_pos gradient) ride in the
            #      reduce launch of the grouped weight gradients below: no launch of their own
            parts = [part2, part1] + ([drel] if isinstance(drel, ops.Partials) else [])
        else:
            dx, dg1, dbe1 = ops.layernorm_bwd(dln1, x, mean1, rstd1, ln1_w.detach(), dres=dx1)
            parts = None
        # ---- the four weight gradients: one grouped launch (dropped samples' rows are skipped, 1/(1-p) on the accumulators)
        jobs = [(dy, h, True, s2), (dz, ln2, True, None), (dx1, o, True, s1), (dqkv, ln1, True, None)]
        shared = _shared_grads if (parts is not None and len(parts) == 2 and rel_pos is None) else None
        if shared is not None:
            first = [shared.get(pid) for pid in ctx.pids]
            if all(a is not None for a in first) and ops.wgrad_group_ok(jobs, rps, dp_c) and ops.wgrad_group_slices(jobs) >= 2:
                # this layer's parameters already have a gradient in this backward (the other crop resolution's pass): add
                # onto it inside the reduce launch instead of handing autograd a second tensor per parameter to add
                g1, b1, wq, bq, wo, bo, g2, b2, w1, bb1, w2, bb2 = first
                layer_wgrads(jobs, rps, dp_c, colparts=parts, accumulate=((w2, w1, wo, wq), (bb2, bb1, bo, bq), [(g2, b2), (g1, b1)]))
                return (dx,) + (None,) * 17
        res = layer_wgrads(jobs, rps, dp_c, params=(fc2_w, fc1_w, proj_w, qkv_w), colparts=parts)
        if parts is not None:
            res, red = res
            (dg2, dbe2), (dg1, dbe1) = red[0], red[1]
            if isinstance(drel, ops.Partials):
                drel = red[2][0].view(m.ntab, m.n_head)
        (dW2, db2), (dW1, db1), (dWo, dbo), (dWq, dbq) = res
        if shared is not None and not any(pid in shared for pid in ctx.pids):
            for pid, t in zip(ctx.pids, (dg1, dbe1, dWq, dbq, dWo, dbo, dg2, dbe2, dW1, db1, dW2, db2)):
                shared[pid] = t.data_ptr()
        return (dx, dg1, dbe1, dWq, dbq, drel, dWo, dbo, dg2, dbe2, dW1, db1, dW2, db2, None, None, None, None)


class PatchMergeFn(Function):
    """PatchMerge (reference models/swin_transformer.py:216-229): patchify(2) -> LayerNorm(4C, 1e-5) -> Linear
    (no bias); the 2x2 gather is folded into the LayerNorm kernel's row addressing."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w, eps):
        x = _c(x)
        H, W = x.shape[1], x.shape[2]
        ln, mean, rstd = ops.layernorm_fwd(x, ln_w.detach(), ln_b.detach(), eps, merge_hw=(H, W))
        ctx.wp = wcast(w, x.dtype)
        y = ops.gemm(ln, ctx.wp[0], 0)
        ctx.save_for_backward(x, ln_w, w, ln, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ln_w, w, ln, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        dW, _ = ops.wgrad(dy, ln, want_bias=False)
        dln = dgrad(dy, ctx.wp, x.dtype)
        side_fence(dy.device, merge=True)   # (the LayerNorm backward is the launch that must not overlap: see side_fence)
        dx, dg, db = ops.layernorm_bwd(dln, x, mean, rstd, ln_w.detach(), merge_hw=(x.shape[1], x.shape[2]))
        return dx, dg, db, dW, None


class SwinPatchEmbedFn(Function):
    """swin.PatchEmbedding on the NCHW image (reference models/swin_transformer.py:208-213 + the permute at :371):
    (py,px,c) patch gather -> Linear(3 p^2 -> C) -> LayerNorm(C, 1e-5)."""

    @staticmethod
    def forward(ctx, x_nchw, w, b, ln_w, ln_b, patch, eps, dtype):
        patches = ops.patch_gather(_img(x_nchw), patch, 0, dtype)
        t = ops.gemm(patches, wcast(w, dtype)[0], 0, bias=b.detach())
        y, mean, rstd = ops.layernorm_fwd(t, ln_w.detach(), ln_b.detach(), eps)
        ctx.save_for_backward(patches, t, mean, rstd, ln_w)
        return y

    @staticmethod
    def backward(ctx, dy):
        side_fence(dy.device)
        patches, t, mean, rstd, ln_w = ctx.saved_tensors
        dt, dg, db = ops.layernorm_bwd(_c(dy), t, mean, rstd, ln_w.detach())
        dW, dbias = ops.wgrad(dt, patches)
        return None, dW, dbias, dg, db, None, None, None


class VitPatchEmbedFn(Function):
    """vit.PatchEmbedding (reference models/vit.py:69-76): Conv2d(3, C, p, stride p) + flatten + transpose as an
    im2col gather + GEMM; output (B, n_patch, C)."""

    @staticmethod
    def forward(ctx, x_nchw, w, b, dtype):
        C, Cin, p, _ = w.shape
        patches = ops.patch_gather(_img(x_nchw), p, 1, dtype)
        B, gh, gw, K = patches.shape
        y = ops.gemm(patches.view(B, gh * gw, K), wcast(w, dtype)[0].view(C, K), 0, bias=b.detach())
        ctx.save_for_backward(patches)
        ctx.wshape = w.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        side_fence(dy.device)
        (patches,) = ctx.saved_tensors
        dW, db = ops.wgrad(_c(dy), patches)
        return None, dW.view(ctx.wshape), db, None


class VitAssembleFn(Function):
    """cat(cls, patches) + pos_embed (reference models/vit.py:140-143)."""

    @staticmethod
    def forward(ctx, patches, cls_token, pos_embed):
        L, C = pos_embed.shape[-2], pos_embed.shape[-1]
        out = ops.vit_assemble_fwd(_c(patches), _c(cls_token.detach().reshape(C)),
                                   _c(pos_embed.detach().reshape(L, C)))
        ctx.shapes = (cls_token.shape, pos_embed.shape)
        return out

    @staticmethod
    def backward(ctx, dx):
        side_fence(dx.device)
        dpatches, dcls, dpos = ops.vit_assemble_bwd(_c(dx))
        return dpatches, dcls.view(ctx.shapes[0]), dpos.view(ctx.shapes[1])


class TokenMeanFn(Function):
    """Mean over the tokens of each image: (B, ..., C) -> (B, C)."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        B, C = x.shape[0], x.shape[-1]
        Tn = x.numel() // (B * C)
        ctx.dims = (B, Tn, C, x.shape)
        return ops.token_mean_fwd(x, B, Tn, C)

    @staticmethod
    def backward(ctx, dy):
        side_fence(dy.device)
        B, Tn, C, shape = ctx.dims
        return ops.token_mean_bwd(_c(dy), B, Tn, C, shape)


# ------------------------------------------------------------------------------------------------- PVT (models/pvt.py)
def conv_as_rows(w):
    """Conv2d weight (out, in, p, p) of a wcast pair -> GEMM weight (out, p*p*in) with columns in (py, px, c) order, the
    column order of ops.patchify_fwd (token-major gather); the transposed copy is made lazily by dgrad."""
    out_ch = w.shape[0]
    return (w.permute(0, 2, 3, 1).reshape(out_ch, -1).contiguous(), None)


class PvtMeta:
    """Static description of one PVT transformer layer: heads, token grid, spatial reduction, leading cls tokens."""

    def __init__(self, n_head, height, width, reduction, skip, eps=1e-6, twins=False):
        self.n_head, self.height, self.width, self.reduction, self.skip, self.eps = n_head, height, width, reduction, skip, eps
        # twins: the reduction conv's operand is gathered the way models/twins.py:69-70 reshapes its 4-D input
        # (ops.twins_subsample_fwd) instead of the token-grid patch gather of pvt.py:44-46
        self.twins = twins


class PatchifyFn(Function):
    """Token-major features (B, skip + H*W, C) -> patch matrix (B*(H/p)*(W/p), p*p*C), columns (py, px, c): the im2col of a
    stride = kernel convolution (PVT's spatial-reduction conv, pvt.py:44-46); backward = the inverse scatter."""

    @staticmethod
    def forward(ctx, x, H, W, p, skip):
        x = _c(x)
        B, _, C = x.shape
        ctx.geom = (x.shape, H, W, p, skip)
        return ops.patchify_fwd(x, B, H, W, C, p, skip)

    @staticmethod
    def backward(ctx, dout):
        side_fence(dout.device)
        shape, H, W, p, skip = ctx.geom
        dx = torch.zeros(shape, dtype=dout.dtype, device=dout.device) if skip else \
            torch.empty(shape, dtype=dout.dtype, device=dout.device)
        ops.patchify_bwd(_c(dout), dx, shape[0], H, W, shape[2], p, skip)
        return dx, None, None, None, None


class TwinsSubsampleFn(Function):
    """Channels-last features (B, H, W, C) -> the patch matrix of twins.MultiHeadedAttention's reduction conv, with the
    reference's reshape kept as written (models/twins.py:69-70; ops.twins_subsample_fwd); backward = the inverse scatter."""

    @staticmethod
    def forward(ctx, x, r):
        x = _c(x)
        ctx.geom = (x.shape, r)
        B, H, W, C = x.shape
        return ops.twins_subsample_fwd(x, B, H, W, C, r)

    @staticmethod
    def backward(ctx, dout):
        side_fence(dout.device)
        shape, r = ctx.geom
        dx = torch.empty(shape, dtype=dout.dtype, device=dout.device)
        ops.twins_subsample_bwd(_c(dout), dx, shape[0], shape[1], shape[2], shape[3], r)
        return dx, None


class SrAttentionFn(Function):
    """softmax(q k^T / 8) v for PVT's (reduced-key) attention on the q / kv projection outputs (pvt.py:51-63)."""

    @staticmethod
    def forward(ctx, q, kv, B, Lq, Lk, n_head, drop=None):
        q, kv = _c(q), _c(kv)
        o, lse = ops.srattn_fwd(q, kv, B, Lq, Lk, n_head, drop=drop)
        ctx.save_for_backward(q, kv, o, lse)
        ctx.geom, ctx.drop = (B, Lq, Lk, n_head), drop
        return o

    @staticmethod
    def backward(ctx, do):
        side_fence(do.device)
        q, kv, o, lse = ctx.saved_tensors
        dq, dkv = ops.srattn_bwd(q, kv, o, _c(do), lse, *ctx.geom, drop=ctx.drop)
        return dq, dkv, None, None, None, None, None


class _SrLayerPlan:
    """Shape-dependent part of the two descriptors of one PVT block / Twins global half (vtx_srlayer_fwd / bwd): buffer
    layouts, workspace sizes, prefilled structures -- the counterpart of _LayerPlan for csrc/layer.hip's second pair."""

    def __init__(self, m, B, L, C, ff, Lk, want_z, has_rs, rps, dp_c, has_srn, splitk):
        lib = _lib.load()
        es = 2
        M, r = B * L, m.reduction
        rows, K = B * Lk, r * r * C
        nH = m.n_head
        off = [0]

        def carve(nbytes):
            o = off[0]
            off[0] = (o + nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
            return o
        f = self.f_off = {k: carve(n * es) for k, n in (("ln1", M * C), ("q", M * C), ("kv", rows * 2 * C), ("o", M * C),
                                                        ("x1", M * C), ("ln2", M * C), ("h", M * ff))}
        f["z"] = carve(M * ff * es) if want_z else None
        f["patches"] = carve(rows * K * es) if r > 1 else None
        f["red"] = carve(rows * C * es) if r > 1 else None
        f["kvin"] = carve(rows * C * es) if r > 1 and has_srn else None
        f["patches_t"] = carve(rows * K * es) if splitk else None
        f["red32"] = carve(rows * C * 4) if splitk else None
        self.splitk_wsb = lib.vtx_wgrad_workspace(K, rows, C) if splitk else 0
        f["splitk_ws"] = carve(self.splitk_wsb) if splitk else None
        for k, n in (("mean1", M), ("rstd1", M), ("mean2", M), ("rstd2", M), ("lse", B * nH * L)):
            f[k] = carve(4 * n)
        f["means"] = carve(4 * rows) if r > 1 and has_srn else None
        f["rstds"] = carve(4 * rows) if r > 1 and has_srn else None
        self.f_bytes = off[0]
        off[0] = 0
        self.ln_wsb = lib.vtx_layernorm_bwd_workspace(M, C)
        self.lns_wsb = lib.vtx_layernorm_bwd_workspace(rows, C) if r > 1 and has_srn else 0
        self.attn_wsb = lib.vtx_srattn_bwd_workspace(B, L, Lk, nH, C // nH)
        na = 4 if r > 1 else 5
        Na, Ka = (ctypes.c_int * na)(*([C, ff, C, C] + ([2 * C] if r == 1 else []))), (ctypes.c_int * na)(*([ff, C, C, C] + ([C] if r == 1 else [])))
        self.wgrad_wsb = lib.vtx_wgrad_group_workspace(na, Na, Ka, M)
        ok = bool(lib.vtx_wgrad_group_ok(ops.BF16, na, Na, Ka, M, int(has_rs), int(rps), float(dp_c))) and na <= lib.vtx_wgrad_group_max()
        self.wgrad2_wsb = 0
        if r > 1:
            Nb, Kb = (ctypes.c_int * 2)(2 * C, C), (ctypes.c_int * 2)(C, K)
            self.wgrad2_wsb = lib.vtx_wgrad_group_workspace(2, Nb, Kb, rows)
            ok = ok and bool(lib.vtx_wgrad_group_ok(ops.BF16, 2, Nb, Kb, rows, 0, 1, 0.0))
        # both grouped weight-gradient launches of the backward must apply (bf16, LDS-DMA shapes); else: call-by-call path
        self.ok = ok
        b = self.b_off = {k: carve(n * es) for k, n in (("dz", M * ff), ("dln2", M * C), ("dx1", M * C), ("dout", M * C),
                                                        ("dq", M * C), ("dkv", rows * 2 * C), ("dkvin", rows * C), ("dln1", M * C))}
        b["dred"] = carve(rows * C * es) if r > 1 and has_srn else None
        b["dpatches"] = carve(rows * K * es) if r > 1 else None
        for k, n in (("ln1_ws", self.ln_wsb), ("ln2_ws", self.ln_wsb), ("lns_ws", self.lns_wsb), ("attn_ws", self.attn_wsb),
                     ("wgrad_ws", self.wgrad_wsb), ("wgrad2_ws", self.wgrad2_wsb)):
            b[k] = carve(n) if n else None
        self.b_bytes = off[0]
        common = dict(dtype=ops.BF16, twins=int(bool(m.twins)), M=M, C=C, ff=ff, nH=nH, L=L, B=B, H=m.height, W=m.width, r=r,
                      skip=m.skip, Lk=Lk)
        self.fwd = _lib.SrLayerFwd(eps=float(m.eps), splitk=int(bool(splitk)), splitk_ws_bytes=self.splitk_wsb, **common)
        self.bwd = _lib.SrLayerBwd(ln_ws_bytes=self.ln_wsb, lns_ws_bytes=self.lns_wsb, attn_ws_bytes=self.attn_wsb,
                                   wgrad_ws_bytes=self.wgrad_wsb, wgrad2_ws_bytes=self.wgrad2_wsb, **common)
        self.K, self.rows = K, rows


_sr_layer_plans = {}


def _sr_layer_plan(m, B, L, C, ff, Lk, want_z, has_rs, rps, dp_c, has_srn, splitk):
    key = (m.n_head, m.height, m.width, m.reduction, m.skip, m.eps, bool(m.twins), B, L, C, ff, Lk, want_z, has_rs, rps, dp_c,
           has_srn, splitk)
    pl = _sr_layer_plans.get(key)
    if pl is None:
        pl = _sr_layer_plans[key] = _SrLayerPlan(m, B, L, C, ff, Lk, want_z, has_rs, rps, dp_c, has_srn, splitk)
    return pl


class PvtLayerFn(Function):
    """One PVT block (reference models/pvt.py:31-68, 99-103):
         x1 = x  + s1 * proj(sr_attn(q(LN1 x), kv(reduce(LN1 x))))      y = x1 + s2 * fc2(silu(fc1(LN2 x1)))
    reduce = Conv2d(C, C, r, stride r) on the token grid + LayerNorm (reduction > 1) as patchify gather + GEMM + LN.
    The global half of a Twins-SVT layer (reference models/twins.py:56-93, 201-202) is the same block without that LayerNorm
    (srn_w = srn_b = None) and with head dim 32.
    Kernels forward: LN, GEMM(q), [gather, GEMM+bias, LN], GEMM(kv), attention, GEMM+residual, LN, GEMM+SiLU,
    GEMM+residual."""

    @staticmethod
    def forward(ctx, x, ln1_w, ln1_b, q_w, kv_w, sr_w, sr_b, srn_w, srn_b, proj_w, proj_b, ln2_w, ln2_b, fc1_w, fc1_b,
                fc2_w, fc2_b, s1, s2, dp_c, meta):
        x = _c(x)
        T = x.dtype
        B, L, C = x.shape
        m = meta
        rps = L
        r = m.reduction
        ctx.one_call = False
        if _LAYER_CALL and _DEFER_REDUCE and x.is_cuda and T == torch.bfloat16 and C % m.n_head == 0 and C // m.n_head in (32, 64):
            y = PvtLayerFn._forward_one_call(ctx, x, ln1_w, ln1_b, q_w, kv_w, sr_w, sr_b, srn_w, srn_b, proj_w, proj_b, ln2_w,
                                             ln2_b, fc1_w, fc1_b, fc2_w, fc2_b, s1, s2, dp_c, m)
            if y is not None:
                return y
        ln1, mean1, rstd1 = ops.layernorm_fwd(x, ln1_w.detach(), ln1_b.detach(), m.eps)
        wq, wkv, wo, w1, w2 = wcast(q_w, T), wcast(kv_w, T), wcast(proj_w, T), wcast(fc1_w, T), wcast(fc2_w, T)
        q = ops.gemm(ln1, wq[0], 0)
        wsr = patches = red = means = rstds = None
        if r > 1:
            if m.twins:                   # columns (c', py, px) = the weight's own layout: no permutation, the plan's transposed copy
                wsr_p = wcast(sr_w, T)
                wsr = (wsr_p[0].view(sr_w.shape[0], -1), wsr_p[1])
                Lk = (m.height // r) * (m.width // r)
                # few rows, long contraction (stage 3 / 4 of Twins-SVT-S: 512 / 128 rows x K = 12 544 / 25 088): as a plain GEMM
                # that is 8-32 workgroups for 100-310 us.  With the operand ALSO gathered transposed and the weight plan's
                # transposed copy it is dW-shaped work -- out[rows, C] = sum_k P^T[k, rows] W^T[k, C] -- for the split-K
                # weight-gradient launch, which fills the chip
                splitk = (_TWINS_SPLITK and T == torch.bfloat16 and wsr[1] is not None and B * Lk <= 512 and B * Lk >= 64 and
                          B * Lk % 8 == 0 and wsr[0].shape[1] >= 4096)
                if splitk:
                    patches, patches_t = ops.twins_subsample_fwd(ln1, B, m.height, m.width, C, r, transposed=True)
                    red32, _ = ops.wgrad(patches_t, wsr[1], want_bias=False)
                    red = ops.bias_cast(red32, sr_b.detach(), T)
                else:
                    patches = ops.twins_subsample_fwd(ln1, B, m.height, m.width, C, r)
            else:
                wsr = conv_as_rows(wcast(sr_w, T)[0])
                patches = ops.patchify_fwd(ln1, B, m.height, m.width, C, r, m.skip)
                splitk = False
            Lk = (m.height // r) * (m.width // r)
            if not splitk:
                red = ops.gemm(patches, wsr[0], 0, bias=sr_b.detach())
            if srn_w is not None:
                kvin, means, rstds = ops.layernorm_fwd(red, srn_w.detach(), srn_b.detach(), m.eps)
            else:                                                     # Twins-SVT: the sub-sampled tokens go straight to kv
                kvin = red
        else:
            kvin, Lk = ln1, L
        kv = ops.gemm(kvin, wkv[0], 0)
        o, lse = ops.srattn_fwd(q, kv, B, L, Lk, m.n_head)
        x1 = ops.gemm(o, wo[0], 0, bias=proj_b.detach(), resid=x, rowscale=s1, rows_per_scale=rps)
        ln2, mean2, rstd2 = ops.layernorm_fwd(x1, ln2_w.detach(), ln2_b.detach(), m.eps)
        h, z = _act_gemm(ctx, ln2, w1[0], fc1_b.detach(), ACT_SILU)
        y = ops.gemm(h, w2[0], 0, bias=fc2_b.detach(), resid=x1, rowscale=s2, rows_per_scale=rps)
        ctx.save_for_backward(x, ln1_w, ln2_w, srn_w, mean1, rstd1, ln1, q, kv, o, lse, x1, mean2, rstd2, ln2, z, h,
                              patches, red, means, rstds, kvin if r > 1 else None, s1, s2)
        ctx.wp = (wq, wkv, wo, w1, w2, wsr)
        ctx.meta, ctx.rps, ctx.dp_c, ctx.Lk, ctx.sr_shape = m, rps, float(dp_c), Lk, (None if sr_w is None else sr_w.shape)
        return y.view(B, L, C)

    # ---- the same block through ONE C call each way (csrc/layer.hip vtx_srlayer_fwd / bwd): the launches of the code
    #      below in its order with its arguments (bit-identical), one ctypes call and one activation buffer per layer instead of
    #      ~25 calls and ~40 tensor allocations: PVT-Small's host path was 14-15 ms per step against 13 ms of GPU time
    @staticmethod
    def _forward_one_call(ctx, x, ln1_w, ln1_b, q_w, kv_w, sr_w, sr_b, srn_w, srn_b, proj_w, proj_b, ln2_w, ln2_b, fc1_w, fc1_b,
                          fc2_w, fc2_b, s1, s2, dp_c, m):
        T = x.dtype
        B, L, C = x.shape
        r, ff = m.reduction, fc1_w.shape[0]
        Lk = (m.height // r) * (m.width // r) if r > 1 else L
        has_srn = srn_w is not None
        wq, wkv, wo, w1, w2 = wcast(q_w, T), wcast(kv_w, T), wcast(proj_w, T), wcast(fc1_w, T), wcast(fc2_w, T)
        wsr = None
        splitk = False
        if r > 1:
            if m.twins:
                wsr_p = wcast(sr_w, T)
                wsr = (wsr_p[0].view(sr_w.shape[0], -1), wsr_p[1])
                splitk = (_TWINS_SPLITK and wsr[1] is not None and 64 <= B * Lk <= 512 and B * Lk % 8 == 0 and
                          wsr[0].shape[1] >= 4096)
            else:
                wsr = conv_as_rows(wcast(sr_w, T)[0])
        want_z = any(ctx.needs_input_grad)
        pl = _sr_layer_plan(m, B, L, C, ff, Lk, want_z, s1 is not None or s2 is not None, L, float(dp_c), has_srn, splitk)
        if not pl.ok:
            return None
        _check_layer_inputs(x, T, ln1_w, ln1_b, ln2_w, ln2_b, proj_b, fc1_b, fc2_b, sr_b, srn_w, srn_b)
        buf = torch.empty(pl.f_bytes, dtype=torch.uint8, device=x.device)
        y = torch.empty_like(x)
        base, fo = buf.data_ptr(), pl.f_off
        at = lambda k: None if fo[k] is None else base + fo[k]
        d = _copy_desc(pl.fwd)
        d.rows_per_scale = L
        d.x, d.y = x.data_ptr(), y.data_ptr()
        d.ln1_w, d.ln1_b, d.ln2_w, d.ln2_b = ln1_w.data_ptr(), ln1_b.data_ptr(), ln2_w.data_ptr(), ln2_b.data_ptr()
        d.srn_w, d.srn_b = _dp(srn_w), _dp(srn_b)
        d.wq, d.wkv, d.wo, d.w1, d.w2 = wq[0].data_ptr(), wkv[0].data_ptr(), wo[0].data_ptr(), w1[0].data_ptr(), w2[0].data_ptr()
        if r > 1:
            d.wsr, d.wsr_t, d.bsr = wsr[0].data_ptr(), _dp(wsr[1]) if splitk else None, sr_b.data_ptr()
        d.bo, d.b1, d.b2 = proj_b.data_ptr(), fc1_b.data_ptr(), fc2_b.data_ptr()
        d.s1, d.s2 = _dp(s1), _dp(s2)
        for k in ("ln1", "q", "patches", "patches_t", "red32", "red", "kvin", "kv", "o", "x1", "ln2", "z", "h", "mean1", "rstd1",
                  "mean2", "rstd2", "means", "rstds", "lse", "splitk_ws"):
            setattr(d, k, at(k))
        _lib.check(_lib.load().vtx_srlayer_fwd(ctypes.byref(d), ops._stream()), "vtx_srlayer_fwd")
        ctx.save_for_backward(x, buf, ln1_w, ln2_w, srn_w, s1, s2, fc1_b)
        ctx.wp = (wq, wkv, wo, w1, w2, wsr)
        ctx.plan, ctx.one_call = pl, True
        ctx.meta, ctx.rps, ctx.dp_c, ctx.Lk, ctx.sr_shape = m, L, float(dp_c), Lk, (None if sr_w is None else sr_w.shape)
        ctx.shapes = (q_w.shape, kv_w.shape, proj_w.shape, fc1_w.shape, fc2_w.shape)
        return y.view(B, L, C)

    @staticmethod
    def _backward_one_call(ctx, dy):
        x, buf, ln1_w, ln2_w, srn_w, s1, s2, fc1_b = ctx.saved_tensors
        pl, m = ctx.plan, ctx.meta
        if pl.f_off["z"] is None:
            raise VtxError("vtx: this layer's forward ran without a graph (no pre-activation was kept)")
        wq, wkv, wo, w1, w2, wsr = ctx.wp
        dy = _c(dy)
        dev = x.device
        B, L, C = x.shape
        r = m.reduction
        T = x.dtype
        has_srn = srn_w is not None
        # transposed weight operands exactly where vtx.functional.dgrad would use them (LDS-DMA shapes); made now when the weight
        # plan has none (the PVT reduction conv's permuted rows)
        def tr(wp):
            w2d = wp[0].view(wp[0].shape[0], -1)
            if not ops.glds_ok(w2d.shape[1], w2d.shape[0]):
                return None
            return wp[1] if wp[1] is not None else w2d.t().contiguous()
        wqt, wkvt, wot, w1t, w2t = tr(wq), tr(wkv), tr(wo), tr(w1), tr(w2)
        wsrt = tr(wsr) if r > 1 else None
        scratch = torch.empty(pl.b_bytes, dtype=torch.uint8, device=dev)
        dx = torch.empty_like(x)
        f32 = dict(dtype=torch.float32, device=dev)
        qs, kvs, os_, f1s, f2s = ctx.shapes
        dWq, dWkv, dWo, dW1, dW2 = (torch.empty(sh, **f32) for sh in (qs, kvs, os_, f1s, f2s))
        ff = f1s[0]
        small = torch.empty(6 * C + ff + (3 * C if r > 1 else 0), **f32).split((C, C, C, C, C, C, ff) + ((C, C, C) if r > 1 else ()))
        dg1, dbe1, dbo, dg2, dbe2, db2, db1 = small[:7]
        dbsr, dgs, dbs = (small[7], small[8], small[9]) if r > 1 else (None, None, None)
        dWsr = torch.empty((C, pl.K), **f32) if r > 1 else None
        base, fo, sb, bo = buf.data_ptr(), pl.f_off, scratch.data_ptr(), pl.b_off
        fa = lambda k: None if fo[k] is None else base + fo[k]
        ba = lambda k: None if bo[k] is None else sb + bo[k]
        d = _copy_desc(pl.bwd)
        d.rows_per_scale, d.scale_const = ctx.rps, ctx.dp_c
        d.dy, d.x, d.dx = dy.data_ptr(), x.data_ptr(), dx.data_ptr()
        for k in ("ln1", "q", "patches", "red", "kv", "o", "x1", "ln2", "z", "h", "mean1", "rstd1", "mean2", "rstd2", "means",
                  "rstds", "lse"):
            setattr(d, k, fa(k))
        d.kvin = fa("kvin") if (r > 1 and has_srn) else (fa("red") if r > 1 else fa("ln1"))
        d.ln1_w, d.ln2_w, d.srn_w = ln1_w.data_ptr(), ln2_w.data_ptr(), _dp(srn_w)
        d.b1 = fc1_b.data_ptr()                     # (the fused MLP's backward recomputes z)
        d.wq, d.wkv, d.wo, d.w1, d.w2 = wq[0].data_ptr(), wkv[0].data_ptr(), wo[0].data_ptr(), w1[0].data_ptr(), w2[0].data_ptr()
        d.wqt, d.wkvt, d.wot, d.w1t, d.w2t = _dp(wqt), _dp(wkvt), _dp(wot), _dp(w1t), _dp(w2t)
        if r > 1:
            d.wsr, d.wsrt = wsr[0].data_ptr(), _dp(wsrt)
        d.s1, d.s2 = _dp(s1), _dp(s2)
        for k in ("dz", "dln2", "dx1", "dout", "dq", "dkv", "dkvin", "dred", "dpatches", "dln1", "ln1_ws", "ln2_ws", "lns_ws",
                  "attn_ws", "wgrad_ws", "wgrad2_ws"):
            setattr(d, k, ba(k))
        d.dWq, d.dWkv, d.dWo, d.dbo, d.dW1, d.db1, d.dW2, d.db2 = (dWq.data_ptr(), dWkv.data_ptr(), dWo.data_ptr(), dbo.data_ptr(),
                                                                   dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr())
        d.dg1, d.dbe1, d.dg2, d.dbe2 = dg1.data_ptr(), dbe1.data_ptr(), dg2.data_ptr(), dbe2.data_ptr()
        if r > 1:
            d.dWsr, d.dbsr = dWsr.data_ptr(), dbsr.data_ptr()
            if has_srn:
                d.dgs, d.dbs = dgs.data_ptr(), dbs.data_ptr()
        side = st = None
        if _deferred and not ops.timing():
            st = _side_states.get(dev)
            if st is None:
                st = _side_states[dev] = _SideState(dev)
            side = st.stream.cuda_stream
            st.keep.append((scratch, buf, dy, x, s1, s2, ctx.wp, (wqt, wkvt, wot, w1t, w2t, wsrt)))
            st.pending = True
        _lib.check(_lib.load().vtx_srlayer_bwd(ctypes.byref(d), ops._stream(), side), "vtx_srlayer_bwd")
        dWsr_out = None
        if r > 1:
            co, _, pp, _ = ctx.sr_shape
            if m.twins:
                dWsr_out = dWsr.view(co, C, pp, pp)                            # already the parameter's layout
            elif st is not None:                                               # (py, px, c) columns back to (c, py, px): after the
                with torch.cuda.stream(st.stream):                             # weight-gradient launch, on its stream
                    dWsr_out = dWsr.view(co, pp, pp, C).permute(0, 3, 1, 2).contiguous()
                st.keep.append((dWsr,))
            else:
                dWsr_out = dWsr.view(co, pp, pp, C).permute(0, 3, 1, 2).contiguous()
        return (dx.view(B, L, C), dg1, dbe1, dWq, dWkv, dWsr_out, dbsr, dgs if has_srn else None, dbs if has_srn else None, dWo,
                dbo, dg2, dbe2, dW1, db1, dW2, db2, None, None, None, None)

    @staticmethod
    def backward(ctx, dy):
        if ctx.one_call:
            return PvtLayerFn._backward_one_call(ctx, dy)
        (x, ln1_w, ln2_w, srn_w, mean1, rstd1, ln1, q, kv, o, lse, x1, mean2, rstd2, ln2, z, h, patches, red, means,
         rstds, kvin, s1, s2) = ctx.saved_tensors
        wq, wkv, wo, w1, w2, wsr = ctx.wp
        m, rps, dp_c, Lk = ctx.meta, ctx.rps, ctx.dp_c, ctx.Lk
        T = x.dtype
        B, L, C = x.shape
        r = m.reduction
        dy = _c(dy)
        # ---- MLP branch
        dz = dgrad(dy, w2, T, act=ACT_DSILU, aux_in=z, rowscale=s2, rows_per_scale=rps)
        dln2 = dgrad(dz, w1, T)
        # (round 3: the layer's column reductions -- LayerNorm dgamma / dbeta of LN2, LN1 and the reduction's LayerNorm -- ride in
        #  the reduce launch of the grouped weight gradients, like the Swin / ViT layers: -3 launches per layer, off the main stream)
        if _DEFER_REDUCE:
            dx1, part2 = ops.layernorm_bwd(dln2, x1, mean2, rstd2, ln2_w.detach(), dres=dy, defer=True)
        else:
            dx1, dg2, dbe2 = ops.layernorm_bwd(dln2, x1, mean2, rstd2, ln2_w.detach(), dres=dy)
        # ---- attention branch
        do = dgrad(dx1, wo, T, rowscale=s1, rows_per_scale=rps)
        dq, dkv = ops.srattn_bwd(q, kv, o, do, lse, B, L, Lk, m.n_head)
        dWsr = dbsr = dgs = dbs = None
        jobs = [(dy, h, True, s2), (dz, ln2, True, None), (dx1, o, True, s1), (dq, ln1, False, None)]
        if r > 1:
            dkvin = dgrad(dkv, wkv, T)
            part_s = None
            if srn_w is None:
                dred = dkvin
            elif _DEFER_REDUCE:
                dred, part_s = ops.layernorm_bwd(dkvin, red, means, rstds, srn_w.detach(), defer=True)
            else:
                dred, dgs, dbs = ops.layernorm_bwd(dkvin, red, means, rstds, srn_w.detach())
            dpatches = dgrad(dred, wsr, T)
            dln1 = dgrad(dq, wq, T)
            if m.twins:
                ops.twins_subsample_bwd(dpatches, dln1, B, m.height, m.width, C, r, accumulate=True)
            else:
                ops.patchify_bwd(dpatches, dln1, B, m.height, m.width, C, r, m.skip, accumulate=True)
            co, _, pp, _ = ctx.sr_shape

            def unpermute(res):                                        # (py, px, c) columns back to (c, py, px)
                (a, _), (b, bb) = res
                if m.twins:                                            # already the parameter's layout
                    return a, b.view(co, C, pp, pp), bb
                return a, b.view(co, pp, pp, C).permute(0, 3, 1, 2).contiguous(), bb
            dWkv, dWsr, dbsr = layer_wgrads([(dkv, kvin, False, None), (dred, patches, True, None)], post=unpermute)
        else:
            jobs.append((dkv, ln1, False, None))
            dkvin = dgrad(dkv, wkv, T)
            dln1 = dgrad(dq, wq, T, resid=dkvin)                       # both consumers of LN1's output
        if _DEFER_REDUCE:
            dx, part1 = ops.layernorm_bwd(dln1, x, mean1, rstd1, ln1_w.detach(), dres=dx1, defer=True)
            parts = [part2, part1] + ([part_s] if r > 1 and part_s is not None else [])
            res, red_out = layer_wgrads(jobs, rps, dp_c, colparts=parts)
            (dg2, dbe2), (dg1, dbe1) = red_out[0], red_out[1]
            if r > 1 and part_s is not None:
                dgs, dbs = red_out[2]
        else:
            dx, dg1, dbe1 = ops.layernorm_bwd(dln1, x, mean1, rstd1, ln1_w.detach(), dres=dx1)
            res = layer_wgrads(jobs, rps, dp_c)
        (dW2, db2), (dW1, db1), (dWo, dbo), (dWq, _) = res[:4]
        if r == 1:
            dWkv = res[4][0]
        return (dx.view(B, L, C), dg1, dbe1, dWq, dWkv, dWsr, dbsr, dgs, dbs, dWo, dbo, dg2, dbe2, dW1, db1, dW2, db2,
                None, None, None, None)


class PegFn(Function):
    """Positional-encoding generator of Twins-SVT (reference models/twins.py:25-37): y = x + DepthwiseConv3x3(x) on the
    channels-last feature map, one kernel forward; backward = the mirrored-tap kernel on dy + the weight gradient."""

    @staticmethod
    def forward(ctx, x, w):
        x = _c(x)
        ctx.save_for_backward(x, w)
        return ops.dwconv3_fwd(x, w.detach())

    @staticmethod
    def backward(ctx, dy):
        side_fence(dy.device)
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dx = ops.dwconv3_fwd(dy, w.detach(), adjoint=True) if ctx.needs_input_grad[0] else None
        dw = ops.dwconv3_wgrad(x, dy).view(w.shape) if ctx.needs_input_grad[1] else None
        return dx, dw


class PvtPatchEmbedFn(Function):
    """pvt.PatchEmbedding (reference models/pvt.py:126-140): Conv2d(in, dim, p, stride p) -> LayerNorm(1e-6) ->
    [cls token] + pos.  Stage 1 reads the NCHW image (im2col gather in conv-weight order); later stages read the
    previous stage's token-major features [B, skip + H*W, C] (patchify gather, weight columns permuted to match)."""

    @staticmethod
    def forward(ctx, x, w, b, ln_w, ln_b, cls, pos, patch, grid, skip, eps, dtype):
        T = dtype
        out_ch = w.shape[0]
        if x.dim() == 4:                                               # NCHW image
            B = x.shape[0]
            H, W = x.shape[2] // patch, x.shape[3] // patch
            patches = ops.patch_gather(_img(x), patch, 1, T).view(B * H * W, -1)
            wp = (wcast(w, T)[0].view(out_ch, -1), None)
            ctx.tok = None
        else:
            x = _c(x)
            B, _, C = x.shape
            H, W = grid[0] // patch, grid[1] // patch
            patches = ops.patchify_fwd(x, B, grid[0], grid[1], C, patch, skip)
            wp = conv_as_rows(wcast(w, T)[0])
            ctx.tok = (x.shape, grid, skip)
        t = ops.gemm(patches, wp[0], 0, bias=b.detach())
        tn, mean, rstd = ops.layernorm_fwd(t, ln_w.detach(), ln_b.detach(), eps)
        out = ops.add_pos_fwd(tn.view(B, H * W, out_ch), None if cls is None else _c(cls.detach()), _c(pos.detach()))
        ctx.save_for_backward(patches, t, mean, rstd, ln_w)
        ctx.wp, ctx.patch, ctx.has_cls, ctx.wshape = wp, patch, cls is not None, w.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        side_fence(dout.device)
        patches, t, mean, rstd, ln_w = ctx.saved_tensors
        dtn, dcls, dpos = ops.add_pos_bwd(_c(dout), ctx.has_cls)
        dt, dg, db = ops.layernorm_bwd(dtn.view(-1, dtn.shape[-1]), t, mean, rstd, ln_w.detach())
        dW, dbias = ops.wgrad(dt, patches)
        co, ci, p, _ = ctx.wshape
        dx = None
        if ctx.tok is None:
            dW = dW.view(ctx.wshape)
        else:
            dW = dW.view(co, p, p, ci).permute(0, 3, 1, 2).contiguous()
            if ctx.needs_input_grad[0]:
                shape, grid, skip = ctx.tok
                dpatches = dgrad(dt, ctx.wp, dt.dtype)
                dx = torch.empty(shape, dtype=dt.dtype, device=dt.device)
                if skip:
                    dx[:, :skip].zero_()
                ops.patchify_bwd(dpatches, dx, shape[0], grid[0], grid[1], shape[2], p, skip)
        return dx, dW, dbias, dg, db, dcls, dpos, None, None, None, None, None
