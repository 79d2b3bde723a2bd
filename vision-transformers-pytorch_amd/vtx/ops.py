"""Thin tensor-level wrappers over the C ABI (one Python function per vtx_* entry point).

Every function takes contiguous DEVICE tensors, allocates outputs / workspaces through PyTorch's
caching allocator, enqueues on torch's current HIP stream and returns tensors.  No fallback: CPU
tensors, wrong dtypes or a missing library raise VtxError.
"""
import ctypes
import os

import torch

from . import _lib, options
from ._lib import BF16, F32, VtxError, check


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise VtxError(f"vtx: unsupported dtype {t.dtype} (float32 or bfloat16)")


def _dev(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise VtxError("vtx: the HIP path needs device tensors (there is no CPU fallback); "
                           "move the module / inputs to 'cuda'")
        if not t.is_contiguous():
            raise VtxError("vtx: tensor must be contiguous")
        if t.numel() == 0:
            raise VtxError("vtx: empty tensor (batch of zero samples?) -- the HIP kernels take at least one row")


def _p(t):
    return None if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  Called once per kernel launch (~500 times per Swin-S
    step): the two raw C accessors cost ~0.3 us; torch.cuda.current_stream().cuda_stream builds a Stream object through
    four Python frames (~9 us, 1.7 ms of host time per step -- tools/probe/host_profile.py)."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return _RAW_STREAM(_RAW_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def _f32(t, what):
    if t is not None and t.dtype != torch.float32:
        raise VtxError(f"vtx: {what} must be float32")


# ------------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x, gamma, beta, eps, merge_hw=None):
    """y, mean, rstd.  merge_hw=(H, W): x is (B,H,W,Cs) and rows are PatchMerge's 2x2 gathers (C=4Cs)."""
    _dev(x, gamma, beta)
    _f32(gamma, "gamma"); _f32(beta, "beta")
    lib = _lib.load()
    if merge_hw is None:
        C = x.shape[-1]
        rows = x.numel() // C
        y = torch.empty_like(x)
        merge, H, W = 0, 0, 0
    else:
        H, W = merge_hw
        B, Cs = x.shape[0], x.shape[-1]
        C = 4 * Cs
        rows = B * (H // 2) * (W // 2)
        y = torch.empty((B, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
        merge = 1
    if gamma.numel() != C or beta.numel() != C:
        raise VtxError("vtx: layernorm weight/bias size mismatch")
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    with _timed("ln_fwd_kernel", 0.0, 2.0 * rows * C * x.element_size() + 8.0 * rows):
        check(lib.vtx_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, C, float(eps),
                                    _dt(x), merge, H, W, _stream()), "vtx_layernorm_fwd")
    return y, mean, rstd


class Partials:
    """Per-block partial sums left in a kernel's workspace, to be column-reduced later (colreduce_multi): `ws` keeps
    the memory alive; outputs = (out0 [C], out1 [C] or None)."""
    __slots__ = ("ws", "nb", "C", "ld", "two")

    def __init__(self, ws, nb, C, ld, two):
        self.ws, self.nb, self.C, self.ld, self.two = ws, nb, C, ld, two


def colreduce_multi(parts):
    """One launch for up to 4 deferred reductions (a layer's LayerNorm dgamma / dbeta pairs and its rel_pos gradient):
    -> [(out0, out1 or None)] in order.  Fixed summation order: the same bits as the kernels' own reductions."""
    n = len(parts)
    dev = parts[0].ws.device
    outs = [(torch.empty(p.C, dtype=torch.float32, device=dev),
             torch.empty(p.C, dtype=torch.float32, device=dev) if p.two else None) for p in parts]
    vp = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
    ia = lambda xs: (ctypes.c_int * n)(*xs)
    with _timed("colreduce_multi_kernel", 0.0, sum(4.0 * p.nb * p.ld for p in parts)):
        check(_lib.load().vtx_colreduce_multi(n, vp([p.ws for p in parts]), vp([o[0] for o in outs]), vp([o[1] for o in outs]),
                                              ia([p.nb for p in parts]), ia([p.C for p in parts]), ia([p.ld for p in parts]),
                                              _stream()), "vtx_colreduce_multi")
    return outs


def layernorm_bwd(dy, x, mean, rstd, gamma, dres=None, merge_hw=None, defer=False):
    """dx (= dres + LN'(dy)), dgamma, dbeta -- or with ``defer`` (dx, Partials): the dgamma / dbeta column reduce is left
    to a later colreduce_multi (one launch per layer instead of one per LayerNorm)."""
    _dev(dy, x, mean, rstd, gamma, dres)
    lib = _lib.load()
    C = dy.shape[-1]
    rows = dy.numel() // C
    if merge_hw is None:
        merge, H, W = 0, 0, 0
    else:
        merge, (H, W) = 1, merge_hw
    dx = torch.empty_like(x)
    dgamma = None if defer else torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = None if defer else torch.empty(C, dtype=torch.float32, device=x.device)
    wsb = lib.vtx_layernorm_bwd_workspace(rows, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    with _timed("ln_bwd_kernel" + ("" if defer else " (+colreduce)"), 0.0,
                (3.0 + (dres is not None)) * rows * C * x.element_size() + 8.0 * rows):
        check(lib.vtx_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dgamma),
                                    _p(dbeta), _p(ws), wsb, rows, C, _dt(x), merge, H, W, _stream()),
              "vtx_layernorm_bwd")
    if defer:
        return dx, Partials(ws, lib.vtx_layernorm_bwd_blocks(rows, C), C, 2 * C, True)
    return dx, dgamma, dbeta


# ------------------------------------------------------------------------------- kernel timing hook
class KernelTimer:
    """Optional per-launch HIP-event timing of the GEMM kernels (used by bench.py for the roofline).

    Events are recorded on torch's current stream = the stream the kernel is launched on.  Each record is
    (kernel name as rocprofv3 prints it, algorithmic FLOPs of the launch, start event, end event).
    """

    def __init__(self):
        self.records = []
        self.extra = []            # (name, flops, bytes, ms): launches timed inside libvtx (vtx_timer_*: the one-call layers)
        self.shapes = []           # per `extra` record: "rows x n x k flags f" of the launch (tools: per-shape tables)

    def bracket(self, name, flops, nbytes=0.0):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        self.records.append((name, flops, nbytes, e0, e1))
        return e0, e1

    def summary(self):
        """{kernel: dict(launches, flops, bytes, ms)} -- call after torch.cuda.synchronize().  bytes = algorithmic HBM
        bytes (every operand read once, every output written once)."""
        out = {}
        for name, flops, nbytes, e0, e1 in self.records:
            d = out.setdefault(name, dict(launches=0, flops=0.0, bytes=0.0, ms=0.0))
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += e0.elapsed_time(e1)
        for name, flops, nbytes, ms in self.extra:
            d = out.setdefault(name, dict(launches=0, flops=0.0, bytes=0.0, ms=0.0))
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += ms
        return out

    def by_shape(self):
        """{(kernel, shape string): dict(launches, flops, bytes, ms)} of the launches timed inside libvtx."""
        out = {}
        for (name, flops, nbytes, ms), shape in zip(self.extra, self.shapes):
            d = out.setdefault((name, shape), dict(launches=0, flops=0.0, bytes=0.0, ms=0.0))
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += ms
        return out


_timer = None


class _Timed:
    """``with _timed(name, flops, bytes):`` -- HIP events around the launches inside, on torch's CURRENT stream (= the
    stream the kernels are enqueued on), when a KernelTimer is installed; free otherwise."""
    __slots__ = ("ev",)

    def __init__(self, ev):
        self.ev = ev

    def __enter__(self):
        if self.ev is not None:
            self.ev[0].record()
        return self

    def __exit__(self, *exc):
        if self.ev is not None:
            self.ev[1].record()
        return False


_NOT_TIMED = _Timed(None)


def _timed(name, flops=0.0, nbytes=0.0):
    if _timer is None:
        return _NOT_TIMED
    return _Timed(_timer.bracket(name, float(flops), float(nbytes)))


def timing():
    """True while a KernelTimer is installed (bench.py's sampled steps run single-stream so that per-kernel durations
    stay attributable, see functional.side_wgrad)."""
    return _timer is not None


def _attn_bracket(name, nprob, L, D, rows, hd, es, bwd):
    """HIP-event bracket of an attention launch: algorithmic FLOPs 2*2*L^2*D per problem forward (QK^T, PV), 5 products
    backward (S, dP, dV, dQ, dK); algorithmic bytes q,k,v read + o written (forward) / q,k,v,o,do read + dq,dk,dv
    written (backward) -- scores never touch HBM."""
    if _timer is None:
        return None
    flops = (10.0 if bwd else 4.0) * nprob * L * L * D
    nbytes = (8.0 if bwd else 4.0) * rows * hd * es
    ev = _timer.bracket(name, flops, nbytes)
    ev[0].record()
    return ev


def set_kernel_timer(timer):
    """Install / remove the per-launch timer.  The launches of the one-call layers (vtx_layer_fwd / _bwd) are timed inside
    libvtx (vtx_timer_start / _stop: HIP events on the launch stream around every launch of a layer call); removing the
    timer waits for those events and files them under the names rocprofv3 prints for the kernels."""
    global _timer
    lib = _lib.load()
    if _timer is not None and timer is not _timer:
        cap = 16384
        buf = (_lib.TimerRec * cap)()
        n = lib.vtx_timer_stop(buf, cap)
        for i in range(n):
            r = buf[i]
            _timer.extra.append(_describe_timer_rec(r))
            _timer.shapes.append(f"{r.rows} x {r.n} x {r.k} flags {r.flags & 255}")
    if timer is not None and timer is not _timer:
        lib.vtx_timer_start()
    _timer = timer


def _describe_timer_rec(r):
    """(kernel name as rocprofv3 prints it, algorithmic FLOPs, algorithmic HBM bytes, ms) of one libvtx timer record."""
    fl, rows, n, k = r.flags, r.rows, r.n, r.k
    bf = bool(fl & 32)
    es = 2 if bf else 4
    dt = torch.bfloat16 if bf else torch.float32
    tn = "__bf16" if bf else "float"
    mapped = "true" if fl & 8 else "false"
    D = fl >> 8
    if r.tag == 2:                                                      # GEMM: C[rows, n] over k
        name = gemm_kernel_name(dt, n, 0, K=k, M=rows, mapped=bool(fl & 8), vec=bool(fl & 5))
        nb = es * (rows * k + n * k + rows * n * (1 + bool(fl & 1) + bool(fl & 2) + bool(fl & 4)))
        return name, 2.0 * rows * n * k, float(nb), r.ms
    if r.tag == 1:
        return "ln_fwd_kernel", 0.0, 2.0 * rows * n * es + 8.0 * rows, r.ms
    if r.tag == 5:
        return "ln_bwd_kernel", 0.0, 4.0 * rows * n * es + 8.0 * rows, r.ms
    if r.tag in (3, 6):                                                 # window attention: n heads of 32, k tokens per window
        bwd = r.tag == 6
        nprob = rows // k * n
        name = wattn_bwd_kernel_name(dt, bool(fl & 16)) if bwd else wattn_fwd_kernel_name(dt, bool(fl & 16), rows // k)
        return name, (10.0 if bwd else 4.0) * nprob * k * k * 32, (8.0 if bwd else 4.0) * rows * n * 32 * es, r.ms
    if r.tag in (4, 7):                                                 # global attention: n heads of D, k tokens per image
        bwd = r.tag == 7
        nprob = rows // k * n
        nkt = 4 if k <= 64 else (8 if k <= 128 else 14)
        fast = bf and D == 64 and k <= 224 and options.get("SATTN")
        name = (f"sattn_{'bwd' if bwd else 'fwd'}_kernel<{nkt}, {_sattn_cfg(k)}>" if fast else
                ("lattn_*_kernel" if k > 224 else f"attn_{'bwd' if bwd else 'fwd'}_kernel"))
        return name, (10.0 if bwd else 4.0) * nprob * k * k * D, (8.0 if bwd else 4.0) * rows * n * D * es, r.ms
    if r.tag == 8:                                                      # the layer's grouped weight gradient: n = C, k = ff
        C, ff = n, k
        pairs = ((C, ff), (ff, C), (C, C), (3 * C, C))
        name = wgrad_group_kernel_name(pairs, mapped) + " (+split-K and column reduce)"
        return (name, sum(2.0 * rows * a * b for a, b in pairs),
                sum(2.0 * rows * (a + b) + 4.0 * a * b for a, b in pairs), r.ms)
    if r.tag in (9, 10):                                                # sub-sampled attention: rows queries, n heads of D, k keys
        bwd = r.tag == 10
        return (f"srattn_{'bwd' if bwd else 'fwd'}_kernel<{tn}, {D}>", (10.0 if bwd else 4.0) * rows * n * k * D,
                (4.0 if bwd else 2.0) * rows * n * D * es, r.ms)
    if r.tag in (11, 12, 13):                                           # weight gradients of the PVT / Twins blocks
        if r.tag == 11:
            C, ff = n, k
            pairs = ((C, ff), (ff, C), (C, C), (C, C)) + (((2 * C, C),) if fl & 1 else ())
            name = wgrad_group_kernel_name(pairs) + " (+split-K and column reduce)"
        elif r.tag == 12:
            pairs = ((2 * n, n), (n, k))
            name = wgrad_group_kernel_name(pairs) + " (+split-K reduce)"
        else:
            pairs = ((n, k),)
            name = wgrad_kernel_name(dt, n, k, True) + " (+split-K reduce)"
        return (name, sum(2.0 * rows * a * b for a, b in pairs), sum(es * rows * (a + b) + 4.0 * a * b for a, b in pairs), r.ms)
    if r.tag == 14:                                                     # operand gather / scatter: rows x n elements in and out
        kern = "patchify_kernel" if not fl & 64 else "twins_subsample_kernel"
        return kern, 0.0, (3.0 if fl & 1 else 2.0) * rows * n * es, r.ms
    if r.tag in (15, 16):                                               # fused MLP of the narrow stages: rows x C (n), ff = k
        bwd = r.tag == 16
        code = options.get("MLP_FUSED")
        code = (code // 100 if bwd else code % 100) if code >= 100 else (6 if bwd else 12)         # mlp_fused.hip: mf_fwd_code / mf_bwd_code
        targs = ({4: "4, true, false", 8: "8, true, false", 9: "8, false, false", 12: "12, false, false", 16: "16, false, false"} if not bwd else
                 {4: "4, true, false, 0, false", 5: "4, true, true, 0, false", 6: "4, true, false, 0, true", 7: "8, true, false, 0, false",
                  8: "8, false, false, 0, false", 9: "8, false, true, 0, false"}).get(code, str(code))
        waves = targs
        lnb = bwd and bool(fl & 1)          # (flag 1 on a backward record: the norm_ff backward folded into the epilogue, option LN_FOLD)
        if not bwd and fl & 2:              # (flag 2 on a forward record: norm_ff on the row operands -- x1 in, ln2 and y out: 3 units as well)
            waves = "12, false, false, true"
        if lnb:
            waves = ("4, true, false, 0, true, true" if k % 64 == 0 else "4, true, false, 0, false, true")
        # forward: ln2, x1 in, y out (2 products); backward: ln2, dy in, dln2, h, dz out (z and dh recomputed: 3 products); with the
        # LayerNorm backward folded in: ln2, dy, x1 in, dx1, h, dz out (4 C + 2 ff per row) -- the two launches it replaces move 7 C + 2 ff
        return (f"mlp_{'bwd' if bwd else 'fwd'}_kernel<{n // 32}, {waves}>", (6.0 if bwd else 4.0) * rows * n * k,
                float(es * rows * (((4 if lnb else 3) * n + 2 * k) if bwd else 3 * n) + 2 * es * n * k), r.ms)
    if r.tag == 17:                                                     # dgrad + LayerNorm backward in one launch: rows x C (n) over k
        # dy [rows, k] and the weight in; x and the residual-stream gradient in, dx out (the dln tensor is never stored: the two launches
        # it replaces move rows x (k + 5 n) elements)
        return (f"dgrad_ln_kernel<{k // 32}, {n // 32}, 4>", 2.0 * rows * n * k, float(es * (rows * (k + 3 * n) + n * k) + 8.0 * rows), r.ms)
    if r.tag == 18:                                                     # LayerNorm forward on the row operands of a streaming GEMM: rows x n over k = C
        return (f"gemm_skinny_kernel<{k // 32}, 0, false, 4, true>", 2.0 * rows * n * k, float(es * (rows * (2 * k + n) + n * k) + 8.0 * rows), r.ms)
    return f"vtx_layer launch (tag {r.tag})", 0.0, 0.0, r.ms


def wattn_fwd_kernel_name(dtype, masked, nbn):
    """Mirrors wattn_fwd_launch (attention_win.hip): bf16 with >= 4 096 (image, window) problems per head takes the four-wave kernel
    (option WATTN_FWD4: 1 | 2 always | 0 never)."""
    m = "true" if masked else "false"
    f4 = options.get("WATTN_FWD4")
    if dtype == torch.bfloat16 and (f4 >= 2 or (f4 == 1 and nbn >= 4096)):
        return f"wattn_fwd4_kernel<{m}>"
    return f"wattn_fwd_kernel<{'__bf16' if dtype == torch.bfloat16 else 'float'}, {m}>"


def wattn_bwd_kernel_name(dtype, masked, inverse_map=True):
    """Mirrors wattn_bwd_launch (attention_win.hip): bf16 with the inverse pos map takes the four-wave kernel."""
    m = "true" if masked else "false"
    if dtype == torch.bfloat16 and inverse_map and options.get("WATTN_BWD4"):
        return f"wattn_bwd4_kernel<{m}>"
    return f"wattn_bwd_kernel<{'__bf16' if dtype == torch.bfloat16 else 'float'}, {m}>"


def skinny_ok(N, K, M, vec=False):
    """Mirrors gemm_skinny_ok (gemm_skinny.hip): the weight-resident streaming kernel (contiguous operands assumed);
    ``vec``: the epilogue reads a residual or z (those launches stay on the tiled kernels unless GEMM_SKINNY = 2)."""
    opt = options.get("GEMM_SKINNY")
    fits = any(N % (32 * nc) == 0 and (N // nc) * (K + 8) * 2 + (N // nc) * 4 <= 150 * 1024 for nc in (1, 2, 4))
    return bool(opt) and (not vec or opt == 2) and K in (64, 96, 128) and N >= 32 and M >= 32768 and fits


def glds_ok(N, K):
    """Shapes the LDS-DMA GEMM takes (mirrors gemm_glds_ok in gemm_glds.hip)."""
    return K % 64 == 0 or (K % 32 == 0 and N % 128 == 0)


def cu_count():
    """Compute units of the current device as the library's dispatch heuristics see them (vtx_cu_count)."""
    return _lib.load().vtx_cu_count()



def astat_ok(N, K, M, bias=True, plain=True):
    """Mirrors gemm_astat_ok (gemm_astat.hip) for contiguous bf16 operands: the A-stationary persistent kernel takes C[M, N] over
    192 <= K <= 384 once a launch has two 128 x 128 tiles per CU (M: the rows a mapped launch computes); N % 128 == 64 (a ragged last
    column tile) only for ``plain`` launches: no residual / activation / saved z / row map."""
    mode = options.get("GEMM_ASTAT")
    if not mode or K % 64 or K < 192 or K > 384 or N % 64 or N < 256 or (N > 1536 and bias) or M <= 0 or (N % 128 and not plain):
        return False
    return mode == 2 or 4 * ((M + 127) // 128) * ((N + 127) // 128) >= (5 if mode == 3 else 8) * cu_count()


def pp_nf(N):
    """16-column accumulator tiles per wave of the two-group GEMM = tile width / 64 (mirrors pp_nf, csrc/gemm_pp.hip); 0: not its shape."""
    return 3 if N % 192 == 0 else (4 if N % 256 == 0 else (2 if N % 128 == 0 else 0))


def pp_wmf(M, N):
    """Tile height (in 32-row units) the two-group GEMM picks (mirrors pp_pick_wmf, csrc/gemm_pp.hip)."""
    nf = pp_nf(N)
    cus, ntn = cu_count(), N // (64 * nf)
    wmax = 5 if nf == 4 else 7
    best, cost = wmax, None
    for w in range(wmax, 3, -1):
        tiles = (M + 32 * w - 1) // (32 * w) * ntn
        c = ((tiles + cus - 1) // cus) * (w + 2)
        if cost is None or c < cost:
            best, cost = w, c
    return best


def pp_ok(N, K, M):
    """Mirrors gemm_pp_ok (csrc/gemm_pp.hip) for contiguous bf16 operands: the two-group kernel takes the long contractions
    (K >= 1152, or K >= 768 with N <= 384, or K >= 384 with N = 192) of N % 192 == 0 layers once a launch nearly fills a round (M: the rows it computes)."""
    mode = options.get("GEMM_PP")
    nf = pp_nf(N)
    if not mode or not nf or K % 64 or M <= 0:
        return False
    if mode >= 2:
        return True
    if nf != 3:                      # 128-column tiles from K = 1024 up (PVT-Small stage 2); 256-column tiles only under GEMM_PP = 2
        return nf == 2 and K >= 1024 and 4 * ((M + 127) // 128) * (N // 128) >= 3 * cu_count()
    return (K >= 1152 or (K >= 768 and N <= 384) or (K >= 384 and N == 192)) and 4 * ((M + 127) // 128) * (N // 192) >= 3 * cu_count()


def gemm_kernel_name(dtype, N, mode, out_f32=False, K=0, M=0, mapped=False, vec=False, bias=True):
    """Name of the kernel instantiation vtx_gemm / vtx_wgrad picks (mirrors gemm.hip / gemm_glds.hip); ``mapped``: the
    row-mapped variant of a compacted branch (M = the rows it computes)."""
    t = "__bf16" if dtype == torch.bfloat16 else "float"
    to = "float" if out_f32 else t
    bn = 128 if N % 128 == 0 else (96 if N % 96 == 0 else (64 if N <= 64 else 128))
    if dtype == torch.bfloat16 and mode == 0 and skinny_ok(N, K, M, vec) and not mapped:
        return f"gemm_skinny_kernel<{K // 32}>"
    if dtype == torch.bfloat16 and mode == 0 and K > 0 and glds_ok(N, K) and pp_ok(N, K, M):
        mode_ = options.get("GEMM_PP")
        nf = pp_nf(N)
        w = min(mode_ % 10, 5 if nf == 4 else 7) if 100 <= mode_ < 1000 else pp_wmf(M, N)
        if nf != 3:
            return f"gemm_ppn_kernel<{w}, {'true' if mapped else 'false'}, {nf}>"
        return f"gemm_pp_kernel<{w}, {'true' if mapped else 'false'}, 0>"
    if dtype == torch.bfloat16 and mode == 0 and astat_ok(N, K, M, bias, plain=not vec and not mapped):
        return f"gemm_astat_kernel<{K // 64}, {'true' if mapped else 'false'}, ...>"
    if dtype == torch.bfloat16 and mode == 0 and K > 0 and glds_ok(N, K):
        force = options.get("GLDS_BM")                                     # mirrors glds_pick_bm in gemm_glds.hip
        if force in (64, 128):
            bm = force
        elif bn != 128 or K % 64 != 0:
            bm = 64
        else:
            t128 = ((N + 127) // 128) * ((M + 127) // 128)
            bm = 128 if (t128 >= 800 or 384 < t128 <= 512) else 64
        if K % 64 != 0:
            return f"gemm_glds_kernel<{bm}, {bn}, 32, 3, 2>"
        if bn == 128 and options.get("GLDS_EPI") == 1:                     # mirrors glds_launch_t: wave-private epilogue
            return f"gemm_glds_pv_kernel<{bm}, {2 if bm == 128 else 4}, {'true' if mapped else 'false'}>"
        nwn = 4 if (bn == 128 and options.get("GLDS_WAVES") != 4) else 2
        return f"gemm_glds_kernel<{bm}, {bn}, 64, 2, {nwn}>"
    ta, tb = {0: ("false", "false"), 1: ("false", "true"), 2: ("true", "true")}[mode]
    return f"gemm_kernel<{t}, {to}, 128, {bn}, {ta}, {tb}>"


# ------------------------------------------------------------------------------- GEMM family
ACT_NONE, ACT_SILU, ACT_DSILU, ACT_GELU, ACT_DGELU = 0, 1, 2, 3, 4


def gemm(a, w, mode=0, bias=None, resid=None, rowscale=None, rows_per_scale=1, act=ACT_NONE, aux_in=None,
         want_aux=False, out=None):
    """mode 0: a[M,K] @ w[N,K]^T ; mode 1: a[M,K] @ w[K,N].  Fused epilogue per vtx.h; returns (C, z) if want_aux."""
    _dev(a, w, bias, resid, rowscale, aux_in, out)
    _f32(bias, "bias"); _f32(rowscale, "rowscale")
    if a.dtype != w.dtype:
        raise VtxError(f"vtx: gemm operand dtypes differ ({a.dtype} vs {w.dtype})")
    lib = _lib.load()
    K = a.shape[-1]
    M = a.numel() // K
    if mode == 0:
        N, kw = w.shape
    else:
        kw, N = w.shape
    if kw != K:
        raise VtxError(f"vtx: gemm contraction mismatch ({K} vs {kw})")
    c = out if out is not None else torch.empty(a.shape[:-1] + (N,), dtype=a.dtype, device=a.device)
    aux = torch.empty_like(c) if want_aux else None
    ev = None
    if _timer is not None:
        es = a.element_size()
        nb = es * (M * K + N * K + M * N * (1 + (resid is not None) + bool(want_aux) + (aux_in is not None)))
        ev = _timer.bracket(gemm_kernel_name(a.dtype, N, mode, K=K, M=M, vec=resid is not None or act in (ACT_DSILU, ACT_DGELU),
                                             bias=bias is not None),
                            2.0 * M * N * K, float(nb))
    if ev:
        ev[0].record()
    check(lib.vtx_gemm(mode, _dt(a), _p(a), _p(w), _p(c), M, N, K, K, w.shape[1], N, _p(bias), _p(resid),
                       _p(rowscale), int(rows_per_scale), _p(aux), _p(aux_in), act, _stream()), "vtx_gemm")
    if ev:
        ev[1].record()
    return (c, aux) if want_aux else c


def wgrad_wide_tiles(pairs, want_j=False):
    """128 x 64 J tiles (J = 6 .. 3, the widest whose width divides every Kin) of a grouped weight gradient [(N, Kin), ...], or 0 when
    the group stays on 128 x 128 tiles (mirrors wgrad_wide_tiles, csrc/gemm_wgrad_glds.hip).  ``want_j``: (tiles, J)."""
    on = options.get("WGRAD_WIDE")
    r4 = bool(on & 4)                  # the round-4 rule: whole 128 x 384 tiles only
    cus = cu_count()
    fill = ((on >> 4) & 255) or 85
    all384 = not r4 and all(k % 384 == 0 for _, k in pairs)      # 256-column tiles as the fallback of under-filled 384-column groups (C = 768)
    for J in ((6,) if r4 else (6, 5, 4, 3)) if on else ():
        if J == 4 and not (on & 8) and not all384:
            continue
        need = 75 if (J == 4 and not (on & 8) and fill > 75) else fill
        kw = 64 * J
        if any(k % kw or n % 8 or n < 64 or (r4 and n % 128) for n, k in pairs):
            continue
        tiles = sum(((n + 127) // 128) * (k // kw) for n, k in pairs)
        if 1 <= tiles <= cus and 100 * ((cus // tiles) * tiles) >= need * cus:
            return (tiles, J) if want_j else tiles
    return (0, 0) if want_j else 0


def wgrad_group_kernel_name(pairs, mapped="false"):
    if wgrad_wide_tiles(pairs):
        return f"wgrad_wide_kernel<{mapped}>"
    return f"wgrad_glds_kernel<64, 2, {4 if options.get('WG_WAVES') == 4 else 8}, {mapped}>"


def wgrad_kernel_name(dtype, N, Kin, glds):
    if glds:
        return f"wgrad_glds_kernel<64, 2, {4 if options.get('WG_WAVES') == 4 else 8}, false>"
    t = "__bf16" if dtype == torch.bfloat16 else "float"
    bn = 128 if Kin % 128 == 0 else (96 if Kin % 96 == 0 else (64 if Kin <= 64 else 128))
    return f"gemm_kernel<{t}, float, 128, {bn}, true, true>"


def _wgrad_glds_shape(dtype, N, Kin, rowscale, scale_const):
    return (dtype == torch.bfloat16 and options.get("WGRAD_GLDS") and N % 8 == 0 and Kin % 8 == 0 and N >= 64 and
            Kin >= 64 and (rowscale is None or scale_const > 0))


def wgrad(dy, x, want_bias=True, rowscale=None, rows_per_scale=1, scale_const=0.0, out=None):
    """dW[N,Kin] (fp32), dbias[N] (fp32 or None) from dy[M,N], x[M,Kin].  scale_const > 0: every rowscale value
    is 0 or scale_const (DropPath), see vtx.h.  ``out``: fp32 [N, Kin] destination (e.g. a gradient-bucket view)."""
    _dev(dy, x, rowscale)
    lib = _lib.load()
    N, Kin = dy.shape[-1], x.shape[-1]
    M = dy.numel() // N
    if x.numel() // Kin != M:
        raise VtxError("vtx: wgrad token-count mismatch")
    dW = out if out is not None else torch.empty((N, Kin), dtype=torch.float32, device=x.device)
    db = torch.empty(N, dtype=torch.float32, device=x.device) if want_bias else None
    wsb = lib.vtx_wgrad_workspace(M, N, Kin)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    glds = _wgrad_glds_shape(x.dtype, N, Kin, rowscale, scale_const)
    with _timed(wgrad_kernel_name(x.dtype, N, Kin, glds) + " (+split-K reduce)", 2.0 * M * N * Kin,
                x.element_size() * M * (N + Kin) + 4.0 * N * Kin):
        check(lib.vtx_wgrad(_dt(x), _p(dy), _p(x), _p(dW), _p(db), M, N, Kin, N, Kin, _p(rowscale),
                            int(rows_per_scale), float(scale_const), _p(ws), wsb, _stream()),
              "vtx_wgrad")
    return dW, db


def wgrad_group_ok(jobs, rows_per_scale=1, scale_const=0.0):
    """True when ``jobs`` = [(dy, x, want_bias, rowscale), ...] can run as ONE grouped LDS-DMA weight-gradient launch
    (bf16, same token count, every shape eligible; mirrors vtx_wgrad_group_ok)."""
    lib = _lib.load()
    n = len(jobs)
    if n < 1 or n > lib.vtx_wgrad_group_max():
        return False
    dy0, x0 = jobs[0][0], jobs[0][1]
    M = dy0.numel() // dy0.shape[-1]
    for dy, x, _, _ in jobs:
        if dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or dy.numel() // dy.shape[-1] != M or \
                x.numel() // x.shape[-1] != M:
            return False
    Ns = (ctypes.c_int * n)(*[j[0].shape[-1] for j in jobs])
    Ks = (ctypes.c_int * n)(*[j[1].shape[-1] for j in jobs])
    has_rs = int(any(j[3] is not None for j in jobs))
    return bool(lib.vtx_wgrad_group_ok(BF16, n, Ns, Ks, M, has_rs, int(rows_per_scale), float(scale_const)))


def wgrad_group_slices(jobs):
    """Split-K slices the grouped launch of ``jobs`` runs with (>= 2: its outputs come from the reduce launch, which can
    then accumulate onto existing gradients -- ``wgrad_group(accumulate=...)``)."""
    n = len(jobs)
    M = jobs[0][0].numel() // jobs[0][0].shape[-1]
    Ns = (ctypes.c_int * n)(*[j[0].shape[-1] for j in jobs])
    Ks = (ctypes.c_int * n)(*[j[1].shape[-1] for j in jobs])
    return int(_lib.load().vtx_wgrad_group_slices(n, Ns, Ks, M))


class RawPtr:
    """A device address standing in for a tensor somebody else keeps alive (wgrad_group(accumulate=...))."""
    __slots__ = ("ptr",)

    def __init__(self, ptr):
        self.ptr = int(ptr)

    def data_ptr(self):
        return self.ptr


def wgrad_group(jobs, rows_per_scale=1, scale_const=0.0, outs=None, colparts=None, accumulate=None):
    """The weight gradients of several linears over the SAME tokens in one launch (csrc/gemm_wgrad_glds.hip):
    jobs = [(dy [M, N_i], x [M, Kin_i], want_bias, rowscale or None), ...] -> [(dW_i fp32 [N_i, Kin_i], db_i or None)].
    Split-K partials are summed by one following reduce launch -- deterministic, fixed slice order.
    ``colparts``: up to 4 deferred column reductions (Partials) that ride in that reduce launch; the result is then
    (gradients, [(out0, out1 or None)]) with the bits of colreduce_multi.
    ``accumulate`` = (dW addresses [n], dbias addresses [n], [(out0, out1) addresses per colpart]): nothing is allocated,
    every result is ADDED onto the fp32 gradient at that address (needs wgrad_group_slices(jobs) >= 2); returns None."""
    lib = _lib.load()
    n = len(jobs)
    for dy, x, _, rs in jobs:
        _dev(dy, x, rs)
    dev = jobs[0][1].device
    M = jobs[0][0].numel() // jobs[0][0].shape[-1]
    Nl = [j[0].shape[-1] for j in jobs]
    Kl = [j[1].shape[-1] for j in jobs]
    if accumulate is not None:
        dWs = [RawPtr(a) for a in accumulate[0]]
        dbs = [RawPtr(a) if jobs[i][2] else None for i, a in enumerate(accumulate[1])]
    else:
        dWs = [outs[i] if outs is not None and outs[i] is not None else
               torch.empty((Nl[i], Kl[i]), dtype=torch.float32, device=dev) for i in range(n)]
        dbs = [torch.empty(Nl[i], dtype=torch.float32, device=dev) if jobs[i][2] else None for i in range(n)]
    Ns, Ks = (ctypes.c_int * n)(*Nl), (ctypes.c_int * n)(*Kl)
    lds = (ctypes.c_int64 * n)(*Nl)
    ldx = (ctypes.c_int64 * n)(*Kl)
    vp = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
    wsb = lib.vtx_wgrad_group_workspace(n, Ns, Ks, M)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    flops = sum(2.0 * M * a * b for a, b in zip(Nl, Kl))
    nbytes = sum(2.0 * M * (a + b) + 4.0 * a * b for a, b in zip(Nl, Kl))
    nc = len(colparts) if colparts else 0
    if accumulate is not None:
        couts = [(RawPtr(a), RawPtr(b) if p.two else None) for p, (a, b) in zip(colparts or [], accumulate[2])]
    else:
        couts = [(torch.empty(p.C, dtype=torch.float32, device=dev),
                  torch.empty(p.C, dtype=torch.float32, device=dev) if p.two else None) for p in (colparts or [])]
    cvp = lambda ts: (ctypes.c_void_p * max(nc, 1))(*[None if t is None else t.data_ptr() for t in ts]) if nc else None
    cia = lambda xs: (ctypes.c_int * max(nc, 1))(*xs) if nc else None
    cp = colparts or []
    with _timed(wgrad_group_kernel_name(list(zip(Nl, Kl))) + (" (+split-K and column reduce)" if nc else " (+split-K reduce)"),
                flops, nbytes + sum(4.0 * p.nb * p.ld for p in cp)):
        check(lib.vtx_wgrad_group(BF16, n, vp([j[0] for j in jobs]), vp([j[1] for j in jobs]), vp(dWs), vp(dbs), Ns, Ks,
                                  lds, ldx, vp([j[3] for j in jobs]), int(rows_per_scale), float(scale_const), M, _p(ws),
                                  wsb, nc, cvp([p.ws for p in cp]), cvp([o[0] for o in couts]), cvp([o[1] for o in couts]),
                                  cia([p.nb for p in cp]), cia([p.C for p in cp]), cia([p.ld for p in cp]),
                                  int(accumulate is not None), _stream()),
              "vtx_wgrad_group")
    if accumulate is not None:
        return None
    res = list(zip(dWs, dbs))
    return (res, couts) if colparts is not None else res


# ------------------------------------------------------------------------------- PVT / Twins: spatial-reduction attention
def _sr_head_dim(q, kv, B, Lq, Lk, n_head):
    """Head dim of the operands (64: PVT, 32: Twins-SVT); the kernels exist for these two."""
    if q.dtype != kv.dtype:
        raise VtxError("vtx: srattn operand dtypes differ")
    hd = q.numel() // max(B * Lq, 1)
    D = hd // max(n_head, 1)
    if D not in (32, 64) or D * n_head != hd or q.numel() != B * Lq * hd or kv.numel() != B * Lk * 2 * hd:
        raise VtxError(f"vtx: srattn shape mismatch (q {tuple(q.shape)}, kv {tuple(kv.shape)}, {n_head} heads: head dim must be 32 or 64)")
    return D, hd


def _drop_args(drop, dev, cells=None):
    """(p, seed, keep) of an attention-dropout call -> (float p, int seed, keep pointer); keep: uint8 [problems][Lq][Lk] or None.
    ``cells`` = problems * Lq * Lk of THIS call: a keep mask of any other size (a transposed or per-image layout) would be read out of
    bounds by the kernels' drop_factor -- refused here, forward and backward alike."""
    p, seed, keep = drop
    if not 0.0 < float(p) < 1.0:
        raise VtxError(f"vtx: attention dropout probability {p} outside (0, 1)")
    if keep is not None:
        _dev(keep)
        if keep.dtype != torch.uint8 or not keep.is_contiguous():
            raise VtxError("vtx: attention keep mask must be a contiguous uint8 tensor")
        if cells is not None and keep.numel() != cells:
            raise VtxError(f"vtx: attention keep mask has {keep.numel()} cells, this call needs [problems, Lq, Lk] = {cells}")
    return float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, keep


def attn_keep_mask(nprob, Lq, Lk, p, seed, device):
    """uint8 [nprob, Lq, Lk]: the keep decisions the dropout kernels regenerate from (p, seed) -- tests / external checkers."""
    out = torch.empty((nprob, Lq, Lk), dtype=torch.uint8, device=device)
    check(_lib.load().vtx_attn_keep_mask(_p(out), nprob, Lq, Lk, float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream()),
          "vtx_attn_keep_mask")
    return out


def srattn_fwd(q, kv, B, Lq, Lk, n_head, drop=None):
    """o [B*Lq, h*D], lse from q [B*Lq, h*D] and kv [B*Lk, 2*h*D] (k | v) -- reference models/pvt.py:38-66, twins.py:56-93.
    drop = (p, seed, keep): dropout of the attention probabilities (pvt.py:60)."""
    _dev(q, kv)
    D, hd = _sr_head_dim(q, kv, B, Lq, Lk, n_head)
    o = torch.empty_like(q)
    lse = torch.empty(B * n_head * Lq, dtype=torch.float32, device=q.device)
    tn = "__bf16" if q.dtype == torch.bfloat16 else "float"
    ev = _attn_bracket(f"srattn_fwd_kernel<{tn}, {D}>", B * n_head, Lq, D, B * Lq, hd, q.element_size(), False)
    if ev is not None:                              # Lq x Lk products, not Lq x Lq
        _timer.records[-1] = (_timer.records[-1][0], 4.0 * B * n_head * Lq * Lk * D,
                              q.element_size() * (2.0 * B * Lq * hd + 2.0 * B * Lk * hd)) + _timer.records[-1][3:]
    if drop is not None:
        dp, seed, keep = _drop_args(drop, q.device, B * n_head * Lq * Lk)
        check(_lib.load().vtx_srattn_fwd_drop(_p(q), _p(kv), _p(o), _p(lse), B, Lq, Lk, n_head, D, _dt(q), dp, seed, _p(keep),
                                              _stream()), "vtx_srattn_fwd_drop")
    else:
        check(_lib.load().vtx_srattn_fwd(_p(q), _p(kv), _p(o), _p(lse), B, Lq, Lk, n_head, D, _dt(q), _stream()),
              "vtx_srattn_fwd")
    if ev:
        ev[1].record()
    return o, lse


def srattn_scores(q, kv, B, Lq, Lk, n_head):
    """Pre-softmax scores q k^T / sqrt(D) as (B, n_head, Lq, Lk) -- what pvt.MultiHeadedAttention.forward returns second."""
    _dev(q, kv)
    D, _ = _sr_head_dim(q, kv, B, Lq, Lk, n_head)
    score = torch.empty((B, n_head, Lq, Lk), dtype=q.dtype, device=q.device)
    check(_lib.load().vtx_srattn_scores(_p(q), _p(kv), _p(score), B, Lq, Lk, n_head, D, _dt(q), _stream()), "vtx_srattn_scores")
    return score


def srattn_bwd(q, kv, o, dout, lse, B, Lq, Lk, n_head, drop=None):
    """dq, dkv (deterministic); drop: the forward's (p, seed, keep)."""
    _dev(q, kv, o, dout, lse)
    lib = _lib.load()
    D, hd = _sr_head_dim(q, kv, B, Lq, Lk, n_head)
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    wsb = lib.vtx_srattn_bwd_workspace(B, Lq, Lk, n_head, D)
    ws = torch.empty(wsb, dtype=torch.uint8, device=q.device)
    tn = "__bf16" if q.dtype == torch.bfloat16 else "float"
    ev = _attn_bracket(f"srattn_bwd_kernel<{tn}, {D}>", B * n_head, Lq, D, B * Lq, hd, q.element_size(), True)
    if ev is not None:
        _timer.records[-1] = (_timer.records[-1][0], 10.0 * B * n_head * Lq * Lk * D,
                              q.element_size() * (4.0 * B * Lq * hd + 4.0 * B * Lk * hd)) + _timer.records[-1][3:]
    if drop is not None:
        dp, seed, keep = _drop_args(drop, q.device, B * n_head * Lq * Lk)
        check(lib.vtx_srattn_bwd_drop(_p(q), _p(kv), _p(o), _p(dout), _p(lse), _p(dq), _p(dkv), _p(ws), wsb, B, Lq, Lk, n_head, D,
                                      _dt(q), dp, seed, _p(keep), _stream()), "vtx_srattn_bwd_drop")
    else:
        check(lib.vtx_srattn_bwd(_p(q), _p(kv), _p(o), _p(dout), _p(lse), _p(dq), _p(dkv), _p(ws), wsb, B, Lq, Lk, n_head, D,
                                 _dt(q), _stream()), "vtx_srattn_bwd")
    if ev:
        ev[1].record()
    return dq, dkv


# ------------------------------------------------------------------------------- halo attention (models/halo_transformer.py:22-115)
def window_gather(x, B, H, W, c0, nc, win, halo):
    """[B * nW, (win + 2 halo)^2, nc] neighbourhoods (zero rows outside the map) of channels [c0, c0 + nc) of the map x (B, H, W, ld)."""
    _dev(x)
    ld = x.shape[-1]
    if x.numel() != B * H * W * ld or not x.is_contiguous():
        raise VtxError("vtx: window_gather expects a contiguous (B, H, W, C) map")
    nW, side = (H // win) * (W // win), win + 2 * halo
    out = torch.empty((B * nW, side * side, nc), dtype=x.dtype, device=x.device)
    check(_lib.load().vtx_window_gather(_p(x), _p(out), B, H, W, ld, c0, nc, win, halo, _dt(x), _stream()), "vtx_window_gather")
    return out


def window_scatter(src, out, B, H, W, c0, nc, win, halo):
    """The adjoint of window_gather into channels [c0, c0 + nc) of the map ``out`` (B, H, W, ld): sums over the neighbourhoods."""
    _dev(src, out)
    ld = out.shape[-1]
    nW, side = (H // win) * (W // win), win + 2 * halo
    if out.numel() != B * H * W * ld or not out.is_contiguous():
        raise VtxError("vtx: window_scatter expects a contiguous (B, H, W, C) destination map")
    if src.dtype != out.dtype or not src.is_contiguous() or src.numel() != B * nW * side * side * nc:
        raise VtxError(f"vtx: window_scatter source must be a contiguous [{B * nW}, {side * side}, {nc}] tensor of the map's dtype")
    if c0 < 0 or nc <= 0 or c0 + nc > ld:
        raise VtxError(f"vtx: window_scatter channels [{c0}, {c0 + nc}) outside the map's {ld}")
    check(_lib.load().vtx_window_scatter(_p(src), _p(out), B, H, W, ld, c0, nc, win, halo, _dt(out), _stream()), "vtx_window_scatter")
    return out


def table_bias(table, pos, n_head):
    """bias [n_head, *pos.shape] = table[pos][..., h] for an int64 index tensor of any shape (halo_transformer.py:95-98)."""
    _dev(table, pos)
    _f32(table, "rel_pos")
    if pos.dtype != torch.int64:
        raise VtxError("vtx: pos must be int64 (the reference's buffer dtype)")
    bias = torch.empty((n_head,) + tuple(pos.shape), dtype=torch.float32, device=table.device)
    check(_lib.load().vtx_table_bias(_p(table), _p(pos), _p(bias), pos.numel(), n_head, _stream()), "vtx_table_bias")
    return bias


def table_bias_bwd(full, csr, ntab, n_head):
    """dtable [ntab, n_head] from the full gradient [n_head, cells] through the CSR (order, offsets) of pos."""
    order, offsets = csr
    _dev(full, order, offsets)
    out = torch.empty((ntab, n_head), dtype=torch.float32, device=full.device)
    check(_lib.load().vtx_table_bias_bwd(_p(full), _p(order), _p(offsets), _p(out), full.numel() // n_head, n_head, ntab, _stream()),
          "vtx_table_bias_bwd")
    return out


def xattn_fwd(q, kv, B, Lq, Lk, n_head, bias=None, drop=None):
    """o, lse = softmax(q k^T / sqrt(D) + bias) v; q [B * Lq, h D], kv [B * Lk, 2 h D], bias [h, Lq, Lk] fp32 or None; drop = (p, seed, keep)."""
    _dev(q, kv, bias)
    D, hd = _sr_head_dim(q, kv, B, Lq, Lk, n_head)
    o = torch.empty_like(q)
    lse = torch.empty(B * n_head * Lq, dtype=torch.float32, device=q.device)
    ev = _attn_bracket("lattn_fwd_kernel (cross)", B * n_head, Lq, D, B * Lq, hd, q.element_size(), False)
    if drop is not None:
        dp, seed, keep = _drop_args(drop, q.device, B * n_head * Lq * Lk)
        check(_lib.load().vtx_xattn_fwd_drop(_p(q), _p(kv), _p(o), _p(lse), _p(bias), B, Lq, Lk, n_head, D, _dt(q), dp, seed, _p(keep),
                                             _stream()), "vtx_xattn_fwd_drop")
    else:
        check(_lib.load().vtx_xattn_fwd(_p(q), _p(kv), _p(o), _p(lse), _p(bias), B, Lq, Lk, n_head, D, _dt(q), _stream()), "vtx_xattn_fwd")
    if ev:
        ev[1].record()
    return o, lse


def xattn_bwd(q, kv, o, dout, lse, B, Lq, Lk, n_head, bias=None, drop=None):
    """dq, dkv, dbias (None without a bias): deterministic; drop: the forward's (p, seed, keep)."""
    _dev(q, kv, o, dout, lse, bias)
    lib = _lib.load()
    D, hd = _sr_head_dim(q, kv, B, Lq, Lk, n_head)
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    dbias = torch.empty_like(bias) if bias is not None else None
    wsb = lib.vtx_xattn_bwd_workspace(B, Lq, n_head)
    ws = torch.empty(wsb, dtype=torch.uint8, device=q.device)
    ev = _attn_bracket("lattn_bwd_*_kernel (cross)", B * n_head, Lq, D, B * Lq, hd, q.element_size(), True)
    if drop is not None:
        dp, seed, keep = _drop_args(drop, q.device, B * n_head * Lq * Lk)
        check(lib.vtx_xattn_bwd_drop(_p(q), _p(kv), _p(o), _p(dout), _p(lse), _p(bias), _p(dq), _p(dkv), _p(dbias), _p(ws), wsb, B, Lq, Lk,
                                     n_head, D, _dt(q), dp, seed, _p(keep), _stream()), "vtx_xattn_bwd_drop")
    else:
        check(lib.vtx_xattn_bwd(_p(q), _p(kv), _p(o), _p(dout), _p(lse), _p(bias), _p(dq), _p(dkv), _p(dbias), _p(ws), wsb, B, Lq, Lk,
                                n_head, D, _dt(q), _stream()), "vtx_xattn_bwd")
    if ev:
        ev[1].record()
    return dq, dkv, dbias


def dwconv3_fwd(x, w, adjoint=False):
    """y = x + DepthwiseConv3x3(x) on channels-last x (B, H, W, C), w (C, 1, 3, 3) fp32 -- twins.py:25-37; adjoint: the input
    gradient of the same map (x := dy)."""
    _dev(x, w)
    _f32(w, "peg weight")
    B, H, W, C = x.shape
    if w.numel() != 9 * C:
        raise VtxError(f"vtx: dwconv3 weight has {w.numel()} elements for {C} channels")
    y = torch.empty_like(x)
    check(_lib.load().vtx_dwconv3_fwd(_p(x), _p(w), _p(y), B, H, W, C, int(adjoint), _dt(x), _stream()), "vtx_dwconv3_fwd")
    return y


def dwconv3_wgrad(x, dy):
    """dw (C, 1, 3, 3) fp32 of y = x + DepthwiseConv3x3(x) (deterministic)."""
    _dev(x, dy)
    if x.dtype != dy.dtype or x.shape != dy.shape:
        raise VtxError("vtx: dwconv3_wgrad operands differ in dtype / shape")
    lib = _lib.load()
    B, H, W, C = x.shape
    wsb = lib.vtx_dwconv3_wgrad_workspace(B, H, W, C)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=x.device)
    dw = torch.empty((C, 1, 3, 3), dtype=torch.float32, device=x.device)
    check(lib.vtx_dwconv3_wgrad(_p(x), _p(dy), _p(dw), _p(ws), wsb, B, H, W, C, _dt(x), _stream()), "vtx_dwconv3_wgrad")
    return dw


def twins_subsample_fwd(x, B, H, W, C, r, transposed=False):
    """Patch matrix [B*(H/r)*(W/r), C*r*r] (columns (c', py, px): the Conv2d weight's own layout) of
    twins.MultiHeadedAttention's reduction conv on x [B, H, W, C] (any view of B*H*W*C contiguous elements), with the
    reference's reshape kept as written (twins.py:69-70; include/vtx.h).  ``transposed``: also the [C*r*r, rows] copy."""
    _dev(x)
    out = torch.empty((B * (H // r) * (W // r), C * r * r), dtype=x.dtype, device=x.device)
    out_t = torch.empty((C * r * r, B * (H // r) * (W // r)), dtype=x.dtype, device=x.device) if transposed else None
    check(_lib.load().vtx_twins_subsample_fwd(_p(x), _p(out), _p(out_t), B, H, W, C, r, _dt(x), _stream()),
          "vtx_twins_subsample_fwd")
    return (out, out_t) if transposed else out


def bias_cast(x, bias, dtype):
    """(x [rows, C] fp32 + bias [C] fp32) as ``dtype`` -- the epilogue behind a split-K launch used as a forward GEMM."""
    _dev(x, bias)
    _f32(x, "bias_cast input"); _f32(bias, "bias")
    C = x.shape[-1]
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    code = BF16 if dtype == torch.bfloat16 else F32
    check(_lib.load().vtx_bias_cast(_p(x), _p(bias), _p(out), x.numel() // C, C, code, _stream()), "vtx_bias_cast")
    return out


def twins_subsample_bwd(dout, dx, B, H, W, C, r, accumulate=False):
    """Inverse scatter of twins_subsample_fwd into dx (B*H*W*C elements; optionally accumulating)."""
    _dev(dout, dx)
    check(_lib.load().vtx_twins_subsample_bwd(_p(dout), _p(dx), B, H, W, C, r, int(accumulate), _dt(dx), _stream()),
          "vtx_twins_subsample_bwd")
    return dx


def patchify_fwd(x, B, H, W, C, p, skip=0):
    """Token-major features [B, skip + H*W, C] -> patch matrix [B*(H/p)*(W/p), p*p*C], columns (py, px, c)."""
    _dev(x)
    out = torch.empty((B * (H // p) * (W // p), p * p * C), dtype=x.dtype, device=x.device)
    check(_lib.load().vtx_patchify_fwd(_p(x), _p(out), B, H, W, C, p, skip, _dt(x), _stream()), "vtx_patchify_fwd")
    return out


def patchify_bwd(dout, dx, B, H, W, C, p, skip=0, accumulate=False):
    """Inverse scatter of patchify_fwd into dx [B, skip + H*W, C] (optionally accumulating)."""
    _dev(dout, dx)
    check(_lib.load().vtx_patchify_bwd(_p(dout), _p(dx), B, H, W, C, p, skip, int(accumulate), _dt(dx), _stream()),
          "vtx_patchify_bwd")
    return dx


def add_pos_fwd(x, cls, pos):
    """x [B, T, C] + pos [s + T, C] with an optional cls token row (s = 1) in front (pvt.py:133-137)."""
    _dev(x, cls, pos)
    _f32(cls, "cls_token"); _f32(pos, "pos")
    B, T, C = x.shape
    s = 0 if cls is None else 1
    out = torch.empty((B, T + s, C), dtype=x.dtype, device=x.device)
    check(_lib.load().vtx_add_pos_fwd(_p(x), _p(cls), _p(pos), _p(out), B, T, C, _dt(x), _stream()), "vtx_add_pos_fwd")
    return out


def add_pos_bwd(dout, has_cls):
    _dev(dout)
    B, L, C = dout.shape
    s = 1 if has_cls else 0
    dx = torch.empty((B, L - s, C), dtype=dout.dtype, device=dout.device)
    dpos = torch.empty((L, C), dtype=torch.float32, device=dout.device)
    dcls = torch.empty(C, dtype=torch.float32, device=dout.device) if has_cls else None
    check(_lib.load().vtx_add_pos_bwd(_p(dout), _p(dx), _p(dcls), _p(dpos), B, L - s, C, _dt(dout), _stream()),
          "vtx_add_pos_bwd")
    return dx, dcls, dpos


# ------------------------------------------------------------------------------- optimizer tail
def _ptr_array(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def grad_sqnorm(grads, static=None):
    """[sum g^2, sqrt(sum g^2)] over a list of fp32 gradient tensors (deterministic two-level reduction).
    ``static`` = (numel array, chunk count) of a caller that runs the same tensor list every step (FusedAdamW)."""
    lib = _lib.load()
    if static is None:
        _dev(*grads)
        chunk = lib.vtx_opt_chunk()
        numel = (ctypes.c_int64 * len(grads))(*[g.numel() for g in grads])
        nchunks = sum((g.numel() + chunk - 1) // chunk for g in grads)
    else:
        numel, nchunks = static
    dev = grads[0].device
    partial = torch.empty(max(nchunks, 1), dtype=torch.float32, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    with _timed("grad_sqnorm_*_kernel", 0.0, 4.0 * sum(g.numel() for g in grads)):
        check(lib.vtx_grad_sqnorm(len(grads), _ptr_array(grads), numel, _p(partial), _p(out), _stream()), "vtx_grad_sqnorm")
    return out


def adamw_step(params, grads, exp_avg, exp_avg_sq, lrs, wds, norm, max_norm, beta1, beta2, eps, t, static=None):
    """torch.optim.AdamW step t of the listed tensors in one multi-tensor pass (csrc/optim.hip).
    ``static`` = (param / exp_avg / exp_avg_sq address arrays, numel array, total elements) of a caller that steps the same,
    already validated, tensors every time (FusedAdamW): only the gradients are looked at per step."""
    n = len(grads)
    if static is None:
        _dev(*params, *grads, *exp_avg, *exp_avg_sq, norm)
        pa, ma, va = _ptr_array(params), _ptr_array(exp_avg), _ptr_array(exp_avg_sq)
        numel = (ctypes.c_int64 * n)(*[p.numel() for p in params])
        total = sum(p.numel() for p in params)
    else:
        pa, ma, va, numel, total = static
    with _timed("adamw_step_kernel", 0.0, 28.0 * total):     # p, g, m, v read; p, m, v written
        check(_lib.load().vtx_adamw_step(n, pa, _ptr_array(grads), ma,
                                         va, numel, (ctypes.c_float * n)(*lrs),
                                         (ctypes.c_float * n)(*wds), _p(norm), float(max_norm), float(beta1), float(beta2),
                                         float(eps), int(t), _stream()), "vtx_adamw_step")


def ema_update(targets, sources, momentum):
    """targets[i] = momentum * targets[i] + (1 - momentum) * sources[i], one multi-tensor pass (fp32 tensors)."""
    _dev(*targets, *sources)
    n = len(targets)
    if n == 0:
        return
    if len(sources) != n:
        raise VtxError("vtx: ema_update needs as many sources as targets")
    for t, s in zip(targets, sources):
        if t.dtype != torch.float32 or s.dtype != torch.float32 or t.numel() != s.numel():
            raise VtxError("vtx: ema_update needs fp32 tensor pairs of equal size "
                           f"(got {t.dtype} {tuple(t.shape)} / {s.dtype} {tuple(s.shape)})")
    numel = (ctypes.c_int64 * n)(*[t.numel() for t in targets])
    check(_lib.load().vtx_ema_update(n, _ptr_array(targets), _ptr_array(sources), numel, float(momentum), _stream()),
          "vtx_ema_update")


def dino_loss(student, teacher, center, n_crop, student_temp, teacher_temp, gscale=1.0):
    """DINO loss forward + gradient: -> (loss scalar tensor, dstudent, batch_center [K] fp32)."""
    _dev(student, teacher, center)
    _f32(center, "center")
    K = student.shape[-1]
    rows = student.numel() // K
    B = teacher.numel() // K // 2
    if rows != n_crop * B or student.dtype != teacher.dtype:
        raise VtxError("vtx: dino_loss shape / dtype mismatch")
    lib = _lib.load()
    wsb = lib.vtx_dino_loss_workspace(B, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=student.device)
    loss_rows = torch.empty(rows, dtype=torch.float32, device=student.device)
    ds = torch.empty_like(student)
    bc = torch.empty(K, dtype=torch.float32, device=student.device)
    check(lib.vtx_dino_loss(_p(student), _p(teacher), _p(center), _p(ws), wsb, _p(loss_rows), _p(ds), _p(bc), n_crop, B, K,
                            float(student_temp), float(teacher_temp), float(gscale), _dt(student), _stream()),
          "vtx_dino_loss")
    return loss_rows.sum() / ((2 * n_crop - 2) * B), ds, bc


def mix_plan_bytes():
    return _lib.load().vtx_mix_plan_bytes()


def mix_max_rects():
    return _lib.load().vtx_mix_max_rects()


def mix_normalize_erase(images, plan, mean, std, fills=None, nhwc_bf16=False):
    """One pass over a device batch: mixup / cutmix with the partner image, normalise, RandomErasing (zeros, or the
    host-drawn normal values in ``fills`` for the 'rand' / 'pixel' modes).  -> fp32 (N, C, H, W), or with ``nhwc_bf16``
    a bf16 tensor of SHAPE (N, C, H, W) in torch.channels_last memory (physically [N, H, W, C]) -- the layout the
    patch-embedding gather of the models reads directly."""
    _dev(images, plan, mean, std, fills)
    if images.dtype not in (torch.uint8, torch.float32):
        raise VtxError("vtx: input images must be uint8 or float32")
    x = images if images.is_contiguous() else images.contiguous()
    n, c, h, w = x.shape
    if nhwc_bf16:
        out = torch.empty((n, c, h, w), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    else:
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(_lib.load().vtx_mix_normalize_erase(_p(x), int(x.dtype == torch.uint8), _p(plan), _p(mean), _p(std), _p(fills),
                                              _p(out), int(nhwc_bf16), n, c, h, w, _stream()), "vtx_mix_normalize_erase")
    return out


def mix_loss(logits, label1, label2, ratio, eps, reduction="mean"):
    """MixLoss value and its gradient w.r.t. the logits, one kernel.  reduction 'mean': (scalar, d mean / d logits);
    'sum' (the reference treats every other string as sum, loss.py:77-84): (scalar, d sum / d logits); 'none':
    (per-sample losses [B], the per-row Jacobian softmax - target -- the caller scales row b by the incoming gradient)."""
    _dev(logits, label1, label2, ratio)
    x = logits if logits.is_contiguous() else logits.contiguous()
    B, K = x.shape
    l1 = label1.to(torch.int64).contiguous()
    l2 = label2.to(torch.int64).contiguous()
    r = torch.as_tensor(ratio, device=x.device).to(torch.float32).expand(B).contiguous()
    dl = torch.empty_like(x)
    rows = torch.empty(B, dtype=torch.float32, device=x.device)
    gscale = 1.0 if reduction == "mean" else float(B)          # the kernel writes gscale / B * (softmax - target)
    check(_lib.load().vtx_mix_loss(_p(x), _p(l1), _p(l2), _p(r), _p(dl), _p(rows), B, K, float(eps), gscale, _dt(x), _stream()),
          "vtx_mix_loss")
    if reduction == "none":
        return rows, dl
    return (rows.sum() / B if reduction == "mean" else rows.sum()), dl


def l2norm_fwd(x, eps=1e-12):
    """y = x / max(||x||_2, eps) over the last dim; returns (y, row norms)."""
    _dev(x)
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    nrm = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_lib.load().vtx_l2norm_fwd(_p(x), _p(y), _p(nrm), rows, C, float(eps), _dt(x), _stream()), "vtx_l2norm_fwd")
    return y, nrm


def l2norm_bwd(dy, y, nrm):
    _dev(dy, y, nrm)
    C = y.shape[-1]
    dx = torch.empty_like(y)
    check(_lib.load().vtx_l2norm_bwd(_p(dy), _p(y), _p(nrm), _p(dx), y.numel() // C, C, _dt(y), _stream()), "vtx_l2norm_bwd")
    return dx


# ------------------------------------------------------------------------------- data movement
def cast_desc_bytes():
    return _lib.load().vtx_cast_desc_bytes()


def cast_weights(desc, nmat, ntiles, flat, flat_t):
    """Multi-tensor fp32 -> bf16 weight cast (plain + transposed copies), one launch (csrc/cast.hip)."""
    _dev(desc, flat, flat_t)
    with _timed("cast_weights_kernel", 0.0, 8.0 * flat.numel()):
        check(_lib.load().vtx_cast_weights(_p(desc), nmat, ntiles, _p(flat), _p(flat_t), _stream()), "vtx_cast_weights")


def is_nhwc_bf16(x):
    """A (B, C, H, W) bf16 tensor stored channels-last (physically [B, H, W, C]) -- what DeviceMixPipeline(output=
    "nhwc_bf16") hands to the models."""
    return (x.dim() == 4 and x.dtype == torch.bfloat16 and x.shape[1] > 1 and
            x.is_contiguous(memory_format=torch.channels_last))


def patch_gather(x_nchw, patch, order, dtype, kp=None):
    """Image batch -> [B, H/p, W/p, Kp] patch matrix of `dtype` (order 0: Swin (py,px,c); 1: ViT (c,py,px)).  The image
    is (B, C, H, W) fp32 contiguous (the reference's input contract) or bf16 channels-last (is_nhwc_bf16: the device
    input pipeline's output, read without a layout pass)."""
    if is_nhwc_bf16(x_nchw):
        B, Cin, H, W = x_nchw.shape
        K = Cin * patch * patch
        kp = K if kp is None else kp
        out = torch.empty((B, H // patch, W // patch, kp), dtype=dtype, device=x_nchw.device)
        with _timed("patch_gather_nhwc_kernel", 0.0, x_nchw.numel() * 2.0 + out.numel() * out.element_size()):
            check(_lib.load().vtx_patch_gather_nhwc(x_nchw.data_ptr(), _p(out), B, Cin, H, W, patch, kp, order, _dt(out),
                                                    _stream()), "vtx_patch_gather_nhwc")
        return out
    _dev(x_nchw)
    if x_nchw.dtype != torch.float32:
        x_nchw = x_nchw.float()
    B, Cin, H, W = x_nchw.shape
    K = Cin * patch * patch
    kp = K if kp is None else kp
    out = torch.empty((B, H // patch, W // patch, kp), dtype=dtype, device=x_nchw.device)
    with _timed("patch_gather_kernel", 0.0, x_nchw.numel() * 4.0 + out.numel() * out.element_size()):
        check(_lib.load().vtx_patch_gather(_p(x_nchw), _p(out), B, Cin, H, W, patch, kp, order, _dt(out), _stream()),
              "vtx_patch_gather")
    return out


def token_mean_fwd(x, B, Tn, C):
    _dev(x)
    y = torch.empty((B, C), dtype=x.dtype, device=x.device)
    check(_lib.load().vtx_token_mean_fwd(_p(x), _p(y), B, Tn, C, _dt(x), _stream()), "vtx_token_mean_fwd")
    return y


def token_mean_bwd(dy, B, Tn, C, shape):
    _dev(dy)
    dx = torch.empty(shape, dtype=dy.dtype, device=dy.device)
    check(_lib.load().vtx_token_mean_bwd(_p(dy), _p(dx), B, Tn, C, _dt(dy), _stream()), "vtx_token_mean_bwd")
    return dx


def vit_assemble_fwd(patches, cls, pos):
    _dev(patches, cls, pos)
    _f32(cls, "cls_token"); _f32(pos, "pos_embed")
    B, n, C = patches.shape
    out = torch.empty((B, n + 1, C), dtype=patches.dtype, device=patches.device)
    check(_lib.load().vtx_vit_assemble_fwd(_p(patches), _p(cls), _p(pos), _p(out), B, n + 1, C, _dt(patches),
                                           _stream()), "vtx_vit_assemble_fwd")
    return out


def vit_assemble_bwd(dx):
    _dev(dx)
    B, L, C = dx.shape
    dpatches = torch.empty((B, L - 1, C), dtype=dx.dtype, device=dx.device)
    dcls = torch.empty(C, dtype=torch.float32, device=dx.device)
    dpos = torch.empty((L, C), dtype=torch.float32, device=dx.device)
    check(_lib.load().vtx_vit_assemble_bwd(_p(dx), _p(dpatches), _p(dcls), _p(dpos), B, L, C, _dt(dx), _stream()),
          "vtx_vit_assemble_bwd")
    return dpatches, dcls, dpos


# ------------------------------------------------------------------------------- attention cores
def relpos_bias(rel_pos, pos, n_head):
    """bias[h][a][b] = rel_pos[pos[a][b]][h]  (swin_transformer.py:135-136)."""
    _dev(rel_pos, pos)
    _f32(rel_pos, "rel_pos")
    if pos.dtype != torch.int64:
        raise VtxError("vtx: pos must be int64 (the reference's buffer dtype)")
    L = pos.shape[0]
    bias = torch.empty((n_head, L, L), dtype=torch.float32, device=rel_pos.device)
    check(_lib.load().vtx_relpos_bias(_p(rel_pos), _p(pos), _p(bias), L, n_head, _stream()), "vtx_relpos_bias")
    return bias


def pos_csr(pos, ntab):
    """Host-side CSR of the pos table: (a,b) pairs grouped by table index (for the dense rel_pos gradient)."""
    flat = pos.reshape(-1).cpu()
    order = torch.argsort(flat, stable=True).to(torch.int32)
    counts = torch.bincount(flat, minlength=ntab)
    offsets = torch.zeros(ntab + 1, dtype=torch.int32)
    offsets[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return order, offsets


def _sattn_cfg(L=197):
    """(tiles per wave step, waves) template arguments of the ViT attention kernels for sequences of L tokens (asks attention_seq.hip)."""
    w = _lib.load().vtx_sattn_waves(int(L))
    return "2, 4" if w == 4 else f"1, {w}"


def attention_fwd(qkv, B, L, n_head, D, swin=None, bias=None, mask=None, drop=None):
    """o [rows, h*D], lse.  swin = (H, W, win, shift) for window attention, None for global.
    drop = (p, seed, keep): dropout of the attention probabilities (vit.py:39, swin_transformer.py:144) on the register-resident
    kernels; keep: uint8 [problems, L, L] replaces the hash (parity tests)."""
    _dev(qkv, bias, mask)
    H, W, win, shift = swin if swin is not None else (0, 0, 0, 0)
    rows = qkv.numel() // (3 * n_head * D)
    nW = (H // win) * (W // win) if swin is not None else 1
    if rows != B * nW * L:
        raise VtxError("vtx: attention rows mismatch")
    o = torch.empty(qkv.shape[:-1] + (n_head * D,), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B * nW * n_head * L, dtype=torch.float32, device=qkv.device)
    fast = drop is None and swin is None and bias is None and qkv.dtype == torch.bfloat16 and D == 64 and L <= 224      # mirrors sattn_ok
    nkt = 4 if L <= 64 else (8 if L <= 128 else 14)
    long_ = drop is None and swin is None and bias is None and mask is None and L > 224
    ev = _attn_bracket(f"sattn_fwd_kernel<{nkt}, {_sattn_cfg(L)}>" if fast else ("lattn_fwd_kernel" if long_ else "attn_fwd_kernel"),
                       B * nW * n_head, L, D, rows, n_head * D, qkv.element_size(), False)
    if drop is not None:
        dp, seed, keep = _drop_args(drop, qkv.device, B * nW * n_head * L * L)
        check(_lib.load().vtx_attention_fwd_drop(_p(qkv), _p(o), _p(lse), _p(bias), _p(mask), B, L, n_head, D,
                                                 int(swin is not None), H, W, win, int(bool(shift)), _dt(qkv), dp, seed, _p(keep),
                                                 _stream()), "vtx_attention_fwd_drop")
    else:
        check(_lib.load().vtx_attention_fwd(_p(qkv), _p(o), _p(lse), _p(bias), _p(mask), B, L, n_head, D,
                                            int(swin is not None), H, W, win, int(bool(shift)), _dt(qkv), _stream()),
              "vtx_attention_fwd")
    if ev:
        ev[1].record()
    return o, lse


def attention_bwd(qkv, o, dout, lse, B, L, n_head, D, swin=None, bias=None, mask=None, csr=None, ntab=0, drop=None):
    """dqkv, drel_pos (None without bias); drop: the forward's (p, seed, keep)."""
    _dev(qkv, o, dout, lse, bias, mask)
    lib = _lib.load()
    H, W, win, shift = swin if swin is not None else (0, 0, 0, 0)
    dqkv = torch.empty_like(qkv)
    drel, ws, wsb, order, offsets = None, None, 0, None, None
    if bias is not None:
        order, offsets = csr
        _dev(order, offsets)
        drel = torch.empty((ntab, n_head), dtype=torch.float32, device=qkv.device)
    if bias is not None or (swin is None and L > 224):          # bias-gradient slabs / the long kernels' Dq vector
        wsb = lib.vtx_attention_bwd_workspace(B, L, n_head, int(swin is not None), H, W, max(win, 1))
        ws = torch.empty(wsb, dtype=torch.uint8, device=qkv.device)
    rows = qkv.numel() // (3 * n_head * D)
    fast = drop is None and swin is None and bias is None and qkv.dtype == torch.bfloat16 and D == 64 and L <= 224
    nkt = 4 if L <= 64 else (8 if L <= 128 else 14)
    long_ = drop is None and swin is None and bias is None and mask is None and L > 224
    ev = _attn_bracket(f"sattn_bwd_kernel<{nkt}, {_sattn_cfg(L)}>" if fast else ("lattn_bwd_*_kernel" if long_ else "attn_bwd_kernel"),
                       rows // L * n_head, L, D, rows, n_head * D, qkv.element_size(), True)
    if drop is not None:
        dp, seed, keep = _drop_args(drop, qkv.device, rows * n_head * L)
        check(lib.vtx_attention_bwd_drop(_p(qkv), _p(o), _p(dout), _p(lse), _p(bias), _p(mask), _p(order), _p(offsets),
                                         _p(dqkv), _p(drel), ntab, _p(ws), wsb, B, L, n_head, D, int(swin is not None),
                                         H, W, win, int(bool(shift)), _dt(qkv), dp, seed, _p(keep), _stream()),
              "vtx_attention_bwd_drop")
    else:
        check(lib.vtx_attention_bwd(_p(qkv), _p(o), _p(dout), _p(lse), _p(bias), _p(mask), _p(order), _p(offsets),
                                    _p(dqkv), _p(drel), ntab, _p(ws), wsb, B, L, n_head, D, int(swin is not None),
                                    H, W, win, int(bool(shift)), _dt(qkv), _stream()), "vtx_attention_bwd")
    if ev:
        ev[1].record()
    return dqkv, drel


# ------------------------------------------------------------------------------- window attention fast path
def wattn_supported(D, win):
    return D == 32 and win <= 7


def wattn_fwd(qkv, rel_pos, pos, region, B, L, n_head, swin):
    """Window attention forward: region = None (un-shifted) or the uint8 [nW, 64] ids of tables.mask_regions."""
    _dev(qkv, rel_pos, pos, region)
    _f32(rel_pos, "rel_pos")
    if pos.dtype != torch.int64:
        raise VtxError("vtx: pos must be int64 (the reference's buffer dtype)")
    H, W, win, shift = swin
    o = torch.empty(qkv.shape[:-1] + (n_head * 32,), dtype=qkv.dtype, device=qkv.device)
    nW = (H // win) * (W // win)
    lse = torch.empty(B * nW * n_head * L, dtype=torch.float32, device=qkv.device)
    ev = _attn_bracket(wattn_fwd_kernel_name(qkv.dtype, region is not None, B * nW), B * nW * n_head, L, 32,
                       B * nW * L, n_head * 32, qkv.element_size(), False)
    check(_lib.load().vtx_wattn_fwd(_p(qkv), _p(o), _p(lse), _p(rel_pos), _p(pos), _p(region), B, L, n_head, H, W, win,
                                    int(bool(shift)), _dt(qkv), _stream()), "vtx_wattn_fwd")
    if ev:
        ev[1].record()
    return o, lse


_POS_INVERSE = {}


def _pos_inverse(pos, ntab):
    """Device copy of tables.pos_inverse(pos), built once per pos buffer (host argsort + one upload, first backward only).
    Keyed by the buffer's address; valid while the tensor it was built from is alive (a module buffer, never written)."""
    key = (pos.data_ptr(), pos.device, tuple(pos.shape), ntab)
    hit = _POS_INVERSE.get(key)
    if hit is None or hit[0]() is None:
        import weakref
        from . import tables
        for k in [k for k, v in _POS_INVERSE.items() if v[0]() is None]:      # maps of buffers that no longer exist
            del _POS_INVERSE[k]
        cells, count = tables.pos_inverse(pos, ntab)
        hit = (weakref.ref(pos), cells.to(pos.device), count)
        _POS_INVERSE[key] = hit
    return hit[1], hit[2]


def wattn_bwd(qkv, o, dout, lse, rel_pos, pos, region, B, L, n_head, swin, ntab, defer=False, use_inverse=True):
    """-> dqkv, drel_pos [ntab, n_head] -- or with ``defer`` (dqkv, Partials) for a later colreduce_multi.
    ``use_inverse=False`` (tests) takes the kernel's LDS-atomic scatter of the rel_pos gradient instead of the gather over
    the inverse pos map."""
    _dev(qkv, o, dout, lse, rel_pos, pos, region)
    lib = _lib.load()
    inv_cells, inv_count = _pos_inverse(pos, ntab) if use_inverse else (None, 0)
    H, W, win, shift = swin
    dqkv = torch.empty_like(qkv)
    drel = None if defer else torch.empty((ntab, n_head), dtype=torch.float32, device=qkv.device)
    wsb = lib.vtx_wattn_bwd_workspace(B, n_head, H, W, win)
    ws = torch.empty(wsb, dtype=torch.uint8, device=qkv.device)
    nW = (H // win) * (W // win)
    tn = "__bf16" if qkv.dtype == torch.bfloat16 else "float"
    ev = _attn_bracket(wattn_bwd_kernel_name(qkv.dtype, region is not None, use_inverse), B * nW * n_head, L, 32,
                       B * nW * L, n_head * 32, qkv.element_size(), True)     # (+ the small drel_pos column reduce)
    check(lib.vtx_wattn_bwd(_p(qkv), _p(o), _p(dout), _p(lse), _p(rel_pos), _p(pos), _p(region), _p(dqkv), _p(drel),
                            _p(ws), wsb, _p(inv_cells), inv_count, B, L, n_head, H, W, win, int(bool(shift)), _dt(qkv),
                            _stream()),
          "vtx_wattn_bwd")
    if ev:
        ev[1].record()
    if defer:
        return dqkv, Partials(ws, lib.vtx_wattn_bwd_parts(B, n_head, H, W, win), ntab * n_head, lib.vtx_wattn_bwd_part_ld(n_head), False)
    return dqkv, drel
