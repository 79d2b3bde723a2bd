"""-m gpu: kernels must not depend on what a previous workgroup left in the LDS.

LDS is not cleared between workgroups.  Round 5 found (on one box, in the very first launch of a process) non-finite dK from the ViT
attention backward at L = 197: phase A never writes D[q] of the query tile that holds no query (rows 208 .. 223), phase B multiplies
(dP - D) of those rows by p = 0, and 0 x (whatever the LDS held) is NaN when that happens to be a NaN or an infinity.  Here every
LDS-using kernel family runs twice -- after `vtx_debug_lds_poison` filled the LDS of every CU with zeros, and after it filled it with
NaN / -inf bit patterns -- and must give the same bits, all finite."""
import pytest
import torch

from gpu_util import dev

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
PATTERNS = (0x7FC07FC0, 0xFF80FF80, 0x7F807F80)      # bf16 NaN pairs = fp32 NaN; bf16 -inf pairs (fp32 NaN); bf16 +inf pairs (fp32 NaN)


def _mk(shape, seed, dtype, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(dtype)


def _poison(pattern):
    from vtx import _lib, ops
    ops.check(_lib.load().vtx_debug_lds_poison(pattern, 3, ops._stream()), "vtx_debug_lds_poison")


def _flat(out):
    if isinstance(out, torch.Tensor):
        return [out]
    res = []
    for o in out:
        if o is not None:
            res.extend(_flat(o))
    return res


def _check(name, fn):
    dev()
    _poison(0)
    ref = [t.clone() for t in _flat(fn())]
    assert ref, name
    for t in ref:
        assert torch.isfinite(t.float()).all(), f"{name}: non-finite output after a ZERO fill"
    for pat in PATTERNS:
        _poison(pat)
        got = _flat(fn())
        for i, (a, b) in enumerate(zip(ref, got)):
            assert torch.isfinite(b.float()).all(), f"{name}: output {i} has non-finite values after LDS pattern {pat:#x}"
            assert torch.equal(a, b), f"{name}: output {i} depends on the LDS contents (pattern {pat:#x})"
    # ... and with every torch.empty() buffer (outputs, workspaces) NaN-filled by torch's deterministic-mode debug fill: every element of
    # every output is written, nothing is read before it is written
    prev = torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled()
    torch.use_deterministic_algorithms(True, warn_only=True)
    try:
        _poison(PATTERNS[0])
        got = _flat(fn())
    finally:
        torch.use_deterministic_algorithms(prev[0], warn_only=prev[1])
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), f"{name}: output {i} differs when torch.empty() buffers start as NaN"


@pytest.mark.parametrize("L,nH,D,dtype", [(197, 6, 64, BF), (37, 6, 64, BF), (50, 3, 64, BF), (130, 2, 64, BF), (193, 2, 64, BF), (224, 2, 64, BF),
                                          (209, 2, 64, BF), (65, 2, 64, BF), (197, 3, 64, torch.float32), (100, 4, 32, BF), (257, 2, 64, BF),
                                          (40, 2, 32, torch.float32)])
def test_global_attention_forward_and_backward(L, nH, D, dtype):
    from vtx import ops
    B = 24
    qkv, do = _mk((B, L, 3 * nH * D), 1, dtype), _mk((B, L, nH * D), 2, dtype)

    def run():
        o, lse = ops.attention_fwd(qkv, B, L, nH, D)
        return o, lse, ops.attention_bwd(qkv, o, do, lse, B, L, nH, D)

    _check(f"attention L {L} D {D} {dtype}", run)
    drop = (0.1, 1234, None)                               # attention-probability dropout (hash mask): the *_drop entry points

    def run_drop():
        o, lse = ops.attention_fwd(qkv, B, L, nH, D, drop=drop)
        return o, lse, ops.attention_bwd(qkv, o, do, lse, B, L, nH, D, drop=drop)

    _check(f"attention with dropout L {L} D {D} {dtype}", run_drop)


@pytest.mark.parametrize("H,win,nH,shift,dtype,B", [(14, 7, 4, True, BF, 16), (28, 7, 2, False, BF, 8), (56, 7, 3, True, BF, 128), (16, 4, 2, True, BF, 8),
                                                    (20, 5, 3, False, BF, 4), (12, 6, 2, True, BF, 4), (14, 7, 2, True, torch.float32, 8),
                                                    (24, 12, 2, True, BF, 2), (16, 8, 2, False, BF, 4)])
def test_window_attention_forward_and_backward(H, win, nH, shift, dtype, B):
    from oracle import tables
    from vtx import ops, options
    from vtx.tables import mask_regions
    d = dev()
    L, ntab = win * win, (2 * win - 1) ** 2
    pos_np, mask_np = tables.make_pos_mask((H, H), win, shift)
    pos = torch.from_numpy(pos_np).to(d)
    qkv, do = _mk((B, H, H, 3 * nH * 32), 3, dtype), _mk((B, H, H, nH * 32), 4, dtype)
    rel = _mk((ntab, nH), 5, torch.float32, 0.5)
    if ops.wattn_supported(32, win):
        region = mask_regions(torch.from_numpy(mask_np).to(d))[0] if shift else None
        swin = (H, H, win, shift)

        def run():
            o, lse = ops.wattn_fwd(qkv, rel, pos, region, B, L, nH, swin)
            return o, lse, ops.wattn_bwd(qkv, o, do, lse, rel, pos, region, B, L, nH, swin, ntab)

        for f4 in (0, 2):
            with options.override(WATTN_FWD4=f4, WATTN_BWD4=1 if f4 else 0):
                _check(f"wattn {H}x{H} win {win} {dtype} four-wave kernels {bool(f4)}", run)
    else:
        bias = ops.relpos_bias(rel, pos, nH)
        mask = torch.from_numpy(mask_np).to(d) if shift else None
        csr = tuple(t.to(d) for t in ops.pos_csr(torch.from_numpy(pos_np), ntab))

        def run():
            o, lse = ops.attention_fwd(qkv, B, L, nH, 32, swin=(H, H, win, shift), bias=bias, mask=mask)
            return o, lse, ops.attention_bwd(qkv, o, do, lse, B, L, nH, 32, swin=(H, H, win, shift), bias=bias, mask=mask, csr=csr, ntab=ntab)

        _check(f"window attention {H}x{H} win {win} (generic kernels)", run)


@pytest.mark.parametrize("Lq,Lk,nH,D", [(3136, 49, 1, 64), (784, 49, 2, 64), (196, 49, 5, 64), (49, 49, 8, 64), (196, 4, 8, 32), (784, 100, 2, 64),
                                        (100, 196, 2, 32), (64, 144, 2, 32)])
def test_subsampled_and_cross_attention(Lq, Lk, nH, D):
    from vtx import ops
    B = 6
    q, kv, do = _mk((B * Lq, nH * D), 6, BF), _mk((B * Lk, 2 * nH * D), 7, BF), _mk((B * Lq, nH * D), 8, BF)

    def run():
        o, lse = ops.srattn_fwd(q, kv, B, Lq, Lk, nH)
        return o, lse, ops.srattn_bwd(q, kv, o, do, lse, B, Lq, Lk, nH)

    _check(f"srattn Lq {Lq} Lk {Lk} D {D}", run)
    drop = (0.1, 4321, None)

    def run_drop():
        o, lse = ops.srattn_fwd(q, kv, B, Lq, Lk, nH, drop=drop)
        return o, lse, ops.srattn_bwd(q, kv, o, do, lse, B, Lq, Lk, nH, drop=drop)

    _check(f"srattn with dropout Lq {Lq} Lk {Lk} D {D}", run_drop)
    if D == 32:
        bias = _mk((nH, Lq, Lk), 9, torch.float32, 0.3)

        def runx():
            o, lse = ops.xattn_fwd(q, kv, B, Lq, Lk, nH, bias=bias)
            return o, lse, ops.xattn_bwd(q, kv, o, do, lse, B, Lq, Lk, nH, bias=bias)

        _check(f"xattn Lq {Lq} Lk {Lk}", runx)


@pytest.mark.parametrize("rows,C", [(25088, 384), (6272, 768), (12345, 96), (4000, 192), (3000, 320), (777, 64), (1000, 1024)])
def test_layernorm(rows, C):
    from vtx import ops
    x, dy, dres = _mk((rows, C), 10, BF), _mk((rows, C), 11, BF), _mk((rows, C), 12, BF)
    gamma, beta = _mk((C,), 13, torch.float32, 0.2) + 1, _mk((C,), 14, torch.float32, 0.1)

    def run():
        y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5)
        return y, mean, rstd, ops.layernorm_bwd(dy, x, mean, rstd, gamma, dres=dres)

    _check(f"layernorm {rows} x {C}", run)


@pytest.mark.parametrize("M,N,K", [(25088, 384, 1536), (25088 + 77, 1536, 384), (100352, 576, 192), (66395, 96, 96), (5000, 256, 1024), (3000, 320, 1280),
                                   (401408 // 8, 96, 384), (12800, 128, 1024), (1000, 200, 104)])
def test_gemm_and_weight_gradient(M, N, K):
    from vtx import ops
    x, w, b = _mk((M, K), 15, BF), _mk((N, K), 16, BF, 0.05), _mk((N,), 17, torch.float32, 0.1)
    dy, res = _mk((M, N), 18, BF), _mk((M, N), 19, BF)

    def run():
        outs = [ops.gemm(x, w, 0, bias=b), ops.gemm(x, w, 0, bias=b, resid=res), ops.gemm(x, w, 0, bias=b, act=ops.ACT_SILU, want_aux=True)]
        if N % 8 == 0 and K % 8 == 0:
            outs.append(ops.wgrad(dy, x))
        return outs

    _check(f"gemm / wgrad {M} x {N} x {K}", run)


@pytest.mark.parametrize("family", ["swin", "vit", "vit_crops", "pvt", "twins", "halo"])
@pytest.mark.parametrize("layer_call,bf16", [(False, True), (True, True), (False, False)], ids=["call_by_call", "one_call_layers", "fp32"])
def test_whole_models_with_the_lds_poisoned_behind_every_library_call(family, layer_call, bf16, monkeypatch):
    """Forward + backward of a small model of every family with the LDS of every CU refilled (NaN / +-inf patterns in turn) behind EVERY
    call into libvtx -- and every torch.empty() buffer NaN-filled -- logits and all gradients must equal, bit for bit, the run with zero
    fills.  Call by call every kernel of the model starts on poisoned LDS; with the one-call layers (vtx_layer_fwd / _bwd: several launches
    per call) every layer does."""
    from models import HaloTransformer, SwinTransformer, VisionTransformer
    from models.pvt import PyramidVisionTransformer
    from models.twins import TwinsSVT
    from oracle import ref_models as M
    from vtx import _lib, ops
    from vtx import functional as VF
    from vtx.nn import Linear
    d = dev()
    torch.manual_seed(51)
    if family == "swin":
        model = SwinTransformer(image_size=(224, 224), n_class=16, depths=(1, 1, 2, 1), dims=(64, 128, 384, 768), dim_head=32,
                                n_heads=(2, 4, 12, 24), dim_ffs=(256, 512, 1536, 3072), window_size=7, drop_path=0.0)
        x = torch.randn(4, 3, 224, 224, device=d)
    elif family.startswith("vit"):
        model = VisionTransformer(Linear(384, 16), 224, 16, 2, 384, 6, 1536, 0.0, 0.0, 0.0, 0.0)
        x = torch.randn(6, 3, 224, 224, device=d)                # L = 197: 13 live tiles of 14
        if family == "vit_crops":
            x = [x, torch.randn(12, 3, 96, 96, device=d)]        # L = 37: 3 live tiles of 4
    elif family == "pvt":
        cfg = dict(M.PVT_SMALL); cfg["depths"] = (1, 1, 1, 1); cfg["n_class"] = 16
        model = PyramidVisionTransformer(**cfg, drop_path=0.0)
        x = torch.randn(3, 3, 224, 224, device=d)
    elif family == "twins":
        cfg = dict(M.TWINS_SVT_S); cfg["depths"] = (1, 1, 1, 1); cfg["n_class"] = 16
        model = TwinsSVT(**cfg)
        x = torch.randn(3, 3, 224, 224, device=d)
    else:
        model = HaloTransformer(**M.HALO_TINY)
        x = torch.randn(2, 3, 224, 224, device=d)
    model.to(d).train()
    monkeypatch.setattr(VF, "_LAYER_CALL", layer_call)
    lib = _lib.load()
    real = _lib.check
    state = dict(pattern=0, n=0, busy=False)

    def check_then_poison(code, what):
        real(code, what)
        if state["busy"] or "option" in what:
            return
        state["busy"] = True
        try:
            pat = state["pattern"] if state["pattern"] == 0 else PATTERNS[state["n"] % len(PATTERNS)]
            state["n"] += 1
            real(lib.vtx_debug_lds_poison(pat, 1, ops._stream()), "vtx_debug_lds_poison")
        finally:
            state["busy"] = False

    monkeypatch.setattr(_lib, "check", check_then_poison)
    monkeypatch.setattr(ops, "check", check_then_poison)

    def run(pattern):
        state["pattern"], state["n"] = pattern, 0
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=BF, enabled=bf16):
            out = model(x)
        out.float().square().mean().backward()
        torch.cuda.synchronize()
        return out.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, state["n"]

    out0, g0, n0 = run(0)
    # second run: poisoned LDS AND every torch.empty() buffer filled with NaN (torch's deterministic-mode debug fill): nothing the
    # library allocates as an output or a workspace may be read before it is written
    prev = torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled()
    torch.use_deterministic_algorithms(True, warn_only=True)
    assert torch.utils.deterministic.fill_uninitialized_memory
    try:
        assert torch.isnan(torch.empty(1024, device=d)).all(), "the debug fill of torch.empty is not active"
        out1, g1, n1 = run(1)
    finally:
        torch.use_deterministic_algorithms(prev[0], warn_only=prev[1])
    assert n0 == n1 and n0 > 8, f"only {n0} library calls were intercepted"
    assert torch.isfinite(out1).all(), f"{family}: non-finite logits with poisoned LDS"
    assert torch.equal(out0, out1), f"{family}: logits depend on the LDS contents"
    assert g0.keys() == g1.keys() and len(g0) > 10
    for k in g0:
        assert torch.isfinite(g1[k]).all(), f"{family}: gradient of {k} is not finite with poisoned LDS"
        assert torch.equal(g0[k], g1[k]), f"{family}: gradient of {k} depends on the LDS contents"


@pytest.mark.parametrize("family", ["swin", "dino"])
def test_train_steps_with_the_lds_poisoned_behind_every_library_call(family, monkeypatch):
    """Two full bf16 train steps (forward, backward, MixLoss / DINOLoss, clip, fused AdamW, weight casts, EMA teacher) of two identical
    copies of a model -- one with zero fills, one with NaN / inf patterns behind every library call and NaN-filled torch.empty()
    buffers: every parameter must end up with the same bits."""
    import copy

    from vtx import _lib, ops
    from vtx.optim import FusedAdamW
    from vtx.train_step import MixLoss, make_param_groups, train_step
    d = dev()
    torch.manual_seed(61)
    g = torch.Generator(device="cuda").manual_seed(62)
    if family == "swin":
        from models import SwinTransformer
        base = SwinTransformer(image_size=(224, 224), n_class=16, depths=(1, 1, 2, 1), dims=(32, 64, 128, 256), dim_head=32,
                               n_heads=(1, 2, 4, 8), dim_ffs=(128, 256, 512, 1024), window_size=7, drop_path=0.0).to(d).train()
        x = torch.randn(8, 3, 224, 224, device=d, generator=g)
        l1 = torch.arange(8, device=d)
        data = (x, l1, l1.roll(1), torch.rand(8, device=d, generator=g))
    else:
        from models.vit import dino
        from vtx.dino import DINOLoss, dino_train_step
        mk = lambda: dino(224, 16, 2, 128, 2, 512, 0.0, 0.0, 0.0, 0.0, 1024, depth_head=3, dim_head_ff=256, dim_head_bottleneck=64).to(d).train()
        base = mk()
        crops = [torch.randn(4, 3, 224, 224, device=d, generator=g) for _ in range(2)] + [torch.randn(4, 3, 96, 96, device=d, generator=g) for _ in range(3)]
    lib = _lib.load()
    real = _lib.check
    state = dict(pattern=0, n=0, busy=False)

    def check_then_poison(code, what):
        real(code, what)
        if state["busy"] or "option" in what:
            return
        state["busy"] = True
        try:
            pat = state["pattern"] if state["pattern"] == 0 else PATTERNS[state["n"] % len(PATTERNS)]
            state["n"] += 1
            real(lib.vtx_debug_lds_poison(pat, 1, ops._stream()), "vtx_debug_lds_poison")
        finally:
            state["busy"] = False

    monkeypatch.setattr(_lib, "check", check_then_poison)
    monkeypatch.setattr(ops, "check", check_then_poison)

    def run(pattern):
        state["pattern"], state["n"] = pattern, 0
        torch.manual_seed(63)
        if family == "swin":
            model = copy.deepcopy(base)
        else:                                   # (weight_norm parametrisation: no deepcopy)
            model, teacher = mk(), mk()
            model.load_state_dict(base.state_dict())
            teacher.load_state_dict(base.state_dict())
        if family == "swin":
            opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
            crit = MixLoss(0.1)
            losses = [train_step(model, crit, opt, data, clip_grad_norm=5.0, autocast_dtype=BF) for _ in range(2)]
            extra = {}
        else:
            for p in teacher.parameters():
                p.requires_grad = False
            crit = DINOLoss(1024, 5, 0.04, 0.07, 30, 300).to(d)
            opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.04, "dino"), lr=5e-4)
            losses = [dino_train_step(model, teacher, crit, opt, crops, epoch=1, momentum=0.99, clip_grad_norm=3.0, freeze_last_layer=1,
                                      autocast_dtype=BF) for _ in range(2)]
            extra = {"teacher." + n: p.detach().clone() for n, p in teacher.named_parameters()}
            extra["center"] = crit.center.detach().clone()
        torch.cuda.synchronize()
        res = {n: p.detach().clone() for n, p in model.named_parameters()}
        res.update(extra)
        res["losses"] = torch.stack([l.detach().float().reshape(()) for l in losses])
        return res, state["n"]

    a, na = run(0)
    prev = torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled()
    torch.use_deterministic_algorithms(True, warn_only=True)
    try:
        b, nb = run(1)
    finally:
        torch.use_deterministic_algorithms(prev[0], warn_only=prev[1])
    assert na == nb and na > 20
    assert a.keys() == b.keys()
    for k in a:
        assert torch.isfinite(b[k].float()).all(), f"{family}: {k} is not finite after two poisoned train steps"
        assert torch.equal(a[k], b[k]), f"{family}: {k} depends on LDS / uninitialised-buffer contents"


@pytest.mark.parametrize("name,batch,drop_path", [("swin_s", 128, 0.3), ("vit_s16", 256, 0.1), ("pvt_small", 128, 0.1), ("twins_svt_s", 128, 0.1)])
def test_benchmark_configurations_with_the_lds_poisoned_behind_every_library_call(name, batch, drop_path, monkeypatch):
    """The benchmark's own models at its batch sizes (the dispatch of bench.py: two-group / A-stationary / streaming GEMMs, wide weight
    gradients, four-wave window attention, stochastic-depth compaction with host-drawn masks): forward + backward with poisoned LDS and
    NaN-filled buffers against the run with zero fills, same DropPath masks -- logits and gradients bit for bit."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from vtx import _lib, ops
    d = dev()
    torch.manual_seed(71)
    model = bench.build_model(name, drop_path).to(d).train()
    x = torch.randn(batch, 3, 224, 224, device=d, generator=torch.Generator(device="cuda").manual_seed(72))
    lib = _lib.load()
    real = _lib.check
    state = dict(pattern=0, n=0, busy=False)

    def check_then_poison(code, what):
        real(code, what)
        if state["busy"] or "option" in what:
            return
        state["busy"] = True
        try:
            pat = state["pattern"] if state["pattern"] == 0 else PATTERNS[state["n"] % len(PATTERNS)]
            state["n"] += 1
            real(lib.vtx_debug_lds_poison(pat, 1, ops._stream()), "vtx_debug_lds_poison")
        finally:
            state["busy"] = False

    monkeypatch.setattr(_lib, "check", check_then_poison)
    monkeypatch.setattr(ops, "check", check_then_poison)

    def run(pattern):
        state["pattern"], state["n"] = pattern, 0
        torch.manual_seed(73)                                    # the same DropPath masks in both runs
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=BF):
            out = model(x)
        out.float().square().mean().backward()
        torch.cuda.synchronize()
        return out.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, state["n"]

    out0, g0, n0 = run(0)
    prev = torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled()
    torch.use_deterministic_algorithms(True, warn_only=True)
    try:
        out1, g1, n1 = run(1)
    finally:
        torch.use_deterministic_algorithms(prev[0], warn_only=prev[1])
    assert n0 == n1 and n0 > 20
    assert torch.isfinite(out1).all() and torch.equal(out0, out1), f"{name}: logits depend on LDS / uninitialised-buffer contents"
    assert g0.keys() == g1.keys() and len(g0) > 50
    for k in g0:
        assert torch.isfinite(g1[k]).all(), f"{name}: gradient of {k} is not finite with poisoned LDS"
        assert torch.equal(g0[k], g1[k]), f"{name}: gradient of {k} depends on LDS / uninitialised-buffer contents"
