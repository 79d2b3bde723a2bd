"""-m gpu: a COMPACTED transformer layer at the benchmark's row counts against the fp64 oracle.

The full-size evidence so far was property-only (determinism, full batch vs chunks in eval mode, compacted vs uncompacted):
the train-mode compacted path at B = 128 / 256 was compared with itself, never with the oracle.  Here ONE layer of the
benchmark geometry -- Swin-S stage 3 (B = 128, 14 x 14 x 384, window 7, shifted, 12 heads; models/swin_transformer.py:163-197)
and ViT-S/16 (B = 256, 197 x 384, 6 heads; models/vit.py:48-66) -- runs forward + backward in bf16 through the one-call layer
path with host-drawn DropPath masks at rate 0.27 (every GEMM / LayerNorm / attention launch row-mapped over the kept samples),
and is compared with the fp64 oracle evaluated on the kept samples of each branch, rounding where the product stores bf16.
Dropped samples must pass through bit for bit.

Tolerances (relative L2, stated here): branch contribution y - x 1e-2, dx 1e-2, parameter gradients 2e-2 -- the bf16 noise
floor of a whole layer (several chained bf16 roundings; the reference's own bf16-vs-fp64 floor is 8-9.5e-3).
"""
import pytest
import torch
from torch import nn

from gpu_util import check, dev
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


class _OneLayer(nn.Module):
    """A single transformer layer under the scopes a top-level model's forward opens (bf16 weight copies incl. the transposed
    ones the mapped backward multiplies by; one host-side DropPath draw)."""

    def __init__(self, layer, *extra):
        super().__init__()
        self.layer = layer
        self.extra = extra                                    # (PVT layers take the token grid: height, width)
        self._vtx_dp_compaction = True

    def forward(self, x):
        from vtx import functional as VF
        from vtx.nn import drop_path_scope
        with VF.weight_scope(self, x), drop_path_scope(self, x.shape[0], x.device):
            return self.layer(x, *self.extra)


def _randomize(layer, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if p.ndim == 1 and "norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.ndim == 1:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif "rel_pos" in n:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.04 * torch.randn(p.shape, generator=g))


def _run_product(model, x_cpu, gy_cpu, seed, monkeypatch, compacted=True):
    from vtx import functional as VF
    d = dev()
    monkeypatch.setattr(VF, "_LAYER_CALL", True)
    used = []
    real = VF._layer_perms
    monkeypatch.setattr(VF, "_layer_perms", lambda *a: (used.append(real(*a)), used[-1])[1])
    # poison the allocator's free memory: what compaction leaves unwritten must never be read
    junk = torch.full((1 << 28,), float("nan"), device=d, dtype=torch.bfloat16)
    del junk
    x = x_cpu.to(d).requires_grad_(True)
    torch.manual_seed(seed)                                   # the host-side DropPath draw (vtx.nn.drop_path_scope)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = model(x)
    with VF.deferred_wgrad(True):
        (y.float() * gy_cpu.to(d).float()).sum().backward()
    torch.cuda.synchronize()
    if compacted:
        assert used and used[0] is not None, "the layer did not take the compacted path"
    else:
        assert not used or used[0] is None, "a layer with < 8 % of its branches dropped runs over all rows and scales (the benchmark's stage 1)"
    return y.detach(), x.grad.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}, (used[0] if used else None)


def _masks(seed, B, p):
    torch.manual_seed(seed)
    keep = 1.0 - torch.tensor([p, p], dtype=torch.float32).view(-1, 1)
    return torch.rand(2, B) < keep                            # the draw of drop_path_scope, same generator state


def _compare(name, model, x, gy, y, dx, grads, branch_fn, masks, p, both_dropped=True, tol_branch=1e-2):
    """fp64 oracle on the kept samples of each branch; `branch_fn(which, inp, params64)` evaluates a branch."""
    q = R.bf16_round
    B = x.shape[0]
    m1, m2 = masks[0], masks[1]
    P = {n: t.detach().double().cpu().requires_grad_(True) for n, t in model.named_parameters()}
    x64 = x.double().requires_grad_(True)
    scale = 1.0 / (1.0 - p)
    k1, k2 = m1.nonzero().flatten(), m2.nonzero().flatten()
    a = branch_fn(0, x64[k1], P)
    x1 = x64.clone()
    x1 = x1.index_add(0, k1, scale * a)
    x1 = q(x1)                                                # the product stores x1 in bf16
    f = branch_fn(1, x1[k2], P)
    yref = x1.index_add(0, k2, scale * f)
    (yref * gy.double()).sum().backward()
    yc, xc = y.float().cpu(), x.float()
    # samples dropped in both branches pass through bit for bit
    both = (~m1 & ~m2).nonzero().flatten()
    assert both.numel() > 0 or not both_dropped
    assert torch.equal(yc[both], xc[both]), "a sample dropped by both branches must come out unchanged"
    kept = (m1 | m2).nonzero().flatten()
    check(f"{name}: branch contribution y - x (kept samples)", (yc - xc)[kept], (yref.detach() - x.double())[kept], tol_branch)
    check(f"{name}: y", yc, yref.detach(), 4e-3)
    check(f"{name}: dx", dx.float().cpu(), x64.grad, 1e-2)
    worst = 0.0
    for n, g in grads.items():
        ref = P[n].grad
        assert ref is not None, n
        e = check(f"{name}: d {n}", g.float().cpu(), ref, 2e-2)
        worst = max(worst, e)
    return worst


def test_compacted_swin_stage3_layer_at_the_benchmark_batch_vs_fp64_oracle(monkeypatch):
    from models.swin_transformer import TransformerLayer
    B, H, C, nH, dh, ff, w, p = 128, 14, 384, 12, 32, 1536, 7, 0.27
    torch.manual_seed(5)
    layer = TransformerLayer(C, nH, dh, ff, (H, H), w, shift=True, drop_path=p)
    _randomize(layer, 6)
    model = _OneLayer(layer).to(dev()).train()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, H, H, C, generator=g).bfloat16()
    gy = torch.randn(B, H, H, C, generator=g).bfloat16()
    y, dx, grads, perms = _run_product(model, x, gy, 99, monkeypatch)
    masks = _masks(99, B, p)
    assert perms[0][1] == int(masks[0].sum()) and perms[1][1] == int(masks[1].sum()), "the oracle's masks are the product's"
    assert perms[0][1] < B and perms[1][1] < B

    def branch(which, inp, P):
        q = R.bf16_round
        if which == 0:
            h = q(R.layer_norm(inp, P["layer.norm_attn.weight"], P["layer.norm_attn.bias"], 1e-6))
            return R.window_attention(h, q(P["layer.attn.weight.weight"]), P["layer.attn.weight.bias"],
                                      q(P["layer.attn.linear.weight"]), P["layer.attn.linear.bias"],
                                      P["layer.attn.rel_pos.weight"], nH, dh, w, True, q)
        h = q(R.layer_norm(inp, P["layer.norm_ff.weight"], P["layer.norm_ff.bias"], 1e-6))
        return R.feed_forward(h, q(P["layer.ff.0.weight"]), P["layer.ff.0.bias"], q(P["layer.ff.3.weight"]),
                              P["layer.ff.3.bias"], q)

    _compare("swin stage-3 layer B=128 compacted", model, x, gy, y, dx, grads, branch, masks, p)


def test_compacted_vit_layer_at_the_benchmark_batch_vs_fp64_oracle(monkeypatch):
    from models.vit import TransformerLayer
    B, L, C, nH, ff, p = 256, 197, 384, 6, 1536, 0.27
    torch.manual_seed(15)
    layer = TransformerLayer(C, nH, ff, 0.0, 0.0, 0.0, p)
    _randomize(layer, 16)
    model = _OneLayer(layer).to(dev()).train()
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, L, C, generator=g).bfloat16()
    gy = torch.randn(B, L, C, generator=g).bfloat16()
    y, dx, grads, perms = _run_product(model, x, gy, 199, monkeypatch)
    masks = _masks(199, B, p)
    assert perms[0][1] == int(masks[0].sum()) and perms[1][1] == int(masks[1].sum())

    def branch(which, inp, P):
        q = R.bf16_round
        if which == 0:
            h = q(R.layer_norm(inp, P["layer.norm_attn.weight"], P["layer.norm_attn.bias"], 1e-6))
            return R.global_attention(h, q(P["layer.attn.qkv.weight"]), P["layer.attn.qkv.bias"],
                                      q(P["layer.attn.linear.weight"]), P["layer.attn.linear.bias"], nH, q)
        h = q(R.layer_norm(inp, P["layer.norm_ff.weight"], P["layer.norm_ff.bias"], 1e-6))
        return R.feed_forward(h, q(P["layer.ff.0.weight"]), P["layer.ff.0.bias"], q(P["layer.ff.3.weight"]),
                              P["layer.ff.3.bias"], q)

    _compare("vit-s/16 layer B=256 compacted", model, x, gy, y, dx, grads, branch, masks, p)


# ---------------------------------------------------------------------------------------------------------------- round 6
# The step-level kernels of round 5 at the benchmark's STAGE-1 row counts (401 408 rows): the fused MLP (csrc/mlp_fused.hip), the four-wave
# window-attention forward (wattn_fwd4_kernel, dispatched from 4 096 problems per head up), the streaming weight-resident GEMMs and the
# 64-row grouped weight gradient -- directly against the fp64 oracle, not through the fused-vs-unfused chain.  DropPath rate 0.0125 = the
# benchmark's second layer (0.3 x 1 / 24): fewer than 8 % of the branches are dropped, so the layer runs all rows and scales (no compaction).
def _seed_with_drops(B, p, lo=1):
    """a host seed whose two masks each drop >= lo samples (the draw is torch's CPU generator: reproducible here and on the GPU box)"""
    for seed in range(300, 400):
        m = _masks(seed, B, p)
        if int((~m[0]).sum()) >= lo and int((~m[1]).sum()) >= lo:
            return seed
    raise AssertionError("no seed found")


def test_swin_stage1_layer_at_the_benchmark_batch_vs_fp64_oracle(monkeypatch):
    from models.swin_transformer import TransformerLayer
    from vtx import _lib, ops
    B, H, C, nH, dh, ff, w, p = 128, 56, 96, 3, 32, 384, 7, 0.0125
    assert _lib.load().vtx_mlp_fused_ok(1, B * H * H, C, ff) == 1, "the benchmark's stage-1 MLP must take the fused kernels"
    assert ops.wattn_fwd_kernel_name(torch.bfloat16, True, B * (H // w) ** 2) == "wattn_fwd4_kernel<true>"
    torch.manual_seed(25)
    layer = TransformerLayer(C, nH, dh, ff, (H, H), w, shift=True, drop_path=p)
    _randomize(layer, 26)
    model = _OneLayer(layer).to(dev()).train()
    g = torch.Generator().manual_seed(27)
    x = torch.randn(B, H, H, C, generator=g).bfloat16()
    gy = torch.randn(B, H, H, C, generator=g).bfloat16()
    seed = _seed_with_drops(B, p)
    y, dx, grads, _ = _run_product(model, x, gy, seed, monkeypatch, compacted=False)
    masks = _masks(seed, B, p)

    def branch(which, inp, P):
        q = R.bf16_round
        if which == 0:
            h = q(R.layer_norm(inp, P["layer.norm_attn.weight"], P["layer.norm_attn.bias"], 1e-6))
            return R.window_attention(h, q(P["layer.attn.weight.weight"]), P["layer.attn.weight.bias"],
                                      q(P["layer.attn.linear.weight"]), P["layer.attn.linear.bias"],
                                      P["layer.attn.rel_pos.weight"], nH, dh, w, True, q)
        h = q(R.layer_norm(inp, P["layer.norm_ff.weight"], P["layer.norm_ff.bias"], 1e-6))
        return R.feed_forward(h, q(P["layer.ff.0.weight"]), P["layer.ff.0.bias"], q(P["layer.ff.3.weight"]),
                              P["layer.ff.3.bias"], q)

    _compare("swin stage-1 layer B=128 (fused MLP, four-wave attention forward)", model, x, gy, y, dx, grads, branch, masks, p,
             both_dropped=False)


def test_pvt_stage1_layer_at_the_benchmark_batch_vs_fp64_oracle(monkeypatch):
    from models.pvt import TransformerLayer
    from vtx import _lib
    B, H, C, nH, ff, r, p = 128, 56, 64, 1, 512, 8, 0.0125
    assert _lib.load().vtx_mlp_fused_ok(1, B * H * H, C, ff) == 1
    torch.manual_seed(35)
    layer = TransformerLayer(C, nH, ff, reduction=r, drop_path=p)
    _randomize(layer, 36)
    model = _OneLayer(layer, H, H).to(dev()).train()
    g = torch.Generator().manual_seed(37)
    x = torch.randn(B, H * H, C, generator=g).bfloat16()
    gy = torch.randn(B, H * H, C, generator=g).bfloat16()
    seed = _seed_with_drops(B, p)
    y, dx, grads, _ = _run_product(model, x, gy, seed, monkeypatch, compacted=False)
    masks = _masks(seed, B, p)

    def branch(which, inp, P):
        q = R.bf16_round
        if which == 0:
            h = q(R.layer_norm(inp, P["layer.norm_attn.weight"], P["layer.norm_attn.bias"], 1e-6))
            ap = {k[len("layer.attn."):]: (q(v) if v.ndim > 1 else v) for k, v in P.items() if k.startswith("layer.attn.")}
            return R.pvt_attention(h, H, H, ap, nH, r, q)
        h = q(R.layer_norm(inp, P["layer.norm_ff.weight"], P["layer.norm_ff.bias"], 1e-6))
        return R.feed_forward(h, q(P["layer.ff.0.weight"]), P["layer.ff.0.bias"], q(P["layer.ff.3.weight"]),
                              P["layer.ff.3.bias"], q)

    # (branch contribution: 1.3e-2 here -- the key / value side chains a 4 096-term reduction conv, a LayerNorm and the kv projection, each
    #  stored in bf16, in front of the attention; measured 1.00e-2 on the first run of this test)
    _compare("pvt stage-1 layer B=128 (fused MLP, 8 x 8 reduction conv)", model, x, gy, y, dx, grads, branch, masks, p, both_dropped=False,
             tol_branch=1.3e-2)


@pytest.mark.parametrize("name,T,C,ff,J", [("swin stage 3", 25088, 384, 1536, 6), ("pvt stage 3", 25088, 320, 1280, 5),
                                           ("swin stage 2", 100352, 192, 768, 3), ("swin stage 4 (128 x 256 fallback)", 6272, 768, 3072, 4)])
def test_wide_tile_weight_gradients_at_the_benchmark_shapes_vs_fp64(name, T, C, ff, J):
    """the grouped weight gradient of one layer (dWqkv, dWproj, dW1, dW2 and the bias sums) on the wide-tile kernel
    (wgrad_wide_kernel<., ., J>, 128 x 64 J tiles; C = 768: the 128 x 256 fallback) at the benchmark's token counts, against x^T dy in
    fp64 -- until now these shapes were compared with the 128 x 128 tiling only."""
    from vtx import ops
    d = dev()
    g = torch.Generator().manual_seed(T % 97 + C)
    bf = lambda *s, std=1.0: (torch.randn(*s, generator=g) * std).to(torch.bfloat16).to(d)
    pairs = [(bf(T, C), bf(T, 3 * C, std=0.05)), (bf(T, C), bf(T, C, std=0.05)), (bf(T, C), bf(T, ff, std=0.05)), (bf(T, ff), bf(T, C, std=0.05))]
    jobs = [(dy, x, True, None) for x, dy in pairs]
    assert ops.wgrad_group_ok(jobs)
    tiles, j = ops.wgrad_wide_tiles([(dy.shape[1], x.shape[1]) for x, dy in pairs], want_j=True)
    assert tiles > 0 and j == J, f"{name}: this group must run on 128 x {64 * J} tiles (got J = {j}, {tiles} tiles)"
    outs = ops.wgrad_group(jobs)
    for (x, dy), (dw, db) in zip(pairs, outs):
        ref = dy.double().cpu().T @ x.double().cpu()
        check(f"{name}: dW [{dy.shape[1]} x {x.shape[1]}] vs fp64", dw, ref, 2e-5)
        check(f"{name}: db [{dy.shape[1]}] vs fp64", db, dy.double().cpu().sum(0), 2e-5)
