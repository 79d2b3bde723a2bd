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

    def __init__(self, layer):
        super().__init__()
        self.layer = layer
        self._vtx_dp_compaction = True

    def forward(self, x):
        from vtx import functional as VF
        from vtx.nn import drop_path_scope
        with VF.weight_scope(self, x), drop_path_scope(self, x.shape[0], x.device):
            return self.layer(x)


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


def _run_product(model, x_cpu, gy_cpu, seed, monkeypatch):
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
    assert used and used[0] is not None, "the layer did not take the compacted path"
    return y.detach(), x.grad.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}, used[0]


def _masks(seed, B, p):
    torch.manual_seed(seed)
    keep = 1.0 - torch.tensor([p, p], dtype=torch.float32).view(-1, 1)
    return torch.rand(2, B) < keep                            # the draw of drop_path_scope, same generator state


def _compare(name, model, x, gy, y, dx, grads, branch_fn, masks, p):
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
    assert both.numel() > 0
    assert torch.equal(yc[both], xc[both]), "a sample dropped by both branches must come out unchanged"
    kept = (m1 | m2).nonzero().flatten()
    check(f"{name}: branch contribution y - x (kept samples)", (yc - xc)[kept], (yref.detach() - x.double())[kept], 1e-2)
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
