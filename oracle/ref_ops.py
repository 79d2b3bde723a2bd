"""CPU restatement of the reference's ViT / Swin hot-path operators.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pure PyTorch CPU math written
from the reference's semantics, one function per SURVEY.md section 8(a) row,
each citing the reference lines it follows.  Backward passes come from torch
autograd over these forwards (fp32 or fp64), which is the "plain fp32
reference of the same op" for floating-point kernels.

``q`` (optional callable) is applied wherever the bf16 product path stores a
tensor in bf16 (GEMM outputs, LayerNorm outputs, attention probabilities and
outputs); passing ``q=bf16_round`` turns the fp32 oracle into a tight model of
the bf16 kernels' rounding points.  ``q=None`` is the exact reference math.
"""
import math

import torch
import torch.nn.functional as F

from . import tables


def bf16_round(t):
    return t.to(torch.bfloat16).to(t.dtype)


def _q(t, q):
    return t if q is None else q(t)


# --------------------------------------------------------------------------- A4
def layer_norm(x, weight, bias, eps):
    """Row-wise LayerNorm, biased variance (nn.LayerNorm; vit.py:13, swin:12,206,221,277)."""
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * weight + bias


def linear(x, weight, bias=None):
    """y = x W^T + b (nn.Linear)."""
    y = x @ weight.t()
    if bias is not None:
        y = y + bias
    return y


def silu(x):
    return x * torch.sigmoid(x)


# --------------------------------------------------------------------------- A3
def feed_forward(x, w1, b1, w2, b2, q=None):
    """PositionwiseFeedForward: Linear -> SiLU -> Dropout(0) -> Linear (layer.py:186-196)."""
    z = _q(linear(x, w1, b1), q)
    h = _q(silu(z), q)
    return linear(h, w2, b2)


# --------------------------------------------------------------------------- A12
def drop_path_apply(branch, keep_mask, p):
    """DropPath with an injected per-sample mask (layer.py:172-180): x / (1-p) * mask."""
    if keep_mask is None or p == 0:
        return branch
    shape = [branch.shape[0]] + [1] * (branch.ndim - 1)
    return branch / (1.0 - p) * keep_mask.reshape(shape).to(branch.dtype)


# --------------------------------------------------------------------------- A2
def attn_dropout(P, keep, drop_p):
    """F.dropout(attn, p, training=True) with an INJECTED keep mask (vit.py:39, swin_transformer.py:144, pvt.py:60,
    twins.py:88,147): attn * keep / (1 - p) on the softmax output.  keep None / p == 0: identity (eval mode)."""
    if keep is None or drop_p == 0:
        return P
    return P * keep.to(P.dtype) / (1.0 - drop_p)


def global_attention_core(qkv, n_head, q=None, keep=None, drop_p=0.0):
    """Attention core of vit.MultiHeadedAttention (vit.py:30-42) on the QKV projection output.

    qkv: (B, L, 3C) with channel order [q|k|v][head][d] (vit.py:30-34); returns O (B, L, C) in
    [head][d] channel order (vit.py:42); the 1/sqrt(d) scale is applied to the product (vit.py:37).
    keep (B, n_head, L, L): dropout keep mask of the attention probabilities (vit.py:39).
    """
    B, L, C3 = qkv.shape
    C = C3 // 3
    d = C // n_head
    out = qkv.new_zeros(B, L, C)
    for h in range(n_head):
        Q = qkv[..., 0 * C + h * d:0 * C + (h + 1) * d]
        K = qkv[..., 1 * C + h * d:1 * C + (h + 1) * d]
        V = qkv[..., 2 * C + h * d:2 * C + (h + 1) * d]
        S = torch.einsum("bid,bjd->bij", Q, K) / math.sqrt(d)
        P = _q(attn_dropout(torch.softmax(S, -1), None if keep is None else keep[:, h], drop_p), q)
        out[..., h * d:(h + 1) * d] = torch.einsum("bij,bjd->bid", P, V)
    return out


def global_attention(x, w_qkv, b_qkv, w_o, b_o, n_head, q=None, keep=None, drop_p=0.0):
    """vit.MultiHeadedAttention.forward (vit.py:27-45)."""
    qkv = _q(linear(x, w_qkv, b_qkv), q)
    out = _q(global_attention_core(qkv, n_head, q, keep, drop_p), q)
    return linear(out, w_o, b_o)


# --------------------------------------------------------------------------- A7
def patchify(x, s):
    """NHWC (B,H,W,C) -> (B,H/s,W/s,s*s*C), flatten order (py,px,c) (swin:15-22)."""
    B, H, W, C = x.shape
    out = x.new_empty(B, H // s, W // s, s * s * C)
    for py in range(s):
        for px in range(s):
            o = (py * s + px) * C
            out[..., o:o + C] = x[:, py::s, px::s, :]
    return out


def window_token_index(H, W, window, shift):
    """Flat token index (y*W+x, original frame) of every window token, (nW, w*w).

    SURVEY "A9 semantic specification": roll commutes with per-token layers, so
    window (i,j) token (ay,ax) of a shifted layer is the original token at
    ((i*w+ay-r) mod H, (j*w+ax-r) mod W), r = -floor(w/2) (swin:109-111,157-158).
    """
    Y, X = tables.window_coords((H, W), window, shift)
    return torch.from_numpy(Y * W + X)


# --------------------------------------------------------------------------- A9
def window_attention_core(qkv, rel_pos, n_head, dim_head, window, shift, q=None, keep=None, drop_p=0.0):
    """Attention core of swin.MultiHeadedLocalAttention (swin:109-154), roll-free form.

    qkv: (B,H,W,3*h*dh) = output of the ``weight`` Linear on the UN-rolled input (roll commutes with
    per-token layers); returns O (B,H,W,h*dh) at original token positions (inverse partition + roll
    back folded in).  rel_pos: ((2w-1)^2, n_head).  pos / local_mask are rebuilt from oracle.tables
    (bit-exact vs the reference buffers, goldens G1/G2).  keep (B, nW, n_head, ww, ww): dropout keep mask of the
    attention probabilities in the reference's (B, S, H, W^2, W^2) layout (swin:144).
    """
    B, H, W, _ = qkv.shape
    hd = n_head * dim_head
    pos_np, mask_np = tables.make_pos_mask((H, W), window, shift)
    pos = torch.from_numpy(pos_np)
    idx = window_token_index(H, W, window, shift)            # (nW, ww)
    nW, ww = idx.shape
    g = qkv.reshape(B, H * W, 3 * hd)[:, idx.reshape(-1)].reshape(B, nW, ww, 3 * hd)
    bias = rel_pos[pos.reshape(-1)].reshape(ww, ww, n_head)  # swin:135
    o = qkv.new_zeros(B, nW, ww, hd)
    for h in range(n_head):
        Q = g[..., 0 * hd + h * dim_head:0 * hd + (h + 1) * dim_head]
        K = g[..., 1 * hd + h * dim_head:1 * hd + (h + 1) * dim_head]
        V = g[..., 2 * hd + h * dim_head:2 * hd + (h + 1) * dim_head]
        S = torch.einsum("bnid,bnjd->bnij", Q, K) / math.sqrt(dim_head)   # swin:134
        S = S + bias[..., h]
        if shift:
            S = S.masked_fill(torch.from_numpy(mask_np)[None], float("-inf"))  # swin:138-141
        P = _q(attn_dropout(torch.softmax(S, -1), None if keep is None else keep[:, :, h], drop_p), q)
        o[..., h * dim_head:(h + 1) * dim_head] = torch.einsum("bnij,bnjd->bnid", P, V)
    out_tok = qkv.new_zeros(B, H * W, hd)
    out_tok[:, idx.reshape(-1)] = o.reshape(B, nW * ww, hd)  # inverse partition + roll back
    return out_tok.reshape(B, H, W, hd)


def window_attention(x, w_qkv, b_qkv, w_o, b_o, rel_pos, n_head, dim_head,
                     window, shift, q=None, keep=None, drop_p=0.0):
    """swin.MultiHeadedLocalAttention.forward (swin:103-160).  x: (B,H,W,C) NHWC."""
    qkv = _q(linear(x, w_qkv, b_qkv), q)
    o = _q(window_attention_core(qkv, rel_pos, n_head, dim_head, window, shift, q, keep, drop_p), q)
    return linear(o, w_o, b_o)


# --------------------------------------------------------------------------- A7/A10
def swin_patch_embedding(x_nchw, w, b, ln_w, ln_b, q=None):
    """permute -> patchify(4) -> Linear(48->C) -> LayerNorm(eps 1e-5) (swin:208-213,371)."""
    x = x_nchw.permute(0, 2, 3, 1)
    t = _q(linear(patchify(x, 4), w, b), q)
    return layer_norm(t, ln_w, ln_b, 1e-5)


def patch_merge(x, ln_w, ln_b, w, q=None):
    """patchify(2) -> LayerNorm(4C, eps 1e-5) -> Linear(4C->C', no bias) (swin:216-229)."""
    t = _q(layer_norm(patchify(x, 2), ln_w, ln_b, 1e-5), q)
    return linear(t, w, None)


def vit_patch_embedding(x_nchw, w, b, patch):
    """Conv2d(3,C,k=p,s=p) -> flatten(2).transpose(1,2) (vit.py:73,76) as an im2col GEMM.

    Row (b, i, j); column order (c, py, px) = the conv weight's own flattening.
    """
    B, Cin, H, W = x_nchw.shape
    gh, gw = H // patch, W // patch
    cols = (x_nchw.reshape(B, Cin, gh, patch, gw, patch)
            .permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, Cin * patch * patch))
    return linear(cols, w.reshape(w.shape[0], -1), b)


# --------------------------------------------------------------------------- A13
def mix_loss(logits, t1, t2, ratio, eps, reduction="mean"):
    """MixLoss (loss.py:53-86): KL between log_softmax and the mixed smoothed one-hots; 'mean' = sum / B (train.py's),
    'none' = per-sample sums, anything else = the plain sum (loss.py:73-84)."""
    B, K = logits.shape
    logp = torch.log_softmax(logits, -1)
    on, off = 1 - eps + eps / K, eps / K
    d1 = torch.full_like(logp, off)
    d1[torch.arange(B), t1] = on
    d2 = torch.full_like(logp, off)
    d2[torch.arange(B), t2] = on
    r = ratio.reshape(B, 1).to(logp.dtype)
    td = r * d1 + (1 - r) * d2
    kl = torch.where(td > 0, td * (td.log() - logp), torch.zeros_like(td))
    if reduction == "none":
        return kl.sum(1)
    return kl.sum() / B if reduction == "mean" else kl.sum()


def clip_grad_norm(grads, max_norm):
    """nn.utils.clip_grad_norm_ (train.py:296): total L2 norm, coef = max/(total+1e-6) clamped to 1."""
    total = torch.sqrt(sum((g.to(torch.float64) ** 2).sum() for g in grads)).to(grads[0].dtype)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def adamw_step(p, g, m, v, step, lr, beta1, beta2, eps, wd):
    """torch.optim.AdamW single-tensor update (decoupled decay first)."""
    p = p * (1 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


# ------------------------------------------------------------------------------------------ PVT (models/pvt.py)
def sr_attention_core(q, kv, n_head, qf=None, keep=None, drop_p=0.0):
    """Attention core of pvt.MultiHeadedAttention (pvt.py:38, 51-64): q (B, Lq, C) = linear_q output, kv (B, Lk, 2C) =
    linear_kv output whose halves are k | v (pvt.py:51), heads = contiguous channel blocks of C // n_head (pvt.py:35-37),
    score = q k^T / sqrt(d) (pvt.py:55), softmax over keys, out (B, Lq, C) in [head][d] order (pvt.py:66)."""
    B, Lq, C = q.shape
    d = C // n_head
    out = q.new_zeros(B, Lq, C)
    for h in range(n_head):
        Q = q[..., h * d:(h + 1) * d]
        K = kv[..., h * d:(h + 1) * d]
        V = kv[..., C + h * d:C + (h + 1) * d]
        S = torch.einsum("bid,bjd->bij", Q, K) / math.sqrt(d)
        P = _q(attn_dropout(torch.softmax(S, -1), None if keep is None else keep[:, h], drop_p), qf)   # keep (B, n_head, Lq, Lk), pvt.py:60
        out[..., h * d:(h + 1) * d] = torch.einsum("bij,bjd->bid", P, V)
    return out


def pvt_reduce(x, height, width, w_conv, b_conv, g_norm, b_norm, reduction, qf=None):
    """Spatial reduction of the key/value tokens (pvt.py:42-47): tokens -> (B, C, H, W) -> Conv2d(C, C, r, stride r) ->
    tokens -> LayerNorm(eps 1e-6)."""
    B, L, C = x.shape
    img = x.transpose(1, 2).reshape(B, C, height, width)
    red = torch.nn.functional.conv2d(img, w_conv, b_conv, stride=reduction)
    red = _q(red.reshape(B, C, -1).transpose(1, 2), qf)
    return layer_norm(red, g_norm, b_norm, 1e-6)


def pvt_attention(x, height, width, p, n_head, reduction, qf=None, keep=None, drop_p=0.0):
    """pvt.MultiHeadedAttention.forward (pvt.py:31-68) without the returned score; p = dict of the module's tensors
    (linear_q.weight, linear_kv.weight, linear.weight, linear.bias [, reduce_conv.*, reduce_norm.*])."""
    qq = _q(linear(x, p["linear_q.weight"], None), qf)
    kvin = x
    if reduction > 1:
        kvin = _q(pvt_reduce(x, height, width, p["reduce_conv.weight"], p["reduce_conv.bias"], p["reduce_norm.weight"],
                             p["reduce_norm.bias"], reduction, qf), qf)
    kv = _q(linear(kvin, p["linear_kv.weight"], None), qf)
    out = _q(sr_attention_core(qq, kv, n_head, qf, keep, drop_p), qf)
    return linear(out, p["linear.weight"], p["linear.bias"])


def pvt_patch_embedding(x_nchw, w_conv, b_conv, g_norm, b_norm, pos, cls_token, patch):
    """pvt.PatchEmbedding.forward (pvt.py:126-140): Conv2d(in, dim, p, stride p) -> tokens -> LayerNorm(1e-6) ->
    [cls token in front] -> + pos; returns (tokens, (height, width))."""
    out = torch.nn.functional.conv2d(x_nchw, w_conv, b_conv, stride=patch)
    height, width = out.shape[2:]
    out = layer_norm(out.flatten(2).transpose(1, 2), g_norm, b_norm, 1e-6)
    if cls_token is not None:
        out = torch.cat((cls_token.view(1, 1, -1).expand(out.shape[0], -1, -1), out), 1)
    return out + pos.unsqueeze(0), (height, width)


# ------------------------------------------------------------------------------------------ Twins-SVT (twins.py:25-220)
def twins_peg(x, w, qf=None):
    """PositionalEncodingGenerator.forward (twins.py:32-37): NHWC -> NCHW, depthwise Conv2d(dim, dim, 3, padding 1, no bias,
    groups = dim), + the input, back to NHWC."""
    img = x.permute(0, 3, 1, 2)
    out = torch.nn.functional.conv2d(img, w, None, padding=1, groups=w.shape[0]) + img
    return _q(out.permute(0, 2, 3, 1), qf)


def twins_local_attention(x, p, n_head, dim_head, window, qf=None, keep=None, drop_p=0.0):
    """twins.MultiHeadedLocalAttention.forward (twins.py:109-151): qkv Linear, window partition (row-major windows,
    row-major tokens inside), softmax(q k^T / sqrt(d)) v per window and head -- no position bias, no mask, no shift --
    inverse partition, output Linear.  The partition is the un-shifted one of swin (window_attention_core with a zero table)."""
    qkv = _q(linear(x, p["weight.weight"], p["weight.bias"]), qf)
    zero = qkv.new_zeros((2 * window - 1) ** 2, n_head)
    o = _q(window_attention_core(qkv, zero, n_head, dim_head, window, False, qf, keep, drop_p), qf)
    return linear(o, p["linear.weight"], p["linear.bias"])


def twins_global_attention(x, p, n_head, reduction, qf=None, keep=None, drop_p=0.0):
    """twins.MultiHeadedAttention.forward (twins.py:56-93) on NHWC x: q = linear_q(tokens); the key / value tokens are the
    Conv2d(dim, dim, r, stride r) sub-sampled "image" of twins.py:69-72 (see the comment below; NO LayerNorm after it,
    unlike pvt.py:47); k | v =
    linear_kv(.).chunk(2) (twins.py:77); heads = contiguous channel blocks of dim // n_head (twins.py:60-63)."""
    B, H, W, C = x.shape
    tok = x.reshape(B, H * W, C)
    qq = _q(linear(tok, p["linear_q.weight"], None), qf)
    if reduction <= 1:
        # twins.py:74-77 would chunk a 4-D (B, H, W, 2 dim) tensor along dim 2 (the WIDTH); TransformerLayer never builds
        # the module that way (reduction = window_size, twins.py:182)
        raise ValueError("twins.MultiHeadedAttention with reduction 1 is not a meaningful configuration of the reference")
    # twins.py:69-70, kept as written: the module's input is 4-D (B, H, W, dim), so ``input.transpose(1, 2)`` swaps H and W
    # (it does NOT bring the channels forward as in pvt.py:44, where the input is (B, L, dim)) and the following
    # ``reshape(B, dim, H, W)`` reinterprets the (B, W, H, dim) memory as an NCHW image: the "image" the reduction conv sees is
    # a fixed permutation of the feature map's elements, not the feature map
    img = x.transpose(1, 2).reshape(B, C, H, W)
    red = torch.nn.functional.conv2d(img, p["reduce_conv.weight"], p["reduce_conv.bias"], stride=reduction)
    kvin = _q(red.reshape(B, C, -1).transpose(1, 2), qf)
    kv = _q(linear(kvin, p["linear_kv.weight"], None), qf)
    out = _q(sr_attention_core(qq, kv, n_head, qf, keep, drop_p), qf)
    return linear(out, p["linear.weight"], p["linear.bias"]).reshape(B, H, W, C)


# ------------------------------------------------------------------------------------------ Halo (models/halo_transformer.py)
def halo_pos(window, halo):
    """pos[q][k] and the table size of halo attention's relative-position term (halo_transformer.py:41-57): offsets between key
    (ky, kx) of the (window + 2 halo)^2 neighbourhood and query (qy + halo, qx + halo) in that grid, shifted by window + halo - 1."""
    side, off = window + 2 * halo, window + halo - 1
    pos = torch.empty(window * window, side * side, dtype=torch.int64)
    for qy in range(window):
        for qx in range(window):
            for ky in range(side):
                for kx in range(side):
                    pos[qy * window + qx, ky * side + kx] = (ky - qy - halo + off) * side + (kx - qx - halo + off)
    return pos, off * 2 * side + off * 2 + 1


def halo_attention(x, p, n_head, dim_head, window, halo, qf=None, keep=None, drop_p=0.0):
    """halo_transformer.MultiHeadedHaloAttention.forward (halo_transformer.py:58-115) on NHWC x: bias-free qkv Linear; the queries of
    every window x window block against the keys / values of its (window + 2 halo)^2 neighbourhood -- positions outside the map
    are ZERO key / value vectors that take part in the softmax (the zero padding of F.unfold, lines 70-76) -- plus the
    relative-position term rel_pos[pos] (lines 95-98); output Linear per token.  p: weight.weight, linear.weight, linear.bias,
    rel_pos.weight.  keep (B, n_head, windows, window^2, (window + 2 halo)^2): dropout keep mask of the attention probabilities in
    the reference's layout (line 101)."""
    B, H, W, _ = x.shape
    hd, side = n_head * dim_head, window + 2 * halo
    qkv = _q(linear(x, p["weight.weight"], None), qf)
    q, kv = qkv[..., :hd], qkv[..., hd:]
    kvp = torch.nn.functional.pad(kv, (0, 0, halo, halo, halo, halo))                       # zeros around the map (W then H)
    pos, _ = halo_pos(window, halo)
    bias = p["rel_pos.weight"][pos.reshape(-1)].reshape(window * window, side * side, n_head)
    out = x.new_zeros(B, H, W, hd)
    for i in range(H // window):
        for j in range(W // window):
            qw = q[:, i * window:(i + 1) * window, j * window:(j + 1) * window].reshape(B, window * window, n_head, dim_head)
            nb = kvp[:, i * window:i * window + side, j * window:j * window + side].reshape(B, side * side, 2, n_head, dim_head)
            S = torch.einsum("bqhd,bkhd->bhqk", qw, nb[:, :, 0]) / math.sqrt(dim_head) + bias.permute(2, 0, 1)
            P = _q(attn_dropout(torch.softmax(S, -1), None if keep is None else keep[:, :, i * (W // window) + j], drop_p), qf)
            o = torch.einsum("bhqk,bkhd->bqhd", P, nb[:, :, 1]).reshape(B, window, window, hd)
            out[:, i * window:(i + 1) * window, j * window:(j + 1) * window] = o
    return linear(_q(out, qf), p["linear.weight"], p["linear.bias"])


def twins_patch_embedding(x, w, b, ln_w, ln_b, size, qf=None):
    """twins.PatchEmbedding.forward (twins.py:214-220): patchify(size) -> Linear -> LayerNorm(eps 1e-5) on NHWC x."""
    t = _q(linear(patchify(x, size), w, b), qf)
    return layer_norm(t, ln_w, ln_b, 1e-5)


# ------------------------------------------------------------------------------------------ DINO (loss.py:89-152, vit.py:206-262)
def dino_head(x, p, depth=3, qf=None):
    """vit.DINOHead.forward (vit.py:255-262) without BatchNorm: mlp (Linear, exact-erf GELU, ...) -> L2 normalise ->
    weight-normed Linear(bottleneck -> out, no bias) with w = g * v / ||v||_row (nn.utils.weight_norm, dim 0).
    p: mlp.{0,2,4}.weight/bias (depth 3), last.weight_g (out, 1), last.weight_v (out, bottleneck)."""
    out = x
    for i in range(depth):
        out = _q(linear(out, p[f"mlp.{2 * i}.weight"], p[f"mlp.{2 * i}.bias"]), qf)
        if i < depth - 1:
            out = _q(torch.nn.functional.gelu(out), qf)
    out = _q(torch.nn.functional.normalize(out, dim=-1, p=2), qf)
    v = p["last.weight_v"]
    w = p["last.weight_g"] * v / v.norm(dim=1, keepdim=True)
    return linear(out, w, None)


def dino_loss(student, teacher, center, n_crop, student_temp, teacher_temp):
    """DINOLoss.forward (loss.py:122-144): student (n_crop*B, K), teacher (2*B, K), center (1, K);
    loss = mean over the (teacher crop iq, student crop v != iq) pairs of mean_b sum_k -q_iq log_softmax(s_v / ts)."""
    s = (student / student_temp).chunk(n_crop)
    q = torch.softmax((teacher - center) / teacher_temp, -1).detach().chunk(2)
    total, n = 0, 0
    for iq, qq in enumerate(q):
        for v in range(n_crop):
            if v == iq:
                continue
            total = total + torch.sum(-qq * torch.log_softmax(s[v], -1), -1).mean()
            n += 1
    return total / n


def dino_center_update(center, teacher, momentum, world=1):
    """DINOLoss.update_center (loss.py:146-152) for one process group of `world` ranks holding identical batches."""
    batch_center = teacher.sum(0, keepdim=True) * world / (teacher.shape[0] * world)
    return center * momentum + batch_center * (1 - momentum)


# ------------------------------------------------------------------------------------------ F4: input pipeline
def rand_bbox(size, ratio, rng):
    """mix_dataset.py:10-24.  NOTE the reference passes img.shape[1:] = (H, W) for tensors and unpacks it as (w, h): the
    names are swapped for non-square images; the returned x range indexes the LAST tensor dim, the y range dim 1."""
    w, h = size
    r = math.sqrt(1 - ratio)
    cut_w, cut_h = int(w * r), int(h * r)
    cx, cy = rng.randrange(w), rng.randrange(h)
    x1 = min(max(cx - cut_w // 2, 0), w)
    y1 = min(max(cy - cut_h // 2, 0), h)
    x2 = min(max(cx + cut_w // 2, 0), w)
    y2 = min(max(cy + cut_h // 2, 0), h)
    return x1, y1, x2, y2


def erase_pixels(mode, chan, h, w, tgen=None):
    """transforms._get_pixels (transforms.py:309-318): the colour block a rectangle is overwritten with -- per-pixel normal
    draws ('pixel'), one per channel ('rand') or zeros ('const'); tgen = torch CPU generator (None: the global one)."""
    if mode == "pixel":
        return torch.empty((chan, h, w), dtype=torch.float32).normal_(generator=tgen)
    if mode == "rand":
        return torch.empty((chan, 1, 1), dtype=torch.float32).normal_(generator=tgen)
    return torch.zeros((chan, 1, 1), dtype=torch.float32)


def random_erasing_const(img, rng, p=0.5, min_area=0.02, max_area=1 / 3, min_aspect=0.3, max_aspect=None, min_count=1,
                         max_count=None, mode="const", tgen=None):
    """transforms.RandomErasing._erase on one (C, H, W) tensor -- transforms.py:381-409; mode 'const' (zeros), 'rand' or
    'pixel' (normal draws from the torch generator tgen, transforms.py:309-318)."""
    max_aspect = max_aspect or 1 / min_aspect
    la = (math.log(min_aspect), math.log(max_aspect))
    max_count = max_count or min_count
    chan, img_h, img_w = img.shape
    if rng.random() > p:
        return img
    area = img_h * img_w
    count = min_count if min_count == max_count else rng.randint(min_count, max_count)
    for _ in range(count):
        for _attempt in range(10):
            target_area = rng.uniform(min_area, max_area) * area / count
            aspect = math.exp(rng.uniform(*la))
            h = int(round(math.sqrt(target_area * aspect)))
            w = int(round(math.sqrt(target_area / aspect)))
            if w < img_w and h < img_h:
                top = rng.randint(0, img_h - h)
                left = rng.randint(0, img_w - w)
                img[:, top:top + h, left:left + w] = erase_pixels(mode, chan, h, w, tgen)
                break
    return img


def mix_dataset_item(images, labels, index, mixup, cutmix, transform, rng):
    """MixDataset.__getitem__ (mix_dataset.py:36-90) for a dataset of tensors; rng = a random.Random-like object."""
    img1, label1 = images[index].clone(), labels[index]
    apply_mixup, apply_cutmix, ratio = mixup > 0, cutmix > 0, 1
    if apply_mixup or apply_cutmix:
        index2 = index
        while index2 == index:
            index2 = rng.randrange(len(images))
        img2, label2 = images[index2], labels[index2]
    else:
        img2, label2 = img1, label1
    if apply_mixup and apply_cutmix:
        if index % 2 == 0:
            apply_cutmix = False
        else:
            apply_mixup = False
    if apply_mixup:
        ratio = rng.betavariate(mixup, mixup)
        img1 = img1.mul(ratio).add_(img2, alpha=1 - ratio)
    if apply_cutmix:
        ratio = rng.uniform(0, 1) if cutmix == 1 else rng.betavariate(cutmix, cutmix)
        x1, y1, x2, y2 = rand_bbox(img1.shape[1:], ratio, rng)
        img1[:, y1:y2, x1:x2] = img2[:, y1:y2, x1:x2]
        ratio = 1 - ((x2 - x1) * (y2 - y1) / (img1.shape[1] * img1.shape[2]))
    if transform is not None:
        img1 = transform(img1)
    return img1, label1, label2, ratio
