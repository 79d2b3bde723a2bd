"""Window index / mask tables of (shifted-)window attention -- integer arithmetic, bit-exact.

Product-side construction of the two buffers the reference registers in
MultiHeadedLocalAttention.__init__ / make_mask_pos (reference models/swin_transformer.py:42-53,
55-101), in closed form ("A8 closed-form specification"):
  * token a of window n carries original coordinates Y[n,a], X[n,a] (the feature map rolled by
    -floor(w/2) when shifted);
  * dy = Y[n,b] - Y[n,a], dx = X[n,b] - X[n,a] (key minus query);
  * shifted layers: keep = |dx| < w and |dy| < w, local_mask = not keep (True -> -inf), diffs *= keep;
  * pos = (dy[0] + w - 1) * (2w - 1) + (dx[0] + w - 1): the table of WINDOW 0 only, int64.
"""
import torch


def window_coords(input_size, window, shift):
    H, W = input_size
    if H % window or W % window:
        raise ValueError(f"input size {input_size} is not a multiple of the window size {window}")
    r = -(window // 2) if shift else 0
    ys = (torch.arange(H, dtype=torch.int64) - r) % H      # torch.roll by r: rolled[p] = orig[(p - r) mod n]
    xs = (torch.arange(W, dtype=torch.int64) - r) % W
    nh, nw = H // window, W // window
    yy = ys.view(nh, 1, window, 1).expand(nh, nw, window, window)
    xx = xs.view(1, nw, 1, window).expand(nh, nw, window, window)
    return yy.reshape(nh * nw, window * window), xx.reshape(nh * nw, window * window)


def make_pos_mask(input_size, window, shift):
    """-> (pos int64 (w*w, w*w), local_mask bool (nW, w*w, w*w) or None)."""
    Y, X = window_coords(input_size, window, shift)
    dy = Y[:, None, :] - Y[:, :, None]
    dx = X[:, None, :] - X[:, :, None]
    local_mask = None
    if shift:
        keep = (dx.abs() < window) & (dy.abs() < window)
        dy = dy * keep
        dx = dx * keep
        local_mask = ~keep
    pos = (dy[0] + window - 1) * (2 * window - 1) + (dx[0] + window - 1)
    return pos.contiguous(), (local_mask.contiguous() if local_mask is not None else None)


def pos_csr(pos, ntab):
    """CSR of the pos table: (a,b) pairs grouped by table index, for the dense rel_pos gradient."""
    flat = pos.reshape(-1).cpu()
    order = torch.argsort(flat, stable=True).to(torch.int32)
    counts = torch.bincount(flat, minlength=ntab)
    offsets = torch.zeros(ntab + 1, dtype=torch.int32)
    offsets[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return order, offsets


def pos_inverse(pos, ntab):
    """Inverse of a [L, L] pos table for the window-attention backward (include/vtx.h, vtx_wattn_bwd): int32
    cells [count, ntab] with cells[t, b] = the t-th entry q * 64 + key (row stride 64) whose pos is b, in ascending
    (q, key) order, padded with L * 64 (a zero cell); count = the largest bin.  Lane b of the kernel gathers the rel_pos
    gradient of bin b from its LDS copy of dS in that order."""
    L = pos.shape[-1]
    order, offsets = pos_csr(pos, ntab)
    o = order.to(torch.int64)
    flat = (o // L) * 64 + (o % L)
    sizes = (offsets[1:] - offsets[:-1]).to(torch.int64)
    count = max(int(sizes.max()), 1)
    cells = torch.full((count, ntab), L * 64, dtype=torch.int32)
    b = torch.repeat_interleave(torch.arange(ntab), sizes)
    t = torch.arange(flat.numel()) - offsets[:-1].to(torch.int64)[b]
    cells[t, b] = flat.to(torch.int32)
    return cells.contiguous(), count


def mask_regions(local_mask):
    """Region ids of a shifted-window mask: -> (region uint8 (nW, 64), ok).

    The reference's local_mask (swin_transformer.py:82-89) is an equivalence structure -- tokens of a window attend to
    each other iff they come from the same side of the cyclic-shift seams -- so it is carried to the window-attention
    kernels as ONE id per token: region[n, a] = first b with local_mask[n, a, b] == False, and
    local_mask[n, a, b] == (region[n, a] != region[n, b]).  `ok` is False when the given buffer does not have that
    structure (e.g. a hand-edited mask from a checkpoint); callers then use the generic masked kernels instead.
    """
    keep = ~local_mask.bool()
    nW, L, _ = keep.shape
    if L > 64:
        return None, False
    region = keep.to(torch.uint8).argmax(-1)                         # first attended key of every query
    ok = bool(torch.equal(region[:, :, None] == region[:, None, :], keep))
    out = torch.zeros((nW, 64), dtype=torch.uint8, device=local_mask.device)
    out[:, :L] = region.to(torch.uint8)
    return out.contiguous(), ok


def make_halo_pos(window, halo):
    """The relative-position index table of halo attention (reference models/halo_transformer.py:41-57): pos[q][k] for query q of the
    window x window block and key k of its (window + 2 halo)^2 neighbourhood, and the table size.  Offsets are measured in the
    neighbourhood's own grid: dy = ky - (qy + halo) + (window + halo - 1), likewise dx; pos = dy * (window + 2 halo) + dx.
    Returns (pos int64 [window^2, (window + 2 halo)^2], n_table)."""
    side = window + 2 * halo
    q = torch.arange(window)
    k = torch.arange(side)
    off = window + halo - 1
    dy = k.view(1, 1, side, 1) - (q.view(window, 1, 1, 1) + halo) + off          # [qy, qx, ky, kx]
    dx = k.view(1, 1, 1, side) - (q.view(1, window, 1, 1) + halo) + off
    pos = (dy * side + dx).reshape(window * window, side * side).to(torch.int64)
    n_table = off * 2 * side + off * 2 + 1
    return pos, n_table
