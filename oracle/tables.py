"""Integer / boolean window tables of shifted-window attention (bit-exact).

Restates ``MultiHeadedLocalAttention.make_mask_pos`` and the ``pos`` /
``local_mask`` buffer construction of the reference
(/root/reference/models/swin_transformer.py:42-53 ctor, :55-101 make_mask_pos)
in closed form with numpy integer arithmetic only ("A8
closed-form specification").  TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import numpy as np


def window_coords(input_size, window, shift):
    """Original (un-rolled) coordinates carried by every window token.

    Reference: swin_transformer.py:59-77 -- ``meshgrid`` grids rolled by
    ``-floor(window/2)`` when shifted, then cut into (nW, w, w) windows
    (row-major windows, row-major tokens).
    Returns int64 arrays Y, X of shape (nW, w*w).
    """
    H, W = input_size
    assert H % window == 0 and W % window == 0
    r = -(window // 2) if shift else 0
    # rolled[p] = original[(p - r) mod n]   (torch.roll by r)
    ys = (np.arange(H, dtype=np.int64) - r) % H
    xs = (np.arange(W, dtype=np.int64) - r) % W
    nh, nw = H // window, W // window
    Y = np.empty((nh * nw, window * window), dtype=np.int64)
    X = np.empty_like(Y)
    for i in range(nh):
        for j in range(nw):
            yy = ys[i * window:(i + 1) * window]
            xx = xs[j * window:(j + 1) * window]
            Y[i * nw + j] = np.repeat(yy, window)
            X[i * nw + j] = np.tile(xx, window)
    return Y, X


def make_pos_mask(input_size, window, shift):
    """(pos, local_mask) exactly as the reference registers them.

    pos        : int64 (w*w, w*w)   -- table of WINDOW 0 only (swin:44-45)
    local_mask : bool  (nW, w*w, w*w), True = masked (-inf); None if not shift
                 (swin:49-53 stores ``~mask``)
    dy[n,a,b] = Y[n,b] - Y[n,a] (key minus query, swin:79-86); for shifted
    layers keep = |dx|<w & |dy|<w and the diffs are multiplied by keep (90-91).
    """
    Y, X = window_coords(input_size, window, shift)
    dy = Y[:, None, :] - Y[:, :, None]
    dx = X[:, None, :] - X[:, :, None]
    if shift:
        keep = (np.abs(dx) < window) & (np.abs(dy) < window)
        dy = dy * keep
        dx = dx * keep
        local_mask = ~keep
    else:
        local_mask = None
    pos = (dy[0] + window - 1) * (2 * window - 1) + (dx[0] + window - 1)
    return pos.astype(np.int64), local_mask
