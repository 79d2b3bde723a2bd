"""Closed-form tensor fills (no RNG, no library-version dependence).

Weights and inputs of every golden vector are *defined by formula* so that the
committed fixtures only need to hold the reference's OUTPUTS: the inputs are
regenerated bit-identically wherever numpy's float64 ``sin`` is IEEE-correct to
the last place or two (values are rounded to fp32 afterwards, so a 1-ulp fp64
difference never survives).
"""
import zlib

import numpy as np
import torch


def name_seed(name: str) -> int:
    """Stable small integer derived from a tensor name."""
    return zlib.crc32(name.encode()) % 9973


def fill(shape, seed, scale=1.0, offset=0.0, dtype=torch.float32):
    """``offset + scale * sin(phase(i))`` over the flat index ``i``.

    phase(i) = (i + 1) * (12.9898 + 0.001 * seed) + 0.61803 * seed
    -- an irrational-ish stride so neighbouring elements decorrelate and no
    row/column of any realistic shape is constant, symmetric or periodic.
    """
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.float64)
    phase = (i + 1.0) * (12.9898 + 0.001 * float(seed)) + 0.61803 * float(seed)
    v = offset + scale * np.sin(phase)
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def fill_state_dict(state_dict, rel_pos_scale=0.5, weight_scale=0.05):
    """Formula-defined values for every floating tensor of a model state_dict.

    Keyed by tensor NAME (crc32) so the same name gets the same values in the
    reference model, the oracle and the HIP modules.  Integer / bool buffers
    (``pos``, ``local_mask``) are left untouched.
      * LayerNorm weights  -> 1 + 0.1 * sin     * biases -> 0.02 * sin
      * rel_pos.weight     -> rel_pos_scale * sin (reference zero-inits it,
        which would hide bias-gather bugs)
      * cls_token/pos_embed-> 0.05 * sin        * everything else -> 0.05 * sin
    """
    out = {}
    for k, v in state_dict.items():
        if not torch.is_floating_point(v):
            out[k] = v.clone()
            continue
        s = name_seed(k)
        if "norm" in k and k.endswith("weight") and v.ndim == 1:
            t = fill(v.shape, s, 0.1, 1.0)
        elif k.endswith("final_linear.0.weight"):
            t = fill(v.shape, s, 0.1, 1.0)
        elif k.endswith("bias"):
            t = fill(v.shape, s, 0.02)
        elif "rel_pos" in k:
            t = fill(v.shape, s, rel_pos_scale)
        else:
            t = fill(v.shape, s, weight_scale)
        out[k] = t.to(v.dtype)
    return out


def summarize(t: torch.Tensor, max_full=8192, n_sample=4096):
    """Compact pin of a tensor: full copy if small, else strided sample + norms."""
    t = t.detach().to(torch.float64).reshape(-1)
    rec = {
        "numel": np.int64(t.numel()),
        "l2": np.float64(t.norm().item()),
        "sum": np.float64(t.sum().item()),
        "absmax": np.float64(t.abs().max().item()) if t.numel() else np.float64(0),
    }
    if t.numel() <= max_full:
        rec["full"] = t.to(torch.float32).numpy()
    else:
        stride = t.numel() // n_sample
        rec["stride"] = np.int64(stride)
        rec["sample"] = t[::stride][:n_sample].to(torch.float32).numpy()
    return rec


def check_summary(t: torch.Tensor, rec: dict, rtol: float, what: str = ""):
    """Assert ``t`` matches a ``summarize`` record within relative-L2 ``rtol``.

    Returns the achieved relative error (of the full copy or of the sample).
    """
    t = t.detach().to(torch.float64).reshape(-1).cpu()
    assert t.numel() == int(rec["numel"]), f"{what}: numel {t.numel()} != {int(rec['numel'])}"
    if "full" in rec:
        ref = torch.from_numpy(np.asarray(rec["full"])).to(torch.float64).reshape(-1)
        got = t
    else:
        stride = int(rec["stride"])
        ref = torch.from_numpy(np.asarray(rec["sample"])).to(torch.float64).reshape(-1)
        got = t[::stride][: ref.numel()]
    denom = max(ref.norm().item(), 1e-30)
    err = (got - ref).norm().item() / denom
    assert err <= rtol, f"{what}: rel-L2 err {err:.3e} > {rtol:.1e}"
    l2 = float(rec["l2"])
    l2err = abs(t.norm().item() - l2) / max(l2, 1e-30)
    assert l2err <= max(rtol, 1e-6) * 4, f"{what}: L2 norm off by {l2err:.3e}"
    return err
