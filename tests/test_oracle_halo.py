"""Pin the CPU oracle's halo-attention restatement (oracle/ref_ops.py halo_pos / halo_attention, oracle/ref_models.py halo_forward)
against the reference's own outputs (tests/golden/g12_halo.npz, tools/gen_goldens.py halo): index tables bit-exact, the attention
module in fp64 (output, input gradient, every parameter gradient), a small HaloTransformer's logits.  The reference model's backward
raises (in-place residual adds, halo_transformer.py:150-151), so whole-model gradients have no reference to be pinned to."""
import numpy as np
import pytest
import torch

from golden_util import Golden
from oracle import ref_models as M
from oracle import ref_ops as R
from oracle.formula import check_summary, fill, fill_state_dict, name_seed

CASES = {"w7a3": (64, 2, 32, 7, 3, (14, 14)), "w4a1": (64, 2, 32, 4, 1, (8, 12)), "w8a3d64": (128, 2, 64, 8, 3, (16, 16))}


def halo_params(dim, nh, dh, ntab, dtype=torch.float64):
    shapes = {"weight.weight": (3 * nh * dh, dim), "linear.weight": (dim, nh * dh), "linear.bias": (dim,), "rel_pos.weight": (ntab, nh)}
    sd = fill_state_dict({k: torch.zeros(s) for k, s in shapes.items()})
    return {k: v.to(dtype) for k, v in sd.items()}


@pytest.mark.parametrize("tag", sorted(CASES))
def test_halo_tables_bit_exact_and_attention_module_fp64(tag):
    g = Golden("g12_halo")
    dim, nh, dh, w, a, hw = CASES[tag]
    pos, ntab = R.halo_pos(w, a)
    assert ntab == int(g.arr(f"halo_{tag}.ntab"))
    assert np.array_equal(pos.numpy(), g.arr(f"halo_{tag}.pos").astype(np.int64))
    P = {k: v.requires_grad_(True) for k, v in halo_params(dim, nh, dh, ntab).items()}
    x = fill((2, hw[0], hw[1], dim), 91, 1.0, dtype=torch.float64).requires_grad_(True)
    out = R.halo_attention(x, P, nh, dh, w, a)
    check_summary(out, g.rec(f"halo_{tag}.out"), 5e-7, f"halo {tag} out")
    cot = fill(out.shape, name_seed(f"halo_{tag}.cot"), 1.0).double()
    grads = torch.autograd.grad((out * cot).sum(), [x] + list(P.values()))
    check_summary(grads[0], g.rec(f"halo_{tag}.dx"), 5e-7, f"halo {tag} dx")
    for (n, _), gr in zip(P.items(), grads[1:]):
        check_summary(gr, g.rec(f"halo_{tag}.d.{n}"), 5e-7, f"halo {tag} {n}")


def halo_model_params(g, dtype=torch.float32):
    keys = [str(k) for k in g.arr("halo_tiny.state_keys")]
    shapes = [eval(str(s)) for s in g.arr("halo_tiny.state_shapes")]
    sd = {k: torch.zeros(s) for k, s in zip(keys, shapes) if not k.endswith(".pos")}
    return {k: v.to(dtype) for k, v in fill_state_dict(sd).items()}


def test_halo_transformer_logits():
    g = Golden("g12_halo")
    P = halo_model_params(g)
    assert sum(v.numel() for v in P.values()) == int(g.arr("halo_tiny.n_params"))
    x = fill((2, 3, 224, 224), 21, 1.0)
    out = M.halo_forward(P, x, M.HALO_TINY)
    check_summary(out, g.rec("halo_tiny.eval.logits"), 2e-5, "halo tiny fp32 logits")
    check_summary(out, g.rec("halo_tiny.train_fwd.logits"), 2e-5, "halo tiny fp32 logits (train mode, drop_path 0)")
    P64 = {k: v.double() for k, v in P.items()}
    out64 = M.halo_forward(P64, x.double(), M.HALO_TINY)
    check_summary(out64, g.rec("halo_tiny.eval64.logits"), 5e-7, "halo tiny fp64 logits")
