"""Shared case table of the attention-probability dropout fixtures (tests/golden/g11_attn_dropout.npz, written by
tools/gen_goldens.py attn_dropout: the reference's modules in train mode with dropout 0.25 and an injected, recorded keep mask)."""
import numpy as np
import torch

from oracle import ref_ops as R
from oracle.formula import fill, fill_state_dict

P_DROP = 0.25

# name -> (parameter shapes, input shape, input seed, oracle call(x, params, keep))
CASES = {
    "vit_L37": ({"qkv.weight": (384, 128), "qkv.bias": (384,), "linear.weight": (128, 128), "linear.bias": (128,)}, (2, 37, 128), 21,
                lambda x, p, k: R.global_attention(x, p["qkv.weight"], p["qkv.bias"], p["linear.weight"], p["linear.bias"], 2, keep=k,
                                                   drop_p=P_DROP)),
    "vit_L197": ({"qkv.weight": (384, 128), "qkv.bias": (384,), "linear.weight": (128, 128), "linear.bias": (128,)}, (2, 197, 128), 21,
                 lambda x, p, k: R.global_attention(x, p["qkv.weight"], p["qkv.bias"], p["linear.weight"], p["linear.bias"], 2, keep=k,
                                                    drop_p=P_DROP)),
    "swin_s1": ({"weight.weight": (192, 64), "weight.bias": (192,), "linear.weight": (64, 64), "linear.bias": (64,),
                 "rel_pos.weight": (169, 2)}, (2, 14, 14, 64), 22,
                lambda x, p, k: R.window_attention(x, p["weight.weight"], p["weight.bias"], p["linear.weight"], p["linear.bias"],
                                                   p["rel_pos.weight"], 2, 32, 7, True, keep=k, drop_p=P_DROP)),
    "swin_s0": ({"weight.weight": (192, 64), "weight.bias": (192,), "linear.weight": (64, 64), "linear.bias": (64,),
                 "rel_pos.weight": (169, 2)}, (2, 14, 14, 64), 22,
                lambda x, p, k: R.window_attention(x, p["weight.weight"], p["weight.bias"], p["linear.weight"], p["linear.bias"],
                                                   p["rel_pos.weight"], 2, 32, 7, False, keep=k, drop_p=P_DROP)),
    "pvt_r2": ({"linear_q.weight": (128, 128), "linear_kv.weight": (256, 128), "linear.weight": (128, 128), "linear.bias": (128,),
                "reduce_conv.weight": (128, 128, 2, 2), "reduce_conv.bias": (128,), "reduce_norm.weight": (128,),
                "reduce_norm.bias": (128,)}, (2, 64, 128), 23,
               lambda x, p, k: R.pvt_attention(x, 8, 8, p, 2, 2, keep=k, drop_p=P_DROP)),
    "pvt_r1_cls": ({"linear_q.weight": (128, 128), "linear_kv.weight": (256, 128), "linear.weight": (128, 128), "linear.bias": (128,)},
                   (2, 17, 128), 24, lambda x, p, k: R.pvt_attention(x, 4, 4, p, 2, 1, keep=k, drop_p=P_DROP)),
    "twins_local": ({"weight.weight": (192, 64), "weight.bias": (192,), "linear.weight": (64, 64), "linear.bias": (64,)},
                    (2, 14, 14, 64), 25, lambda x, p, k: R.twins_local_attention(x, p, 2, 32, 7, keep=k, drop_p=P_DROP)),
    "twins_global": ({"linear_q.weight": (64, 64), "linear_kv.weight": (128, 64), "linear.weight": (64, 64), "linear.bias": (64,),
                      "reduce_conv.weight": (64, 64, 7, 7), "reduce_conv.bias": (64,)}, (2, 14, 14, 64), 26,
                     lambda x, p, k: R.twins_global_attention(x, p, 2, 7, keep=k, drop_p=P_DROP)),
    "halo_w7a3": ({"weight.weight": (192, 64), "linear.weight": (64, 64), "linear.bias": (64,), "rel_pos.weight": (253, 2)}, (2, 14, 14, 64), 27,
                  lambda x, p, k: R.halo_attention(x, p, 2, 32, 7, 3, keep=k, drop_p=P_DROP)),
}


def kernel_order(name, keep):
    """The recorded mask as the kernels index it, [problems, Lq, Lk] with problem = (image, window, head): the reference's halo
    attention tensor is (B, heads, windows, Lq, Lk), everything else already has the kernels' order."""
    if name.startswith("halo"):
        keep = keep.permute(0, 2, 1, 3, 4)
    return keep.reshape(-1, keep.shape[-2], keep.shape[-1]).contiguous()


def keep_mask(g, name):
    """The recorded keep mask in the reference's attention-tensor layout: (B, heads, L, L) / (B, windows, heads, L, L) / (B, heads, Lq, Lk)."""
    shape = tuple(int(v) for v in g.arr(f"{name}.keepshape"))
    return torch.from_numpy(np.unpackbits(g.arr(f"{name}.keep"))[: int(np.prod(shape))].reshape(shape).copy())


def params(name, dtype=torch.float64):
    shapes = CASES[name][0]
    sd = fill_state_dict({k: torch.zeros(s) for k, s in shapes.items()})
    return {k: v.to(dtype) for k, v in sd.items()}


def case_input(name, dtype=torch.float64):
    _, shape, seed, _ = CASES[name]
    return fill(shape, seed, 1.0, dtype=torch.float64).to(dtype)
