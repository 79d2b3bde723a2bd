"""Round 6, VERDICT r5 item 2 (LayerNorm fold, BUILT): the norm_ff backward inside the epilogue of the fused-MLP backward (vtx_mlp_bwd_ln,
csrc/mlp_fused.hip mlp_bwd_kernel<.., LNB>) against the two launches it replaces (vtx_mlp_bwd + vtx_layernorm_bwd with the deferred
column reduce), at the Swin-S stage-1 shape and the PVT-Small / Twins-SVT-S stage-1 ones.  HIP events on the launch stream, 20 repetitions
after 3 warm-ups, operands rotated over 3 buffer sets (> 256 MB Infinity Cache)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vision-transformers-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from vtx import _lib, ops          # noqa: E402
import test_gpu_mlp_fused as T          # noqa: E402


def timed(fn, sets, reps=20):
    for i in range(3):
        fn(sets[i % len(sets)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(sets[i % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    lib = _lib.load()
    p = lambda t: None if t is None else t.data_ptr()
    for (M, C, ff) in ((401408, 96, 384), (401408, 64, 512), (401408, 64, 256)):
        rps = 3136
        sets = []
        nb = lib.vtx_layernorm_bwd_blocks(M, C)
        wsb = lib.vtx_layernorm_bwd_workspace(M, C)
        for k in range(3):
            ln2_, x1, dy, w1, b1, w2, b2, s = T._operands(M, C, ff, 100 + k, 0.1, rps)
            d = x1.device
            gamma = (1.0 + 0.1 * torch.randn(C)).to(d)
            ln2, mean, rstd = ops.layernorm_fwd(x1, gamma, torch.zeros_like(gamma), 1e-6)
            e = lambda n: torch.empty(M, n, dtype=torch.bfloat16, device=d)
            sets.append(dict(ln2=ln2, x1=x1, dy=dy, w1=w1, b1=b1, w2=w2, s=s, mean=mean, rstd=rstd, gamma=gamma, h=e(ff), dz=e(ff), dln2=e(C), dx1=e(C),
                             ws=torch.empty(wsb, dtype=torch.uint8, device=d)))
        st = ops._stream()

        def f_bwd(o):
            _lib.check(lib.vtx_mlp_bwd(1, p(o["ln2"]), p(o["dy"]), p(o["w1"]), p(o["b1"]), p(o["w2"]), p(o["s"]), rps, p(o["h"]), p(o["dz"]), p(o["dln2"]), M, C, ff, st), "bwd")

        def f_ln(o):
            _lib.check(lib.vtx_layernorm_bwd(p(o["dln2"]), p(o["x1"]), p(o["mean"]), p(o["rstd"]), p(o["gamma"]), p(o["dy"]), p(o["dx1"]), None, None, p(o["ws"]), wsb,
                                             M, C, 1, 0, 0, 0, st), "ln_bwd")

        def f_two(o):
            f_bwd(o); f_ln(o)

        def f_fold(o):
            _lib.check(lib.vtx_mlp_bwd_ln(1, p(o["ln2"]), p(o["dy"]), p(o["w1"]), p(o["b1"]), p(o["w2"]), p(o["s"]), rps, p(o["h"]), p(o["dz"]), p(o["x1"]), p(o["mean"]),
                                          p(o["rstd"]), p(o["gamma"]), p(o["dx1"]), p(o["ws"]), nb, M, C, ff, st), "bwd_ln")

        unit = M * C * 2 / 1e6
        r = ff / C
        tb, tl, tt, tf = timed(f_bwd, sets), timed(f_ln, sets), timed(f_two, sets), timed(f_fold, sets)
        print(f"== M = {M}, C = {C}, ff = {ff}  (one unit = rows x C x 2 B = {unit:.1f} MB)")
        print(f"  fused-MLP backward alone          {tb:8.1f} us ({(3 + 2 * r) * unit / tb:.2f} TB/s of {3 + 2 * r:.0f} units)")
        print(f"  LayerNorm backward alone          {tl:8.1f} us ({4 * unit / tl:.2f} TB/s of 4 units)")
        print(f"  the two launches back to back     {tt:8.1f} us")
        print(f"  folded (vtx_mlp_bwd_ln)           {tf:8.1f} us ({(4 + 2 * r) * unit / tf:.2f} TB/s of {4 + 2 * r:.0f} units)   -> {tt - tf:+.1f} us, {100 * (tf / tt - 1):+.1f} %")
    # ---- the qkv input gradient + norm_attn backward (gemm_skinny.hip dgrad_ln_kernel, option LN_FOLD bit 1)
    for (M, C, K) in ((401408, 96, 288), (401408, 64, 192)):
        sets = []
        nb = lib.vtx_layernorm_bwd_blocks(M, C)
        wsb = lib.vtx_layernorm_bwd_workspace(M, C)
        for k in range(3):
            g = torch.Generator().manual_seed(200 + k)
            d = torch.device("cuda")
            bf = lambda *s, std=1.0: (torch.randn(*s, generator=g) * std).to(torch.bfloat16).to(d)
            dy, x, dres, w = bf(M, K, std=0.05), bf(M, C), bf(M, C, std=0.05), bf(K, C, std=C ** -0.5)
            gamma = (1.0 + 0.1 * torch.randn(C)).to(d)
            _, mean, rstd = ops.layernorm_fwd(x, gamma, torch.zeros_like(gamma), 1e-6)
            e = lambda n: torch.empty(M, n, dtype=torch.bfloat16, device=d)
            sets.append(dict(dy=dy, x=x, dres=dres, w=w, wt=w.t().contiguous(), mean=mean, rstd=rstd, gamma=gamma, dln=e(C), dx=e(C),
                             ws=torch.empty(wsb, dtype=torch.uint8, device=d)))
        st = ops._stream()

        def f_dgrad(o):
            lib.vtx_gemm(0, 1, p(o["dy"]), p(o["wt"]), p(o["dln"]), M, C, K, K, K, C, None, None, None, 1, None, None, 0, st)

        def f_ln(o):
            _lib.check(lib.vtx_layernorm_bwd(p(o["dln"]), p(o["x"]), p(o["mean"]), p(o["rstd"]), p(o["gamma"]), p(o["dres"]), p(o["dx"]), None, None, p(o["ws"]), wsb,
                                             M, C, 1, 0, 0, 0, st), "ln_bwd")

        def f_two(o):
            f_dgrad(o); f_ln(o)

        def f_fold(o):
            _lib.check(lib.vtx_dgrad_ln(1, p(o["dy"]), p(o["wt"]), p(o["x"]), p(o["mean"]), p(o["rstd"]), p(o["gamma"]), p(o["dres"]), p(o["dx"]), p(o["ws"]), nb,
                                        M, C, K, st), "dgrad_ln")

        unit = M * C * 2 / 1e6
        r = K / C
        tg, tl, tt, tf = timed(f_dgrad, sets), timed(f_ln, sets), timed(f_two, sets), timed(f_fold, sets)
        print(f"== dgrad + LayerNorm backward: M = {M}, C = {C}, K = {K}  (one unit = {unit:.1f} MB)")
        print(f"  input gradient alone (vtx_gemm)   {tg:8.1f} us ({(r + 1) * unit / tg:.2f} TB/s of {r + 1:.0f} units)")
        print(f"  LayerNorm backward alone          {tl:8.1f} us ({4 * unit / tl:.2f} TB/s of 4 units)")
        print(f"  the two launches back to back     {tt:8.1f} us")
        print(f"  folded (vtx_dgrad_ln)             {tf:8.1f} us ({(r + 3) * unit / tf:.2f} TB/s of {r + 3:.0f} units)   -> {tt - tf:+.1f} us, {100 * (tf / tt - 1):+.1f} %")
    # ---- forward folds (option LN_FOLD bits 2, 3)
    for (M, C, ff) in ((401408, 96, 384), (401408, 64, 512)):
        rps = 3136
        sets = []
        for k in range(3):
            _, x1, _, w1, b1, w2, b2, s = T._operands(M, C, ff, 300 + k, 0.1, rps)
            d = x1.device
            gamma = (1.0 + 0.1 * torch.randn(C)).to(d); beta = torch.zeros_like(gamma)
            e = lambda n: torch.empty(M, n, dtype=torch.bfloat16, device=d)
            sets.append(dict(x1=x1, w1=w1, b1=b1, w2=w2, b2=b2, s=s, gamma=gamma, beta=beta, ln=e(C), y=e(C), mean=torch.empty(M, device=d), rstd=torch.empty(M, device=d),
                             wq=(torch.randn(3 * C, C) * C ** -0.5).to(torch.bfloat16).to(d), bq=torch.zeros(3 * C, device=d), qkv=e(3 * C)))
        st = ops._stream()

        def f_lnf(o):
            _lib.check(lib.vtx_layernorm_fwd(p(o["x1"]), p(o["gamma"]), p(o["beta"]), p(o["ln"]), p(o["mean"]), p(o["rstd"]), M, C, 1e-6, 1, 0, 0, 0, st), "ln")

        def f_mlp(o):
            _lib.check(lib.vtx_mlp_fwd(1, p(o["ln"]), p(o["w1"]), p(o["b1"]), p(o["w2"]), p(o["b2"]), p(o["x1"]), p(o["s"]), rps, p(o["y"]), None, None, M, C, ff, st), "fwd")

        def f_two(o):
            f_lnf(o); f_mlp(o)

        def f_fold(o):
            _lib.check(lib.vtx_mlp_fwd_ln(1, p(o["x1"]), p(o["gamma"]), p(o["beta"]), 1e-6, p(o["ln"]), p(o["mean"]), p(o["rstd"]), p(o["w1"]), p(o["b1"]), p(o["w2"]), p(o["b2"]),
                                          p(o["s"]), rps, p(o["y"]), M, C, ff, st), "fwd_ln")

        def g_gemm(o):
            lib.vtx_gemm(0, 1, p(o["ln"]), p(o["wq"]), p(o["qkv"]), M, 3 * C, C, C, C, 3 * C, p(o["bq"]), None, None, 1, None, None, 0, st)

        def g_two(o):
            f_lnf(o); g_gemm(o)

        def g_fold(o):
            _lib.check(lib.vtx_ln_gemm(1, p(o["x1"]), p(o["gamma"]), p(o["beta"]), 1e-6, p(o["ln"]), p(o["mean"]), p(o["rstd"]), p(o["wq"]), p(o["bq"]), p(o["qkv"]), M, C, 3 * C, st), "ln_gemm")

        tl, tm, tt, tf = timed(f_lnf, sets), timed(f_mlp, sets), timed(f_two, sets), timed(f_fold, sets)
        print(f"== forward: M = {M}, C = {C}, ff = {ff}")
        print(f"  LayerNorm forward alone {tl:8.1f} us | fused-MLP forward alone {tm:8.1f} us | back to back {tt:8.1f} us | folded (vtx_mlp_fwd_ln) {tf:8.1f} us -> {tt - tf:+.1f} us, {100 * (tf / tt - 1):+.1f} %")
        tg, tt, tf = timed(g_gemm, sets), timed(g_two, sets), timed(g_fold, sets)
        print(f"  qkv GEMM alone          {tg:8.1f} us | LayerNorm + GEMM back to back {tt:8.1f} us | folded (vtx_ln_gemm) {tf:8.1f} us -> {tt - tf:+.1f} us, {100 * (tf / tt - 1):+.1f} %")


if __name__ == "__main__":
    main()
