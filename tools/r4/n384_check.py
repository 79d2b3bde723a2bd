"""128 x 384 tiles (one workgroup per row strip, option GLDS_N384) vs the default tiling on the N = 384, K >= 768 GEMMs."""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
from vtx import ops, options
ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=20); a = ap.parse_args()
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
NSET = 6
for M, T in ((25088, 196), (50432, 197), (19600, 196)):
    for name, K, kind in (("fc2 fwd", 1536, "resid_dp"), ("fc1 dgrad", 1536, "plain"), ("qkv dgrad", 1152, "plain")):
        sets = []
        for _ in range(NSET):
            d = dict(x=rn(M, K).bfloat16(), w=(rn(384, K) * 0.03).bfloat16(), bias=rn(384))
            if kind == "resid_dp":
                d["resid"] = rn(M, 384).bfloat16(); d["rowscale"] = (torch.rand(M // T, device=dev, generator=g) > 0.3).float() / 0.7
            sets.append(d)
        def run(d):
            if kind == "resid_dp": return ops.gemm(d["x"], d["w"], 0, bias=d["bias"], resid=d["resid"], rowscale=d["rowscale"], rows_per_scale=T)
            return ops.gemm(d["x"], d["w"], 0)
        def timeit():
            for d in sets: run(d)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.iters): run(sets[i % NSET])
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / a.iters * 1e3
        with options.override(GLDS_N384=0):
            ref = run(sets[0]); t0 = timeit()
        with options.override(GLDS_N384=1):
            got = run(sets[0]); t1 = timeit()
        print(f"M {M:6d} {name:10s} K {K:5d} {kind:9s}: default {t0:6.1f} us   128 x 384 tiles {t1:6.1f} us  ({t0 / t1:4.2f}x)  {2.0 * M * 384 * K / t1 / 1e6:6.1f} TFLOP/s  bitwise equal: {torch.equal(ref, got)}", flush=True)
