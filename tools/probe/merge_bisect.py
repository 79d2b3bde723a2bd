"""Which tensor of a PatchMerge backward differs FIRST when a train step is not bit-reproducible?

The full-size Swin-S step with the weight gradients on the side stream and no fence (VTX_SIDE_FENCE=0, call-by-call layers:
VTX_LAYER_CALL=0) is not bit-reproducible in 10-20 % of trials (tools/probe/determinism_stress.py).  This probe records, for
every PatchMergeFn.backward of every trial, clones of the chain's tensors IN EXECUTION ORDER (dy in, dW, dln, dx, dgamma,
dbeta) and of every transformer layer's incoming gradient, compares them with trial 0 and prints the first one that differs,
with the rows / columns of the differing elements.
    VTX_LAYER_CALL=0 VTX_SIDE_FENCE=0 python tools/probe/merge_bisect.py [N]
"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch
import bench
from vtx import functional as VF, ops
from vtx.optim import FusedAdamW
from vtx.train_step import MixLoss, make_param_groups, train_step

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
MODE = os.environ.get("BISECT_MODE", "clone")      # clone: record tensors | light: only dx of the merges
dev = torch.device("cuda")
rec = []


def merge_backward(ctx, dy):
    x, ln_w, w, ln, mean, rstd = ctx.saved_tensors
    dy = VF._c(dy)
    dW, _ = ops.wgrad(dy, ln, want_bias=False)
    dln = VF.dgrad(dy, ctx.wp, x.dtype)
    VF.side_fence(dy.device, merge=True)
    dx, dg, db = ops.layernorm_bwd(dln, x, mean, rstd, ln_w.detach(), merge_hw=(x.shape[1], x.shape[2]))
    C = dln.shape[-1]
    if MODE == "clone":
        rec.append({"C": C, "dy": dy.clone(), "dW": dW.clone(), "dln": dln.clone(), "dx": dx.clone(), "dg": dg.clone(),
                    "db": db.clone(), "x": x, "mean": mean, "rstd": rstd, "g": ln_w.detach(),
                    "xc": x.clone(), "meanc": mean.clone(), "rstdc": rstd.clone()})
    else:
        rec.append({"C": C, "dx": dx.clone()})
    return dx, dg, db, dW, None


VF.PatchMergeFn.backward = staticmethod(merge_backward)

torch.manual_seed(0)
model = bench.build_model("swin_s", 0.3).to(dev).train()
init = {k: v.clone() for k, v in model.state_dict().items()}
g = torch.Generator().manual_seed(3)
B = 128
x = torch.randn(B, 3, 224, 224, generator=g).to(dev)
l1 = torch.randint(0, 1000, (B,), generator=g).to(dev)
data = (x, l1, l1.roll(1), torch.rand(B, generator=g).to(dev))
crit = MixLoss(0.1)
ref = None
bad = 0
first = {}
for t in range(N):
    model.load_state_dict(init)
    opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
    torch.manual_seed(7)
    rec.clear()
    for _ in range(2):
        train_step(model, crit, opt, data)
    torch.cuda.synchronize()
    cur = [dict(r) for r in rec]
    if ref is None:
        ref = cur
        print(f"trial 0: {len(ref)} PatchMerge backward calls recorded, C = {[r['C'] for r in ref]}", flush=True)
        continue
    found = None
    for i, (a, b) in enumerate(zip(ref, cur)):
        for k in ("dy", "dW", "dln", "dx", "dg", "db"):
            if k in a and not torch.equal(a[k], b[k]):
                found = (i, k)
                break
        if found:
            break
    if found:
        bad += 1
        i, k = found
        first[(i, k)] = first.get((i, k), 0) + 1
        if bad <= 4:
            a, b = ref[i][k], cur[i][k]
            C = ref[i]["C"]
            d = (a.float() - b.float()).abs()
            nz = d.flatten().nonzero().flatten()
            print(f"trial {t}: first differing tensor: call {i} (C = {C}) tensor {k}, {nz.numel()} elements, max |d| {float(d.max()):.3e} "
                  f"(|a| at those: {float(a.float().flatten()[nz].abs().mean()):.3e})", flush=True)
            if k == "dx":
                # dx is (B, H, W, Cs): which MERGED rows (b, i, j) do the differing elements belong to?
                Bq, H, W, Cs = a.shape
                idx = nz.cpu()
                c = idx % Cs; xx = (idx // Cs) % W; yy = (idx // (Cs * W)) % H; bb = idx // (Cs * W * H)
                mrow = (bb * (H // 2) + yy // 2) * (W // 2) + xx // 2
                rows = sorted(set(mrow.tolist()))
                print(f"    merged rows touched: {rows[:16]}{' ...' if len(rows) > 16 else ''} ({len(rows)} rows)", flush=True)
                if MODE == "clone":
                    same_in = all(torch.equal(ref[i][q], cur[i][q]) for q in ("dy", "dln"))
                    print(f"    inputs of that LayerNorm backward bit-identical to trial 0: {same_in}; "
                          f"dgamma / dbeta identical: {torch.equal(ref[i]['dg'], cur[i]['dg'])} / {torch.equal(ref[i]['db'], cur[i]['db'])}", flush=True)
                    r = cur[i]
                    again = ops.layernorm_bwd(r["dln"], r["x"], r["mean"], r["rstd"], r["g"], merge_hw=(H, W))[0]
                    torch.cuda.synchronize()
                    print(f"    same kernel re-run on the recorded inputs now: equals trial 0 {torch.equal(again, ref[i]['dx'])}, "
                          f"equals the bad output {torch.equal(again, b)}", flush=True)
                    print(f"    x / mean / rstd as cloned at backward time equal trial 0's: {torch.equal(ref[i]['xc'], r['xc'])} / "
                          f"{torch.equal(ref[i]['meanc'], r['meanc'])} / {torch.equal(ref[i]['rstdc'], r['rstdc'])}; the live saved tensors still "
                          f"equal their clones: {torch.equal(r['x'], r['xc'])} / {torch.equal(r['mean'], r['meanc'])} / {torch.equal(r['rstd'], r['rstdc'])}", flush=True)
                    # fp64 reference of the LayerNorm backward on the recorded inputs, all rows
                    C4 = 4 * Cs
                    xg = r["xc"].view(Bq, H // 2, 2, W // 2, 2, Cs).permute(0, 1, 3, 2, 4, 5).reshape(-1, C4).double()
                    dl = r["dln"].reshape(-1, C4).double()
                    mu, rs_ = r["meanc"].double()[:, None], r["rstdc"].double()[:, None]
                    xh = (xg - mu) * rs_
                    gv = dl * r["g"].double()[None]
                    want = rs_ * (gv - gv.mean(1, keepdim=True) - xh * (gv * xh).mean(1, keepdim=True))
                    def merged(t):
                        return t.view(Bq, H // 2, 2, W // 2, 2, Cs).permute(0, 1, 3, 2, 4, 5).reshape(-1, C4).double()
                    for nm, t in (("trial 0", a), ("this trial", b), ("re-run", again)):
                        e = (merged(t) - want).abs().sum(1) / want.abs().sum(1)
                        worst = e.topk(4)
                        print(f"    {nm:10s}: per-row relative L1 error vs fp64: median {float(e.median()):.2e}, worst rows "
                              f"{worst.indices.tolist()} = {[f'{v:.2e}' for v in worst.values.tolist()]}", flush=True)
                    rr = rows[0]
                    ratio = (merged(b)[rr] / merged(a)[rr])
                    ok = merged(a)[rr].abs() > 1e-9
                    print(f"    row {rr}: this trial / trial 0 element ratio: median {float(ratio[ok].median()):.5f}, min {float(ratio[ok].min()):.5f}, max {float(ratio[ok].max()):.5f}; "
                          f"rstd[row] {float(r['rstdc'][rr]):.6e}, neighbours rstd[row-4..row+4] {[f'{float(v):.4e}' for v in r['rstdc'][max(rr - 4, 0):rr + 5]]}", flush=True)
                    d2 = (merged(again) != merged(a)).any(1).nonzero().flatten()
                    print(f"    re-run vs trial 0: {d2.numel()} rows differ: {d2[:12].tolist()}", flush=True)
                    outs = [ops.layernorm_bwd(r["dln"], r["x"], r["mean"], r["rstd"], r["g"], merge_hw=(H, W))[0] for _ in range(20)]
                    torch.cuda.synchronize()
                    print(f"    20 more standalone re-runs: equal to the first re-run {sum(torch.equal(o, again) for o in outs)}, equal to trial 0 "
                          f"{sum(torch.equal(o, a) for o in outs)}", flush=True)
print(f"merge_bisect: {bad} of {N - 1} trials differ; first differing (call, tensor) histogram: {first}  (env: " +
      " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VTX_") or k == "BISECT_MODE") + ")")
