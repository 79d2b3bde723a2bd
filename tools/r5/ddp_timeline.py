#!/usr/bin/env python3
"""When does each gradient bucket close during backward, and what would be left exposed at 2 / 4 / 8 ranks?  (VERDICT r4 #5a)

Runs the benchmark's train step on ONE GPU with ``GradAllReduce(force=True)`` (1-rank RCCL group: every hook, pack and
collective launch of the N > 1 path runs) and stamps, with HIP events on the stream each thing is enqueued on:
    backward start | every bucket's close (the moment its all-reduce is enqueued) | backward end | finish() returned
From the stamps and the bucket sizes it derives the exposed all-reduce time for W ranks under two link models taken from
/opt/skills/guides (xGMI: 7 links x ~153 GB/s per GPU, point to point):
    ring      one ring over one link per hop: t = 2 (W - 1) / W x bytes / 153 GB/s          (per-link bound; what a naive ring gives)
    direct    reduce-scatter + all-gather over all W - 1 links at once: t = 2 x (bytes / W) / 153 GB/s
A bucket's collective starts when the bucket has closed AND the previous one has finished; exposed = end of the last one - backward end.

    python tools/r5/ddp_timeline.py [--model swin_s|vit_s16] [--steps 6]   (writes a markdown table to stdout)
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))

import torch
import torch.distributed as dist

LINK_GBS = 153.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="swin_s")
    ap.add_argument("--steps", type=int, default=6)
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import bench
    from vtx import functional as VF
    from vtx.ddp import GradAllReduce
    from vtx.optim import FusedAdamW
    from vtx.train_step import MixLoss, make_param_groups, backward_ddp

    batch = bench.default_batch(a.model)
    drop_path = 0.3 if a.model == "swin_s" else 0.1
    torch.manual_seed(0)
    model = bench.build_model(a.model, drop_path).to(dev).train()
    ddp = GradAllReduce(model, force=True)
    crit = MixLoss(eps=0.1)
    opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(batch, 3, 224, 224, device=dev, generator=g)
    l1 = torch.randint(0, 1000, (batch,), device=dev, generator=g)
    l2, ratio = l1.roll(1), torch.rand(batch, device=dev, generator=g)

    stamps = {}
    real_launch = ddp._launch

    def launch(b):
        side = VF.side_stream_after_current(b.params[0].device) if b.params[0].is_cuda else None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(side if side is not None else torch.cuda.current_stream())      # the stream the pack + collective go to
        stamps.setdefault("close", {})[b.index] = ev
        real_launch(b)

    ddp._launch = launch
    rows = []
    for it in range(a.steps):
        stamps.clear()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x)
            loss = crit(out, l1, l2, ratio)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        backward_ddp(loss, ddp, True, "boundary", True)
        e1.record()
        ddp.finish()
        e2.record()
        opt.step(max_grad_norm=5.0)
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        if it >= 2:                       # (the first backward learns the arrival order with per-parameter hooks)
            rows.append((e0.elapsed_time(e1), e0.elapsed_time(e2), {k: e0.elapsed_time(v) for k, v in stamps["close"].items()}))
    n = len(rows)
    bwd = sum(r[0] for r in rows) / n
    fin = sum(r[1] for r in rows) / n
    closes = {k: sum(r[2][k] for r in rows) / n for k in rows[0][2]}
    print(f"## {a.model}, batch {batch}, bf16, one MI355X, GradAllReduce(force=True) on a 1-rank RCCL group, mean of {n} steps\n")
    print(f"backward (GPU time, main stream): **{bwd:.2f} ms**; finish() returned at {fin:.2f} ms\n")
    print("| bucket | parameters | MB (fp32) | first .. last parameter | closes at (ms after backward start) | % of backward |")
    print("|---|---|---|---|---|---|")
    for b in ddp.buckets:
        print(f"| {b.index} | {len(b.params)} | {b.flat_numel * 4 / 2**20:.1f} | `{b.names[0]}` .. `{b.names[-1]}` | "
              f"{closes[b.index]:.2f} | {100 * closes[b.index] / bwd:.0f} % |")
    total = sum(b.flat_numel * 4 for b in ddp.buckets)
    print(f"\npayload {total / 1e6:.1f} MB per step\n")
    print("| ranks | model | per-bucket all-reduce (ms) | last collective ends at (ms) | exposed after backward (ms) | % of the 1-GPU step |")
    print("|---|---|---|---|---|---|")
    step_ms = float(os.environ.get("STEP_MS", "0")) or None
    for W in (2, 4, 8):
        for name, f in (("ring", lambda by: 2 * (W - 1) / W * by / (LINK_GBS * 1e9)),
                        ("direct", lambda by: 2 * (by / W) / (LINK_GBS * 1e9))):
            t = 0.0
            per = []
            for b in ddp.buckets:
                d = f(b.flat_numel * 4) * 1e3
                per.append(d)
                t = max(t, closes[b.index]) + d
            exposed = max(0.0, t - bwd)
            pct = f"{100 * exposed / step_ms:.1f} %" if step_ms else "-"
            print(f"| {W} | {name} | {', '.join(f'{d:.2f}' for d in per)} | {t:.2f} | {exposed:.2f} | {pct} |")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
