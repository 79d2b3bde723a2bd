"""what the data-parallel machinery costs on ONE GPU -- Swin-S (B = 128) and ViT-S/16 (B = 256) bf16 train steps with
GradAllReduce(force=True) over a one-rank RCCL group (every autograd hook, bucket sink, packing copy, all_reduce launch on the side stream
and finish() live; the collective itself has nothing to exchange) against the world-1 bypass, alternating arms on the same box,
`--steps` timed steps per arm and repetition.  Target of the review: <= +1 %.

    python tools/r6/ddp_overhead_one_gpu.py [--steps 100] [--reps 2] > profiles/round6_ddp_overhead_one_gpu.txt
"""
import argparse, os, socket, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch, torch.distributed as dist
import bench
from vtx.ddp import GradAllReduce
from vtx.optim import FusedAdamW
from vtx.train_step import MixLoss, make_param_groups, train_step

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--side-arms", action="store_true", help="also run both arms with the side-stream weight gradients switched off (is the overlap alive under the machinery?)")
ap.add_argument("--models", default="swin_s,vit_s16")
args = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print(f"one-rank RCCL group (RCCL {'.'.join(map(str, torch.cuda.nccl.version()))}), {torch.cuda.get_device_name(0)}; {args.steps} timed steps per arm, "
      f"{args.reps} repetitions, arms alternate; bf16 autocast, fused clip + AdamW, side-stream weight gradients on")
crit = MixLoss(0.1)
from vtx import functional as VF
for name, B, dp in [m for m in (("swin_s", 128, 0.3), ("vit_s16", 256, 0.1)) if m[0] in args.models.split(",")]:
    x = torch.randn(B, 3, 224, 224, device=dev); l1 = torch.randint(0, 1000, (B,), device=dev)
    data = (x, l1, l1.roll(1), torch.rand(B, device=dev))
    res = {False: [], True: []}
    if args.side_arms:
        for side in (True, False):
            for use in (False, True):
                VF._SIDE_ENABLED = side
                torch.manual_seed(0)
                model = bench.build_model(name, dp).to(dev).train()
                opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
                ddp = GradAllReduce(model, force=use)
                for _ in range(8):
                    train_step(model, crit, opt, data, clip_grad_norm=5.0, autocast_dtype=torch.bfloat16, ddp=ddp)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(args.steps):
                    train_step(model, crit, opt, data, clip_grad_norm=5.0, autocast_dtype=torch.bfloat16, ddp=ddp)
                torch.cuda.synchronize()
                print(f"{name:8s} side-stream weight gradients {'on ' if side else 'OFF'}, machinery {'ON ' if use else 'off'}: {1e3 * (time.perf_counter() - t0) / args.steps:.3f} ms/step")
                if use:
                    ddp.remove()
                del model, opt, ddp
                torch.cuda.empty_cache()
        VF._SIDE_ENABLED = True
    for rep in range(args.reps):
        for use in (False, True):
            torch.manual_seed(0)
            model = bench.build_model(name, dp).to(dev).train()
            opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
            ddp = GradAllReduce(model, force=use)
            assert ddp.active == use
            for _ in range(8):
                train_step(model, crit, opt, data, clip_grad_norm=5.0, autocast_dtype=torch.bfloat16, ddp=ddp)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(args.steps):
                train_step(model, crit, opt, data, clip_grad_norm=5.0, autocast_dtype=torch.bfloat16, ddp=ddp)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / args.steps
            res[use].append(ms)
            nb, mb = len(ddp.buckets), sum(b.flat_numel for b in ddp.buckets) * 4 / 2**20
            print(f"{name:8s} rep {rep}: data-parallel machinery {'ON  (' + str(nb) + ' buckets, ' + format(mb, '.0f') + ' MB all-reduced per step)' if use else 'off (world-1 bypass)'}: {ms:.3f} ms/step")
            if use:
                ddp.remove()
            del model, opt, ddp
            torch.cuda.empty_cache()
    off, on = min(res[False]), min(res[True])
    moff, mon = sum(res[False]) / len(res[False]), sum(res[True]) / len(res[True])
    print(f"{name:8s} best-of-{args.reps}: off {off:.3f} ms, ON {on:.3f} ms -> {100 * (on / off - 1):+.2f} %;  mean: off {moff:.3f}, ON {mon:.3f} -> {100 * (mon / moff - 1):+.2f} %\n")
print("side-stream concurrency probe (vtx.functional._concurrent_stream):", VF.side_stream_report)
dist.destroy_process_group()
