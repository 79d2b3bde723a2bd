#!/usr/bin/env python3
"""Headline benchmark: images/sec of a full training step (fwd + bwd + grad all-reduce + clip + AdamW step)
of Swin-S 224x224, bf16 autocast, batch 128 per GPU, synthetic data (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; RCCL gradient all-reduce (vtx.ddp) overlapped with backward; weak scaling (per-GPU batch
fixed).  ``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself
(re-exec through torch.distributed.run on 127.0.0.1, like the reference's ``dist.launch(main, conf.n_gpu, ...)``,
train.py:389-396).  Rank 0 prints ONE JSON line.  `roofline` = the HIP kernel with the largest share of the step
(every launch class is bracketed with HIP events on its launch stream in --event-steps steps that run right BEHIND the
timed region -- same process, same data; they run single-stream so that durations are attributable; the K timed steps
carry no brackets) + the per-kernel table;
`cpu_baseline` = the CPU oracle (a port of the reference's path, parity-checked against the reference's own outputs)
timed on this box's host cores as BASELINE.md section 3 prescribes.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "vision-transformers-pytorch_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

SWIN_S = dict(image_size=(224, 224), n_class=1000, depths=(2, 2, 18, 2), dims=(96, 192, 384, 768), dim_head=32,
              n_heads=(3, 6, 12, 24), dim_ffs=(384, 768, 1536, 3072), window_size=7)
# 3 x forward GEMM FLOPs per image (pvt_small: 2*M*N*K over patch embeddings, q / kv / proj / sr-conv
# / MLP GEMMs and the Lq x Lk attention products = 7.631 GFLOP forward = the PVT paper's 3.8 GMACs)
PVT_SMALL = dict(image_size=224, n_class=1000, in_dim=3, depths=(3, 4, 6, 3), patch_embed_dims=(64, 128, 320, 512),
                 n_heads=(1, 2, 5, 8), dim_ffs=(512, 1024, 1280, 2048), reductions=(8, 4, 2, 1))
# Twins-SVT-S in the reference's formulation (models/twins.py: one layer = one locally-grouped + one sub-sampled block, so depths
# (1, 1, 5, 2) are the paper's (2, 2, 10, 4); the sub-sampling conv has kernel = stride = window_size 7, twins.py:182): the
# hyper-parameters are the paper's, like PVT's they are not in the reference repository.  5.689 GFLOP forward per image.
TWINS_SVT_S = dict(n_class=1000, depths=(1, 1, 5, 2), dims=(64, 128, 256, 512), dim_head=32, n_heads=(2, 4, 8, 16),
                   dim_ffs=(256, 512, 1024, 2048), window_size=7)
TRAIN_GFLOP_PER_IMG = {"swin_s": 52.45, "vit_s16": 27.59, "pvt_small": 22.89, "twins_svt_s": 17.07,
                       # DINO DeiT-S/16 per SOURCE image: student 2 x 224^2 + 8 x 96^2 crops fwd+bwd, teacher 2 x 224^2 fwd,
                       # heads (384-2048-2048-256-65536) on 10 + 2 feature rows: 3 x (2 x 9.197 + 8 x 1.618 + 10 x 0.0449)
                       # + (2 x 9.197 + 2 x 0.0449) = 113.8 GFLOP
                       "dino": 113.8}
PEAK_BF16_TFLOPS = 2500.0                                       # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_TBPS = 8.0                                              # HBM3E peak (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3                                         # dense fp32 MFMA peak (parity mode)


def pmc_traffic(model, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (tools/pmc_traffic.sh: FETCH_SIZE x2 [gfx950 correction] + WRITE_SIZE, separate passes) -- PMC collection cannot
    run inside the timed process, so the figure comes from profiles/ (newest round first) together with the commit
    the profile was taken at (`_meta.commit` in the file; the kernels may have changed since: compare with HEAD);
    None when no summary is committed or the kernel is not in it."""
    for rnd in (6, 5, 4, 3, 2, 1):
        path = os.path.join(REPO, "profiles", f"round{rnd}_pmc_traffic_{model}.json")
        try:
            tab = json.load(open(path))
        except OSError:
            continue
        key = kernel.split(" (+")[0].replace(" ", "")
        if key.endswith(",...>"):
            # a kernel family the timer names by its leading template arguments (gemm_astat_kernel<NKT, mapped, ...>: the epilogue
            # kinds are separate instantiations): dispatch-weighted mean over the instantiations that match the prefix
            pre = key[:-len("...>")]
            hits = [v for k, v in tab.items() if k != "_meta" and k.replace(" ", "").startswith(pre)]
            n = sum(v.get("dispatches", 1) for v in hits)
            if n:
                return (sum((v["fetch_x2_bytes"] + v["write_bytes"]) * v.get("dispatches", 1) for v in hits) / n,
                        os.path.relpath(path, REPO), tab.get("_meta", {}).get("commit"))
        for k, v in tab.items():
            if k != "_meta" and k.replace(" ", "") == key:
                return (v["fetch_x2_bytes"] + v["write_bytes"], os.path.relpath(path, REPO),
                        tab.get("_meta", {}).get("commit"))
    return None, None, None


def build_model(name, drop_path):
    from models import SwinTransformer, VisionTransformer
    from vtx.nn import Linear
    if name == "swin_s":
        return SwinTransformer(**SWIN_S, drop_path=drop_path)          # config/swin-transformer-s.conf:1-12
    if name == "dino":
        from models.vit import dino                               # config/dino_deit-s-16.conf:1-19
        return dino(image_size=224, window_size=16, depth=12, dim=384, n_head=6, dim_ff=1536, dropout=0.0, drop_attn=0.0,
                    drop_ff=0.0, drop_path=drop_path, dim_head_out=65536, use_bn=False, norm_last_layer=False,
                    depth_head=3, dim_head_ff=2048, dim_head_bottleneck=256)
    if name == "pvt_small":
        from models.pvt import PyramidVisionTransformer
        return PyramidVisionTransformer(**PVT_SMALL, drop_path=drop_path)   # BASELINE.json cfg-4 (PVT paper hyper-parameters)
    if name == "twins_svt_s":
        from models.twins import TwinsSVT
        return TwinsSVT(**TWINS_SVT_S, drop_path=drop_path)                 # not a BASELINE.json configuration: --model only
    return VisionTransformer(Linear(384, 1000), 224, 16, 12, 384, 6, 1536, 0.0, 0.0, 0.0, drop_path)


def cpu_baseline_dino(batch, steps, cores=None):
    cores = cores or torch.get_num_threads()
    return _cpu_baseline_dino(batch, steps, cores)


def _cpu_baseline_dino(batch, steps, cores):
    """DINO step on the CPU oracle: student on 2 x 224^2 + 8 x 96^2 crops, teacher on the 2 global crops, heads, loss,
    backward, AdamW, momentum update -- fp32, bounded sample."""
    from oracle import ref_models as M
    from oracle import ref_ops as R
    model = build_model("dino", 0.0)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items() if torch.is_floating_point(v)}
    Tt = {k: v.detach().clone() for k, v in P.items()}
    hp = lambda D: {k[len("head."):]: v for k, v in D.items() if k.startswith("head.")}
    fwd = lambda D, crops: M.vit_forward(D, crops, M.VIT_S16, head=lambda f: R.dino_head(f, hp(D)))
    opt = torch.optim.AdamW(list(P.values()), lr=5e-4, weight_decay=0.04)
    crops = [torch.randn(batch, 3, 224, 224) for _ in range(2)] + [torch.randn(batch, 3, 96, 96) for _ in range(8)]
    center = torch.zeros(1, 65536)

    def step():
        with torch.no_grad():
            tout = fwd(Tt, crops[:2])
        loss = R.dino_loss(fwd(P, crops), tout, center, 10, 0.1, 0.04)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            for k in Tt:
                Tt[k].mul_(0.996).add_(P[k].detach(), alpha=0.004)

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return dict(value=round(batch * steps / dt, 3), unit="images/sec", cores=cores, kind="port",
                sample=f"dino DeiT-S/16 step (10 crops/image) fp32 on CPU oracle, batch {batch}, {steps} timed steps "
                       f"after 1 warm-up ({dt:.1f} s)")


def host_cpu():
    """(physical cores, model name) of this box's host CPU."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        cores = os.cpu_count()
    return int(cores or 1), model


def cpu_baseline(name, batch, steps):
    """the reference path restated on the CPU (oracle/, kind 'port'), fp32, train mode,
    torch.set_num_threads(physical cores); torch.manual_seed(0); x = randn(batch = 32, 3, 224, 224); 1 warm-up + 3 timed
    ``model(x).sum().backward()`` iterations (forward + backward, no optimizer)."""
    from oracle import ref_models as M
    from oracle import ref_ops as R
    cores, cpu_model = host_cpu()
    torch.set_num_threads(cores)
    if name == "dino":
        return cpu_baseline_dino(min(batch, 4), max(1, steps - 1), cores)
    torch.manual_seed(0)
    model = build_model(name, 0.0)                                      # parameter container only (CPU, never run)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()
         if torch.is_floating_point(v)}
    if name == "swin_s":
        fwd = lambda x: M.swin_forward(P, x, M.SWIN_S)
    elif name == "pvt_small":
        fwd = lambda x: M.pvt_forward(P, x, M.PVT_SMALL)
    elif name == "twins_svt_s":
        fwd = lambda x: M.twins_forward(P, x, M.TWINS_SVT_S)
    else:
        fwd = lambda x: M.vit_forward(P, x, M.VIT_S16, head=lambda f: R.linear(f, P["head.weight"], P["head.bias"]))
    torch.manual_seed(0)
    x = torch.randn(batch, 3, 224, 224)

    def it():
        t = time.perf_counter()
        fwd(x).sum().backward()
        for p in P.values():
            p.grad = None
        return time.perf_counter() - t

    warm = it()
    if warm > 75.0:                      # bound the sample: one timed iteration when a single one already takes > 75 s
        steps = 1
    ts = [it() for _ in range(steps)]
    dt = sum(ts)
    return dict(value=round(batch * steps / dt, 3), unit="images/sec", cores=cores, cpu_model=cpu_model, kind="port",
                sample=f"{name} model(x).sum().backward() fp32 on the CPU oracle, x = randn({batch}, 3, 224, 224), "
                       f"{steps} timed iterations after 1 warm-up ({warm:.1f} s warm-up, {dt:.1f} s timed), "
                       f"torch.set_num_threads({cores})")


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks through torch.distributed.run on
    127.0.0.1 (one process per GPU) and pass this process's arguments through -- what the reference's
    dist.launch(main, conf.n_gpu, ...) (train.py:389-396) does with spawn."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


def selftest_launch(rank, world):
    """--selftest-launch: the launcher / rendezvous plumbing only (gloo, CPU): every rank contributes rank + 1 to an
    all-reduce; rank 0 prints the observed world size and the sum.  Used by tests/test_ddp_gloo.py."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1.0])
        dist.all_reduce(t)
        seen = dist.get_world_size()
        dist.barrier()
        dist.destroy_process_group()
    else:
        t, seen = torch.tensor([1.0]), 1
    if rank == 0:
        print(json.dumps({"selftest_launch": True, "world_size_observed": seen, "sum": t.item()}))


def preflight(args, dev, rank, world):
    """N > 1 pre-flight (VERDICT r4 #5b): everything the data-parallel branch depends on is exercised and checked BEFORE a
    model is built, so that the first real 8-GPU run is not also the first execution of that branch -- and fails loudly, with a
    message that names the cause, instead of hanging in a collective or printing a wrong number.  Returns a dict of what was seen
    (rank 0 prints it on stderr; ``--preflight-only`` prints it as the JSON line and stops)."""
    info = {"world": world, "backend": dist.get_backend(), "rank0_device": str(dev)}
    if dist.get_world_size() != world:
        raise SystemExit(f"pre-flight: process group has {dist.get_world_size()} ranks, WORLD_SIZE says {world}")
    nccl = dist.get_backend() == "nccl"
    if nccl:
        ndev = torch.cuda.device_count()
        info["devices_visible"] = ndev
        if ndev < world and not args.share_gpu:
            raise SystemExit(f"pre-flight: {world} ranks but only {ndev} visible GPU(s) on this node (one process per GPU; "
                             "--share-gpu is the test-only way to oversubscribe)")
        info["rccl_version"] = ".".join(map(str, torch.cuda.nccl.version()))
        info["ipc_mode_legacy"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
        if info["ipc_mode_legacy"] != "0":
            print("pre-flight WARNING: HSA_ENABLE_IPC_MODE_LEGACY is not 0 -- RCCL's intra-node IPC may fail with "
                  "'hipIpcGetMemHandle: invalid argument' on this driver", file=sys.stderr)
    cdev = dev if nccl else torch.device("cpu")
    # every rank is there and they agree on who they are: sum of ranks, max of ranks
    t = torch.tensor([float(rank), 1.0], device=cdev)
    dist.all_reduce(t)
    if t[0].item() != world * (world - 1) / 2 or t[1].item() != world:
        raise SystemExit(f"pre-flight: all_reduce(sum) of the ranks gave {t.tolist()} for world {world}")
    # the collective GradAllReduce uses: fp32 average of a bucket-sized buffer (ncclAvg on RCCL; sum + divide elsewhere)
    n = 1 << 20
    buf = torch.full((n,), float(rank + 1), device=cdev)
    if nccl:
        try:
            dist.all_reduce(buf, op=dist.ReduceOp.AVG)
        except Exception as exc:     # noqa: BLE001
            raise SystemExit(f"pre-flight: ReduceOp.AVG is not supported by this RCCL build ({type(exc).__name__}: {exc})")
        info["reduce_op_avg"] = True
    else:
        dist.all_reduce(buf)
        buf /= world
        info["reduce_op_avg"] = False
    want = (world + 1) / 2
    if not torch.allclose(buf, torch.full_like(buf, want)):
        raise SystemExit(f"pre-flight: averaged all_reduce returned {buf[:4].tolist()} instead of {want}")
    # side-stream collective behind an event of the compute stream (how buckets are reduced during backward)
    if nccl:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            w = dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=True)
        w.wait()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if not torch.allclose(buf, torch.full_like(buf, want)):
            raise SystemExit("pre-flight: side-stream all_reduce changed the averaged buffer")
    dist.barrier()
    return info


def describe_buckets(ddp):
    """Bucket layout of a GradAllReduce (printed on stderr by rank 0 of an N > 1 run)."""
    return [{"bucket": b.index, "params": len(b.params), "MB": round(b.flat_numel * 4 / 2**20, 2), "first": b.names[0],
             "last": b.names[-1]} for b in ddp.buckets]


def padded_tile_ratio(kernel, shape):
    """MFMA work issued / algorithmic work of an attention launch: scores are computed in whole 16 x 16 tiles.  `shape` is the timer
    record's "rows x n x k flags f" (vtx/ops.py _describe_timer_rec): window / global attention: k tokens per problem on both sides
    (Swin 49 -> 64; ViT 197 -> 208 query rows x 224 keys: the fast path pairs key tiles); sub-sampled attention: the `rows` queries
    are whole tiles per image up to the last one, k reduced keys (49 / 50 -> 64)."""
    try:
        rows, n, k = (int(t) for t in shape.split(" flags")[0].split(" x "))
    except ValueError:
        return 1.0
    if k <= 0:
        return 1.0
    up = lambda v, m: (v + m - 1) // m * m
    if kernel.startswith("srattn"):
        return up(k, 16) / k
    if kernel.startswith("sattn"):
        return up(k, 16) * up(k, 32) / (k * k)
    return up(k, 16) ** 2 / (k * k)


def measured_peaks(dev):
    """What THIS box does on the two rooflines ("record the exact peak figure used, ideally a measured ...
    micro-benchmark on the box"): dense bf16 MFMA rate of a register-only v_mfma_f32_32x32x16_bf16 loop (libvtx vtx_debug_mfma_peak, 16
    waves per CU) and the HBM rate of a 1-GiB device-to-device copy (read + write bytes).  ~50 ms of GPU time, outside every timed region.
    The `frac` figures of the line stay on the NOMINAL peaks (2.5 PFLOP/s, 8 TB/s); these are reported beside them."""
    import ctypes
    from vtx import _lib, ops
    lib = _lib.load()
    sink = torch.zeros(16, device=dev)
    fl = ctypes.c_double(0.0)
    best = 0.0
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.vtx_debug_mfma_peak(20000, 16, sink.data_ptr(), ctypes.byref(fl), ops._stream()), "vtx_debug_mfma_peak")
        e1.record()
        torch.cuda.synchronize()
        if it:
            best = max(best, fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    bw = 0.0
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        if it:
            bw = max(bw, 2.0 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    torch.cuda.empty_cache()
    return {"mfma_bf16_tflops": round(best, 1), "hbm_copy_gbps": round(bw, 1),
            "how": "register-only v_mfma_f32_32x32x16_bf16 loop, 16 waves per CU; torch device-to-device copy of 1 GiB fp32 (read + write)"}


def default_batch(name):
    """Per-GPU batch of BASELINE.json's configurations: 256 ViT-S/16 (cfg-2), 128 Swin-S / PVT-Small (cfg-3 / 4), 64 DINO (cfg-5)."""
    return 256 if name == "vit_s16" else (64 if name == "dino" else 128)


def run_workload(model_name, batch, steps, warmup, args, dev, rank, world):
    """`warmup` untimed + `steps` timed train steps of one workload (barrier + synchronize on both sides, max over ranks);
    -> dict(value, ms_per_step, workload, roofline)."""
    from vtx import ops
    from vtx.ddp import GradAllReduce
    from vtx.train_step import MixLoss, make_param_groups, train_step

    drop_path = 0.3 if model_name == "swin_s" else 0.1
    torch.manual_seed(0)                       # identical init on every rank (+ rank-0 broadcast in GradAllReduce)
    model = build_model(model_name, drop_path).to(dev).train()
    # --force-ddp (N = 1): a one-rank RCCL group with every hook / bucket / pack / collective launch / finish() of the N > 1 run live --
    # the code path the 8-GPU run takes, on the one GPU there is (profiles/round6_ddp_overhead_one_gpu.txt, tests/test_gpu_bench_line.py)
    ddp = GradAllReduce(model, force=bool(getattr(args, "force_ddp", False)))
    if world > 1 and rank == 0:
        print(f"[bench] gradient buckets ({model_name}): {json.dumps(describe_buckets(ddp))}", file=sys.stderr)
    ac = torch.bfloat16 if args.dtype == "bf16" else None
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    torch.manual_seed(4242 + rank)             # DropPath masks differ per rank from here on (the model is already built)

    def make_opt(groups, lr):
        if args.optimizer == "fused":
            from vtx.optim import FusedAdamW
            return FusedAdamW(groups, lr=lr)                      # clip + AdamW: two multi-tensor HIP kernels
        return torch.optim.AdamW(groups, lr=lr, fused=True)       # torch's fused AdamW + foreach clip_grad_norm_

    if model_name == "dino":
        # train_dino.py:188-288 with config/dino_deit-s-16.conf: 2 global 224^2 + 8 local 96^2 crops per image, momentum
        # teacher (no grad), DINOLoss over 65536 outputs, AdamW (wd_skip dino), clip 3.0; epoch >= freeze_last_layer
        from vtx.dino import DINOLoss, dino_train_step
        teacher = build_model("dino", 0.0).to(dev).train()
        teacher.load_state_dict(model.state_dict())
        for p in teacher.parameters():
            p.requires_grad = False
        criterion = DINOLoss(65536, 10, 0.04, 0.07, 30, 300).to(dev)
        opt = make_opt(make_param_groups(model.named_parameters(), 0.04, "dino"), 5e-4)
        crops = [torch.randn(batch, 3, 224, 224, device=dev, generator=g) for _ in range(2)] + \
                [torch.randn(batch, 3, 96, 96, device=dev, generator=g) for _ in range(8)]
        workload = (f"DINO DeiT-S/16 step, {batch} images/GPU x (2 x 224^2 + 8 x 96^2 crops), head 384-2048-2048-256-65536, "
                    "momentum teacher 0.996, clip 3.0, AdamW(lr 5e-4, wd 0.04)")

        def step():
            return dino_train_step(model, teacher, criterion, opt, crops, epoch=1, momentum=0.996, clip_grad_norm=3.0,
                                   freeze_last_layer=1, autocast_dtype=ac, ddp=ddp)
    else:
        criterion = MixLoss(eps=0.1)
        opt = make_opt(make_param_groups(model.named_parameters(), 0.05, "vit"), 1e-3)
        x = torch.randn(batch, 3, 224, 224, device=dev, generator=g)
        l1 = torch.randint(0, 1000, (batch,), device=dev, generator=g)
        l2 = l1.roll(1)
        ratio = torch.rand(batch, device=dev, generator=g)
        data = (x, l1, l2, ratio)
        workload = (f"{model_name} 224x224 train step, batch {batch}/GPU, drop_path {drop_path}, "
                    "MixLoss(eps 0.1), clip 5.0, AdamW(lr 1e-3, wd 0.05), grad_accum 1")

        def step():
            return train_step(model, criterion, opt, data, clip_grad_norm=5.0, autocast_dtype=ac, ddp=ddp)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    assert torch.isfinite(loss).item(), "non-finite loss"

    # The event-sampled step(s) run BEHIND the timed region (round 6): the same step of the same process on the same
    # data, every launch class bracketed with HIP events on its launch stream -- a sampled step records ~1 200 events and runs
    # single-stream (~20 % slower than a timed step), which used to sit inside the K timed steps and cost the headline ~1.2 %.  Every rank
    # runs them (the data-parallel collectives stay matched); rank 0 reads them.
    timer = None if args.no_kernel_events else ops.KernelTimer()
    nsampled = 0
    step_events = []
    for i in range(max(1, args.event_steps) if timer is not None else 0):
        ops.set_kernel_timer(timer)
        nsampled += 1
        step_events.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
        step_events[-1][0].record()
        step()
        step_events[-1][1].record()
        ops.set_kernel_timer(None)
    # ... and two single-stream steps WITHOUT brackets: the denominator of `kernels_coverage` -- what share of a single-stream step's GPU
    # time the table's kernels account for; the remainder is launch boundaries (~1.5-1.9 us x ~600 dependent launches) and torch glue,
    # not the ~1 200 event records of the sampled step (those are what `sampled_step_ms` additionally carries)
    single_ms = None
    if timer is not None:
        from vtx import functional as VF_
        prev_side, VF_._SIDE_ENABLED = VF_._SIDE_ENABLED, False
        try:
            step()
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            step()
            step()
            eb.record()
            torch.cuda.synchronize()
            single_ms = ea.elapsed_time(eb) / 2
        finally:
            VF_._SIDE_ENABLED = prev_side
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()

    value = batch * world * steps / dt
    roof = None
    if rank == 0:
        if timer is not None:
            allk = timer.summary()
            if args.shape_table and model_name == args.model:      # (the headline workload's table; the secondaries of the same process do not overwrite it)
                rows_ = sorted(timer.by_shape().items(), key=lambda kv: -kv[1]["ms"])
                with open(args.shape_table, "w") as f:
                    f.write(f"{model_name} B = {batch}: launches timed inside libvtx over {nsampled} sampled step(s), HIP events on the launch stream; "
                            "shape = rows x n x k of the timer record (GEMM: C[rows, n] over k; LayerNorm: rows x C; attention: tokens x heads x "
                            "tokens per problem; grouped weight gradient: tokens x C x ff); bytes / flops algorithmic\n\n")
                    f.write("| kernel | shape | launches / step | avg us | ms / step | GB/s | TFLOP/s |\n|---|---|---|---|---|---|---|\n")
                    for (kn, shp), v in rows_:
                        us = v["ms"] / v["launches"] * 1e3
                        f.write(f"| `{kn}` | {shp} | {v['launches'] / nsampled:.1f} | {us:.1f} | {v['ms'] / nsampled:.3f} | "
                                f"{v['bytes'] / v['ms'] / 1e6:.0f} | {v['flops'] / v['ms'] / 1e9:.0f} |\n")
            peak = PEAK_BF16_TFLOPS if ac else PEAK_F32_TFLOPS
            sampled_ms = sum(a.elapsed_time(b) for a, b in step_events) / nsampled

            def row(k, v):
                sec = v["ms"] * 1e-3
                r = dict(launches_per_step=round(v["launches"] / nsampled, 1), ms_per_step=round(v["ms"] / nsampled, 3),
                         avg_launch_us=round(1e3 * v["ms"] / v["launches"], 2),
                         hbm_gbps_algorithmic=round(v["bytes"] / sec / 1e9, 1),
                         frac_hbm=round(v["bytes"] / sec / 1e12 / PEAK_HBM_TBPS, 4))
                if v["flops"] > 0:
                    r.update(tflops=round(v["flops"] / sec / 1e12, 1), frac_mfma=round(v["flops"] / sec / 1e12 / peak, 4),
                             # what the kernel's own algorithmic intensity (FLOP per HBM byte, scores on chip) allows
                             hbm_bound_tflops=round(v["flops"] / v["bytes"] * PEAK_HBM_TBPS, 1))
                return r

            # "report both algorithmic and padded-tile FLOPs" for the attention cores -- what the MFMA pipe is
            # asked to do: every 16-token query / key tile whole (49 -> 64 tokens per window, 197 -> 208 x 224 per ViT image, 49 / 50
            # reduced keys -> 64); from the shapes of the in-library timer records of the sampled steps
            padded = {}
            for (kn, shp), v in timer.by_shape().items():
                if "attn" in kn and v["flops"] > 0:
                    padded[kn] = padded.get(kn, 0.0) + v["flops"] * padded_tile_ratio(kn, shp)

            def row_p(k, v):
                r = row(k, v)
                if k in padded:
                    sec = v["ms"] * 1e-3
                    r["padded_tile_tflops"] = round(padded[k] / sec / 1e12, 1)
                    r["padded_tile_frac_mfma"] = round(padded[k] / sec / 1e12 / peak, 4)
                return r

            table = {k: row_p(k, v) for k, v in sorted(allk.items(), key=lambda kv: -kv[1]["ms"])}
            name, d = max(allk.items(), key=lambda kv: kv[1]["ms"])
            traffic, tsrc, tcommit = pmc_traffic(model_name, name)
            # the roofline that bounds the kernel: MFMA when its algorithmic intensity (FLOP per algorithmic HBM byte)
            # exceeds the machine balance 2.5 PFLOP/s / 8 TB/s = 312 FLOP/B, else HBM.  With K = C <= 768 and fused
            # epilogue streams every GEMM-class kernel of these models sits BELOW the balance point (forward / dgrad
            # 150-290 FLOP/B, grouped weight gradients ~220): they are HBM-bound kernels, like the attention cores
            mfma = d["flops"] > 0 and d["flops"] / d["bytes"] >= peak / PEAK_HBM_TBPS
            if mfma:
                ach, unit, pk = d["flops"] / (d["ms"] * 1e-3) / 1e12, "TFLOP/s", peak
            else:
                ach, unit, pk = d["bytes"] / (d["ms"] * 1e-3) / 1e9, "GB/s", PEAK_HBM_TBPS * 1e3
            roof = dict(bound="mfma" if mfma else "hbm", kernel=name, achieved=round(ach, 2), peak=pk, unit=unit,
                        frac=round(ach / pk, 4), traffic=traffic, traffic_source=tsrc, traffic_profile_commit=tcommit,
                        intensity_flop_per_byte=round(d["flops"] / d["bytes"], 1) if d["bytes"] else None,
                        frac_mfma=round(d["flops"] / (d["ms"] * 1e-3) / 1e12 / peak, 4),
                        frac_hbm=round(d["bytes"] / (d["ms"] * 1e-3) / 1e12 / PEAK_HBM_TBPS, 4),
                        algorithmic_bytes_per_launch=round(d["bytes"] / d["launches"]),
                        algorithmic_flops_per_launch=round(d["flops"] / d["launches"]),
                        launches_per_step=round(d["launches"] / nsampled, 1), event_sampled_steps=nsampled,
                        avg_launch_us=round(1e3 * d["ms"] / d["launches"], 2),
                        end_to_end_frac=round(value / world * TRAIN_GFLOP_PER_IMG[model_name] / 1e3 / peak, 4),
                        # the same with the FLOPs of the rows actually LAUNCHED (the timer tags of the sampled steps):
                        # stochastic-depth compaction skips the dropped samples' rows, so the nominal figure above counts
                        # work that is never issued -- this one is the MFMA utilisation of the work done
                        executed_flop_frac=round(sum(v["flops"] for v in allk.values()) / nsampled
                                                 / (dt / steps) / 1e12 / peak, 4),
                        executed_gflop_per_step=round(sum(v["flops"] for v in allk.values()) / nsampled / 1e9, 1),
                        # every launch class of the step (HIP events on the launch stream, sampled steps are
                        # single-stream): coverage = sum of the table / GPU time of the sampled steps
                        sampled_step_ms=round(sampled_ms, 3), single_stream_step_ms=round(single_ms, 3),
                        kernels_coverage=round(sum(v["ms"] for v in allk.values()) / nsampled / single_ms, 4),
                        kernels_coverage_of_sampled_step=round(sum(v["ms"] for v in allk.values()) / nsampled / sampled_ms, 4),
                        # (kernels_coverage: against a single-stream step without brackets -- the remainder is launch boundaries and
                        #  torch glue; _of_sampled_step: against the bracketed step itself, whose ~1 200 event records are most of what it
                        #  is missing; against the TIMED step the table sums to >= 1 because the timed steps overlap the weight gradients
                        #  on a second stream)
                        kernels_sum_ms=round(sum(v["ms"] for v in allk.values()) / nsampled, 3),
                        kernels_sum_over_timed_step=round(sum(v["ms"] for v in allk.values()) / nsampled / (1e3 * dt / steps), 4),
                        kernels=table)
    return dict(value=value, ms_per_step=1e3 * dt / steps, workload=workload, roofline=roof,
                ddp=dict(active=bool(ddp.active), buckets=len(ddp.buckets), world=ddp.world,
                         reduced_MB=round(sum(b.flat_numel for b in ddp.buckets) * 4 / 2**20, 1)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="swin_s", choices=["swin_s", "vit_s16", "pvt_small", "dino", "twins_svt_s"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 128 swin_s / pvt_small, 256 vit_s16, 64 dino)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--shape-table", default="", help="write the per-(kernel, shape) table of the sampled steps' in-library launches to this file")
    ap.add_argument("--event-steps", type=int, default=1,
                    help="event-sampled steps run BEHIND the timed region (HIP-event brackets around every kernel launch of the step, "
                         "single-stream: the per-kernel table and the `roofline` object come from them; the K timed steps carry no "
                         "brackets)")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--selftest-launch", action="store_true", help="launcher plumbing only (gloo on CPU, no model)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short ViT-S/16 / PVT-Small / DINO / Twins-SVT-S runs that follow the headline at --gpus 1 (`secondary`)")
    ap.add_argument("--secondary-steps", type=int, default=20)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of the N > 1 run: nccl (= RCCL, the product path) | gloo (TEST ONLY: lets the "
                         "world > 1 branch of this file run where RCCL cannot, e.g. two ranks on one GPU with --share-gpu)")
    ap.add_argument("--preflight-only", action="store_true",
                    help="N > 1: run only the pre-flight of the data-parallel branch (ranks, devices, RCCL version, ReduceOp.AVG, "
                         "side-stream collective) and print what it saw")
    ap.add_argument("--force-ddp", action="store_true",
                    help="--gpus 1 only: run the data-parallel machinery of the N > 1 run (RCCL process group of ONE rank, gradient buckets, "
                         "autograd hooks, packing copies, one all_reduce launch per bucket on the side stream, finish()) instead of bypassing "
                         "it -- what that machinery costs with nothing to exchange; the line carries `ddp`")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST ONLY: every rank uses cuda:(LOCAL_RANK mod device count) -- two ranks on the one GPU of a test box")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                     # never returns
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
    if args.selftest_launch:
        return selftest_launch(rank, world)
    if args.share_gpu:
        local_rank %= max(torch.cuda.device_count(), 1)
    cpu_only = args.preflight_only and args.backend == "gloo" and not torch.cuda.is_available()
    if cpu_only:
        dev = torch.device("cpu")                              # (the pre-flight's own CPU test: gloo, no GPU)
    else:
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} visible GPU(s) "
                             "(one process per GPU; --share-gpu is the test-only way to oversubscribe)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        pf = preflight(args, dev, rank, world)
        if rank == 0:
            print(f"[bench] pre-flight ok: {json.dumps(pf)}", file=sys.stderr)
        if args.preflight_only:
            if rank == 0:
                print(json.dumps({"preflight": pf}))
            dist.destroy_process_group()
            return
    elif args.preflight_only:
        print(json.dumps({"preflight": {"world": 1, "note": "single process: nothing to check"}}))
        return
    elif args.force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    batch = args.batch or default_batch(args.model)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.model, args.cpu_batch, args.cpu_steps)
        if args.model == "swin_s":
            # SURVEY.md section 8(d) / BASELINE.json cfg-1: ViT-S/16 forward + backward on 32 x 3 x 224 x 224 on the host cores too (~10 s)
            cpu["also"] = [dict(cpu_baseline("vit_s16", args.cpu_batch, args.cpu_steps), model="vit_s16")]

    from vtx import functional as VF
    res = run_workload(args.model, batch, args.steps, args.warmup, args, dev, rank, world)
    secondary = None
    if world == 1 and not args.no_secondary and args.model == "swin_s" and args.dtype == "bf16":
        # the other BASELINE.json configurations (cfg-2 / 4 / 5) in the same process, after the headline's timed region:
        # ~10 steps each, one event-sampled step; headline keys and timing untouched
        secondary = []
        for name in ("vit_s16", "pvt_small", "dino", "twins_svt_s"):
            import gc
            gc.collect()
            torch.cuda.empty_cache()               # the previous workload's model / activations: start from an empty allocator
            try:
                r = run_workload(name, default_batch(name), args.secondary_steps, 5, args, dev, rank, world)
            except Exception as exc:               # a secondary workload never takes the headline line down with it
                secondary.append({"model": name, "error": f"{type(exc).__name__}: {exc}"[:300]})
                continue
            rf = r["roofline"] or {}
            secondary.append({"workload": r["workload"], "model": name, "value": round(r["value"], 2), "unit": "images/sec",
                              "ms_per_step": round(r["ms_per_step"], 3), "steps": args.secondary_steps, "warmup": 5,
                              "roofline": {k: rf.get(k) for k in ("kernel", "bound", "frac", "frac_hbm", "frac_mfma",
                                                                  "avg_launch_us", "launches_per_step", "end_to_end_frac",
                                                                  "executed_flop_frac")}})
    peaks = None
    if rank == 0 and not args.no_kernel_events:
        try:
            peaks = measured_peaks(dev)
            rf = res["roofline"]
            if rf and peaks["mfma_bf16_tflops"] > 0 and peaks["hbm_copy_gbps"] > 0 and (rf["bound"] == "hbm" or args.dtype == "bf16"):
                # the dominant kernel's fraction against what this box measurably does (beside `frac`, which stays on the nominal peak)
                rf["frac_of_measured_peak"] = round(rf["achieved"] / (peaks["mfma_bf16_tflops"] if rf["bound"] == "mfma" else peaks["hbm_copy_gbps"]), 4)
        except Exception as exc:           # noqa: BLE001  (a measurement helper never takes the headline line down)
            peaks = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    if rank == 0:
        line = {
            "metric": "images/sec training (fwd+bwd+step)", "value": round(res["value"], 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(res["ms_per_step"], 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": res["workload"],
                       "global_batch": batch * world, "parallelism": f"dp{world}"},
            "world_size_observed": dist.get_world_size() if world > 1 else 1,
            "backend": (args.backend if world > 1 else None),
            "rccl_version": ".".join(map(str, torch.cuda.nccl.version())) if world > 1 and args.backend == "nccl" else None,
            "side_stream_wgrad": bool(VF._SIDE_ENABLED), "side_stream_probe": VF.side_stream_report or None, "ddp": res["ddp"],
            "roofline": res["roofline"], "measured_peaks": peaks, "cpu_baseline": cpu, "secondary": secondary,
        }
        print(json.dumps(line))
    if world > 1 or (args.force_ddp and dist.is_initialized()):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
