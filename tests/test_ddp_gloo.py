"""N > 1 path on CPU: vtx.ddp.GradAllReduce over the gloo backend, world_size 2 (runs without a GPU).

Checks the reference's DDP contract (train.py:102-107): rank-0 parameters/buffers are broadcast at
construction, and after backward every rank holds the MEAN of the per-rank gradients; also with gradient
accumulation (all-reduce on every micro-batch, no no_sync -- SURVEY appendix item 7) and bucket boundaries.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed):
    torch.manual_seed(seed)
    m = nn.Sequential(nn.Linear(24, 64), nn.SiLU(), nn.LayerNorm(64), nn.Linear(64, 33), nn.SiLU(), nn.Linear(33, 5))
    m.register_buffer("flag", torch.tensor([seed % 2 == 0]))           # bool buffer (like local_mask)
    m.register_buffer("table", torch.arange(6, dtype=torch.int64) + seed)  # int64 buffer (like pos)
    return m


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vtx.ddp import GradAllReduce, assign_buckets
        model = _model(seed=rank)                     # different init per rank: the broadcast must fix it
        ddp = GradAllReduce(model, bucket_bytes=6000, first_bucket_bytes=1000)
        ref = _model(seed=0)
        for (n, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
            assert torch.equal(a, b), f"rank {rank}: {n} not broadcast from rank 0"
        assert len(ddp.buckets) >= 3
        assert sum(len(b.params) for b in ddp.buckets) == len(list(model.parameters()))
        assert ddp.buckets[0].names[0] == "5.bias"    # reverse registration order

        def local_grads(m, step):
            torch.manual_seed(100 + 10 * step + rank)
            x = torch.randn(7, 24)
            m(x).square().sum().backward()

        # expected: mean over ranks of the local gradients (computed redundantly on every rank)
        exp = None
        for r in range(world):
            m2 = _model(seed=0)
            torch.manual_seed(100 + r)
            m2(torch.randn(7, 24)).square().sum().backward()
            g = [p.grad.clone() for p in m2.parameters()]
            exp = g if exp is None else [a + b for a, b in zip(exp, g)]
        exp = [e / world for e in exp]

        local_grads(model, 0)
        ddp.finish()
        for p, e in zip(model.parameters(), exp):
            assert torch.allclose(p.grad, e, rtol=1e-6, atol=1e-7)

        # second micro-batch WITHOUT zero_grad: accumulate, all-reduce again (reference: no no_sync)
        exp2 = None
        for r in range(world):
            m2 = _model(seed=0)
            torch.manual_seed(110 + r)
            m2(torch.randn(7, 24)).square().sum().backward()
            g = [p.grad.clone() for p in m2.parameters()]
            exp2 = g if exp2 is None else [a + b for a, b in zip(exp2, g)]
        exp2 = [a + b / world for a, b in zip(exp, exp2)]
        local_grads(model, 1)
        ddp.finish()
        for p, e in zip(model.parameters(), exp2):
            assert torch.allclose(p.grad, e, rtol=1e-5, atol=1e-6)

        # (gradients existed at the entry of that backward: the object has gone back to per-parameter counting for good --
        # a bucket-level trigger cannot verify arrival then)
        assert ddp._counting_only and all(len(b.handles) == len(b.params) for b in ddp.buckets)
        # set_to_none + a fresh backward re-arms the buckets
        model.zero_grad(set_to_none=True)
        local_grads(model, 0)
        ddp.finish()
        for p, e in zip(model.parameters(), exp):
            assert torch.allclose(p.grad, e, rtol=1e-6, atol=1e-7)
        assert all(len(b.handles) == len(b.params) for b in ddp.buckets)
        ddp.remove()
        ddp = GradAllReduce(model, bucket_bytes=6000, first_bucket_bytes=1000)
        model.zero_grad(set_to_none=True)
        local_grads(model, 0)
        ddp.finish()
        for p, e in zip(model.parameters(), exp):
            assert torch.allclose(p.grad, e, rtol=1e-6, atol=1e-7)

        # bucket-level hooks (VERDICT r3 #7a): after the first counted backward ONE hook per bucket is left -- on the parameter
        # whose gradient arrived last
        multi = [b for b in ddp.buckets if len(b.params) > 1]
        assert multi and all(len(b.handles) == 1 and b.trigger is not None for b in multi)
        # ... and if the order ever changes (here: the trigger is forced onto the FIRST-arriving parameter, so that it fires
        # while the bucket's other gradients are still missing) the bucket is reduced late, by finish(), never wrongly;
        # it counts per parameter again and re-learns the order in that same backward
        b = multi[0]
        ddp._count_hooks(b)
        b.learned = b.params[0]
        ddp._adopt_trigger(b)
        assert b.trigger is b.params[0]
        model.zero_grad(set_to_none=True)
        local_grads(model, 0)
        assert b.work is None and b.trigger is None            # fired early, found gradients missing, stood down
        ddp.finish()
        for p, e in zip(model.parameters(), exp):
            assert torch.allclose(p.grad, e, rtol=1e-6, atol=1e-7)
        assert len(b.handles) == len(b.params)
        model.zero_grad(set_to_none=True)
        local_grads(model, 0)
        ddp.finish()
        for p, e in zip(model.parameters(), exp):
            assert torch.allclose(p.grad, e, rtol=1e-6, atol=1e-7)
        assert len(b.handles) == 1 and b.trigger is not b.params[0]

        # reset(): a step abandoned between backward() and finish() leaves the object usable
        model.zero_grad(set_to_none=True)
        local_grads(model, 1)                                   # reduced buckets, handed-out state ... and no finish()
        ddp.reset()
        model.zero_grad(set_to_none=True)
        local_grads(model, 0)
        ddp.finish()
        for p, e in zip(model.parameters(), exp):
            assert torch.allclose(p.grad, e, rtol=1e-6, atol=1e-7)
        # per-parameter counting in every backward stays available
        ddp.remove()
        ddp2 = GradAllReduce(model, bucket_bytes=6000, first_bucket_bytes=1000, bucket_hooks=False)
        for _ in range(2):
            model.zero_grad(set_to_none=True)
            local_grads(model, 0)
            ddp2.finish()
        assert all(len(bb.handles) == len(bb.params) for bb in ddp2.buckets)
        for p, e in zip(model.parameters(), exp):
            assert torch.allclose(p.grad, e, rtol=1e-6, atol=1e-7)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_grad_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_world1_is_a_bypass():
    from vtx.ddp import GradAllReduce
    m = _model(0)
    ddp = GradAllReduce(m)
    assert ddp.world == 1 and ddp.buckets == []
    m(torch.randn(3, 24)).sum().backward()
    ddp.finish()
    assert all(p.grad is not None for p in m.parameters())


def test_bucket_assignment_sizes():
    from vtx.ddp import assign_buckets
    m = _model(0)
    b = assign_buckets(list(m.named_parameters()), bucket_bytes=10_000, first_bucket_bytes=100)
    assert [x for bb in b for x in bb.names] == [n for n, _ in reversed(list(m.named_parameters()))]
    assert b[0].numel * 4 <= 100 or len(b[0].params) == 1
    for bb in b[1:]:
        assert bb.numel * 4 <= 10_000 or len(bb.params) == 1


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` without a launcher must start 2 ranks itself (re-exec through torch.distributed.run
    on 127.0.0.1, like the reference's dist.launch(main, conf.n_gpu, ...), train.py:389-396) -- the command the
    round-end driver uses.  --selftest-launch runs the rendezvous plumbing only (gloo on CPU, no model)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--selftest-launch"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["world_size_observed"] == 2 and out["sum"] == 3.0
    # launcher and flag disagreeing is an error, not a silent 1-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--selftest-launch"],
                        capture_output=True, text=True, timeout=120, env=env2)
    assert r2.returncode != 0 and "disagree" in (r2.stderr + r2.stdout)


def test_bench_preflight_checks_the_data_parallel_branch_before_any_model():
    """VERDICT r4 #5b: `bench.py --gpus N` runs a pre-flight of the N > 1 branch (ranks agree, averaged all-reduce of a
    bucket-sized buffer is right) and fails LOUDLY on a node that cannot run it.  Here: 2 gloo ranks on CPU pass; an nccl run
    on a machine with fewer GPUs than ranks stops with a message that says so instead of hanging in a collective."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--backend", "gloo", "--preflight-only"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["preflight"]
    assert out["world"] == 2 and out["backend"] == "gloo" and out["reduce_op_avg"] is False
    assert "pre-flight ok" in r.stderr
    if torch.cuda.device_count() < 2:
        r2 = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--preflight-only"],
                            capture_output=True, text=True, timeout=300, env=env)
        assert r2.returncode != 0 and "visible GPU" in (r2.stderr + r2.stdout)


def _sink_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vtx import functional as VF
        from vtx.ddp import GradAllReduce

        class SinkLinear(torch.autograd.Function):
            """Stand-in for the HIP layer functions: its backward writes dW into functional.grad_sink(weight) when there
            is one and returns that tensor (what ops.wgrad_group(outs=...) does on the GPU)."""

            @staticmethod
            def forward(ctx, x, w):
                ctx.save_for_backward(x, w)
                return x @ w.t()

            @staticmethod
            def backward(ctx, dy):
                x, w = ctx.saved_tensors
                out = VF.grad_sink(w)
                dW = dy.t() @ x
                if out is not None:
                    out.copy_(dW)
                    dW = out
                return dy @ w, dW

        torch.manual_seed(0)
        lin = nn.Linear(16, 8, bias=True)
        model = nn.Sequential(lin, nn.SiLU(), nn.Linear(8, 4))
        ddp = GradAllReduce(model, bucket_bytes=1 << 20, first_bucket_bytes=1 << 20)
        for step in range(2):                         # second step: the grads were reset to None -> the sink is used again
            torch.manual_seed(10 + rank + 7 * step)
            x = torch.randn(5, 16)
            h = SinkLinear.apply(x, lin.weight) + lin.bias
            model[2](torch.nn.functional.silu(h)).square().sum().backward()
            b, off = ddp._slot[id(lin.weight)]
            assert lin.weight.grad.data_ptr() == b.flat.data_ptr() + 4 * off, "weight grad must live in the bucket (no copy)"
            ddp.finish()
            # reference: mean over ranks of the plain-autograd gradient
            exp = 0
            for r in range(world):
                torch.manual_seed(0)
                ref = nn.Sequential(nn.Linear(16, 8), nn.SiLU(), nn.Linear(8, 4))
                ref.load_state_dict(model.state_dict())
                torch.manual_seed(10 + r + 7 * step)
                ref(torch.randn(5, 16)).square().sum().backward()
                exp = exp + ref[0].weight.grad / world
            assert torch.allclose(lin.weight.grad, exp, rtol=1e-5, atol=1e-6), f"rank {rank} step {step}"
            model.zero_grad(set_to_none=True)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_weight_gradients_land_in_the_bucket_without_a_copy():
    """Persistent bucket views: a backward that writes its weight gradient into functional.grad_sink(param) makes
    autograd adopt the bucket view as .grad (no per-step packing copy), and the all-reduced result equals plain DDP."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sink_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


# ------------------------------------------------------------------ gradient accumulation under GradAllReduce (round 3)
class _SinkLinearFn(torch.autograd.Function):
    """y = x W^T whose weight gradient is written where functional.grad_sink(W) says -- the protocol of
    functional.layer_wgrads (the grouped HIP weight-gradient launch), restated with torch ops so that it runs on CPU."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        from vtx import functional as VF
        x, w = ctx.saved_tensors
        sink = VF.grad_sink(w)
        dW = dy.t() @ x
        if sink is not None:
            sink.copy_(dW)                  # "the kernel's final sum lands in the bucket slot"
            dW = sink
        return dy @ w, dW


class _SharedNet(nn.Module):
    """One weight feeding TWO graph nodes of a backward (DINO's backbone: one pass per crop resolution) + a plain layer."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.w = nn.Parameter(torch.randn(12, 12) * 0.3)
        self.odd = nn.Parameter(torch.randn(507))          # like Swin's rel_pos table: leaves the next slot unaligned
        self.head = nn.Linear(12, 3)

    def forward(self, x):
        h = torch.tanh(_SinkLinearFn.apply(x, self.w))
        h = torch.tanh(_SinkLinearFn.apply(h, self.w))     # same Parameter, second node
        return self.head(h) + self.odd.sum() * 1e-3


def _accum_worker(rank, world, port, q, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vtx.ddp import SLOT_ALIGN, GradAllReduce
        from vtx.train_step import accumulation_boundary, backward_ddp
        model = _SharedNet()
        ddp = GradAllReduce(model, bucket_bytes=1500, first_bucket_bytes=100)
        assert len(ddp.buckets) >= 2
        for b in ddp.buckets:                              # slots on 128-byte boundaries
            assert all(o % SLOT_ALIGN == 0 for o in b.offsets)
            assert all(v.data_ptr() % (4 * SLOT_ALIGN) == b.flat.data_ptr() % (4 * SLOT_ALIGN) for v in b.views)
        accum = 2
        torch.manual_seed(7)
        xs = torch.randn(world * accum, 5, 12)             # micro-batch (rank r, index i) = xs[r * accum + i]
        ys = torch.randn(world * accum, 5, 3)

        # what one process sees on the concatenated batch: mean over all world * accum micro-batch losses
        ref = _SharedNet()
        sum(((ref(xs[k]) - ys[k]) ** 2).mean() for k in range(world * accum)).div(world * accum).backward()
        exp = [p.grad.clone() for p in ref.parameters()]

        for rep in range(2):                               # twice: finish() must re-arm buckets AND sinks
            for i in range(accum):
                k = rank * accum + i
                loss = ((model(xs[k]) - ys[k]) ** 2).mean() / accum
                boundary = accumulation_boundary(accum, i)
                backward_ddp(loss, ddp, boundary, mode, fresh=(i == 0))
            ddp.finish()
            for (n, p), e in zip(model.named_parameters(), exp):
                assert torch.allclose(p.grad, e, rtol=1e-5, atol=1e-6), f"{mode} rep {rep}: {n} wrong " \
                    f"(max err {(p.grad - e).abs().max():.3e} vs max |g| {e.abs().max():.3e})"
            model.zero_grad(set_to_none=True)

        # epoch tail (train.py:285 second clause): a loader of 3 micro-batches with windows of 2 steps at i = 1 AND at i = 2;
        # the tail window holds one micro-batch (still divided by grad_accum, train.py:281) and the epoch ends on clean gradients
        loader_len = 3
        assert [accumulation_boundary(accum, i, loader_len) for i in range(loader_len)] == [False, True, True]
        assert [accumulation_boundary(accum, i) for i in range(loader_len)] == [False, True, False]
        stepped = []
        for i in range(loader_len):
            k = (rank * accum + i) % (world * accum)
            loss = ((model(xs[k]) - ys[k]) ** 2).mean() / accum
            boundary = accumulation_boundary(accum, i, loader_len)
            backward_ddp(loss, ddp, boundary, mode, fresh=(i % accum == 0))
            if boundary:
                ddp.finish()
                stepped.append(i)
                if i == loader_len - 1:
                    ref2 = _SharedNet()
                    sum(((ref2(xs[(r * accum + i) % (world * accum)]) - ys[(r * accum + i) % (world * accum)]) ** 2).mean() / accum
                        for r in range(world)).div(world).backward()
                    for (n, p), e in zip(model.named_parameters(), [q_.grad for q_ in ref2.parameters()]):
                        assert torch.allclose(p.grad, e, rtol=1e-5, atol=1e-6), f"{mode} epoch tail: {n} wrong"
                model.zero_grad(set_to_none=True)
        assert stepped == [1, 2]

        # a forgotten finish() is reported as what it is, at the backward that trips over it
        loss = ((model(xs[0]) - ys[0]) ** 2).mean()
        loss.backward()
        try:
            ((model(xs[1]) - ys[1]) ** 2).mean().backward()
            raised = ""
        except RuntimeError as e:
            raised = str(e)
        assert "no_sync" in raised and "finish()" in raised, raised
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["boundary", "every"])
def test_grad_accumulation_world2_equals_one_process_on_the_concatenated_batch(mode):
    """VERDICT r2 #1 / ADVICE r2: `grad_accumulation: 2` (config/swin-transformer-s.conf:33) under data parallelism.  2 ranks x
    2 micro-batches must give every rank the gradient of ONE process on the concatenated batch -- with the all-reduce on the
    boundary only (no_sync in between) and with the reference's all-reduce on every micro-batch (train.py:283-299 under DDP).
    The model shares one weight between two graph nodes (both nodes were handed the same bucket slot)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_accum_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_micro_step_is_required_when_accumulating():
    """a defaulted micro_step = 0 with grad_accum > 1 never reached the optimizer step."""
    from vtx.train_step import accumulation_boundary
    assert accumulation_boundary(1, None) and accumulation_boundary(1, 5)
    assert [accumulation_boundary(2, i) for i in range(4)] == [False, True, False, True]
    with pytest.raises(ValueError, match="micro_step"):
        accumulation_boundary(2, None)


class _TwoOrders(nn.Module):
    """Two Linears whose ORDER in the graph is chosen per call: their gradients arrive in the opposite order."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.a = nn.Linear(8, 8)
        self.b = nn.Linear(8, 8)

    def forward(self, x, swap):
        return self.a(self.b(x)) if swap else self.b(self.a(x))


def _order_worker(rank, world, port, q, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vtx.ddp import GradAllReduce
        torch.manual_seed(11)
        xs = [torch.randn(5, 8) for _ in range(6)]

        def expected(pairs):
            m = _TwoOrders()
            for x, swap in pairs:
                m(x, swap).square().sum().backward()
            return [p.grad.clone() for p in m.parameters()]

        model = _TwoOrders()
        ddp = GradAllReduce(model, bucket_bytes=1 << 20, first_bucket_bytes=1 << 20, force=True)   # one bucket, 4 parameters
        assert len(ddp.buckets) == 1

        def step(pairs, mode):
            """one optimizer step of len(pairs) micro-batches"""
            for i, (x, swap) in enumerate(pairs):
                last = i == len(pairs) - 1
                if mode == "boundary" and not last:
                    with ddp.no_sync():
                        model(x, swap).square().sum().backward()
                else:
                    model(x, swap).square().sum().backward()
                    if mode == "every" or last:
                        ddp.finish()
            got = [p.grad.clone() for p in model.parameters()]
            exp = expected(pairs)
            for g, e, (n, _) in zip(got, exp, model.named_parameters()):
                assert torch.allclose(g, e, atol=1e-5), f"{case}: {n} max err {(g - e).abs().max().item():.3g}"

        # learn the arrival order on plain steps (bucket-level trigger adopted after the first)
        for k in range(2):
            step([(xs[k], False)], "plain")
            model.zero_grad(set_to_none=True)
        assert ddp.buckets[0].trigger is not None
        if case == "no_sync_swapped":
            # accumulated gradients exist at the entry of the boundary backward and the order is the other one
            step([(xs[2], False), (xs[3], True)], "boundary")
            model.zero_grad(set_to_none=True)
            step([(xs[4], True), (xs[5], False)], "boundary")
        elif case == "every_swapped":
            step([(xs[2], False), (xs[3], True)], "every")
        elif case == "zero_grad_in_place":
            model.zero_grad(set_to_none=False)         # the gradients installed by finish() stay, zeroed
            step([(xs[2], True)], "plain")
            model.zero_grad(set_to_none=False)
            step([(xs[3], False)], "plain")
        elif case == "finish_twice":
            try:
                ddp.finish()
            except RuntimeError as e:
                assert "no backward() ran outside no_sync()" in str(e)
            else:
                raise AssertionError("a second finish() in a row must raise")
            with ddp.no_sync():
                model(xs[2], False).square().sum().backward()
            try:
                ddp.finish()
            except RuntimeError as e:
                assert "no backward() ran outside no_sync()" in str(e)
            else:
                raise AssertionError("finish() after only no_sync backwards must raise")
            model.zero_grad(set_to_none=True)
            step([(xs[3], False)], "plain")             # still usable
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["no_sync_swapped", "every_swapped", "zero_grad_in_place", "finish_twice"])
def test_bucket_trigger_never_reduces_before_the_last_gradient(case):
    """with gradients present at backward entry the bucket-level trigger's `grad is not None` check is
    vacuous; a changed arrival order then reduced the bucket early (max error 5.2 in the advisor's reproduction)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    p = ctx.Process(target=_order_worker, args=(0, 1, port, q, case))
    p.start()
    rank, msg = q.get(timeout=180)
    p.join(timeout=60)
    assert msg == "ok", msg
