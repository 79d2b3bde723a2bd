"""-m gpu: parity AT THE BENCHMARK'S OWN DISPATCH.

The small-shape kernel tests never reach the kernel variants the headline numbers run on: ``glds_pick_bm`` chooses the
128 x 128 x 64 2 x 4-wave LDS-DMA GEMM only for launches with >= 800 such tiles, the grouped weight-gradient launch
only splits 4-ways at the real token counts, the persistent window-attention loop runs > 1 round per wave only with
thousands of windows per head.  Here the exact launches of bench.py (Swin-S B = 128 / ViT-S/16 B = 256, bf16) are
compared with the fp64 CPU oracle, and every dispatch switch (vtx.options) is flipped in-process to check that the
variants agree BIT FOR BIT where they must (same summation order per output element).

Tolerances as in gpu_util.TOL: bf16-stored outputs 4e-3 (one bf16 rounding of an fp64-exact value), fp32 gradient
outputs 2e-5.
"""
import pytest
import torch

from gpu_util import TOL, check, dev
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _mk(shape, seed, dtype, scale=1.0, device="cpu"):
    g = torch.Generator(device=device).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=device) * scale).to(dtype)


def _mm64(a, b_t):
    """a [M, K] @ b_t [N, K]^T in fp64 on the CPU (the oracle of every GEMM here)."""
    return a.double() @ b_t.double().t()


BIG128 = "gemm_glds_pv_kernel<128, 2, false>"            # 128 x 128 tiles, 4 x 2 waves, wave-private epilogue (the benchmark default for K > 384)


def _bench_kernel(K, astat, M=0, N=0):
    """The kernel the benchmark runs these shapes on: round 5's two-group kernel for the long contractions (option GEMM_PP, default 1),
    round 4's A-stationary persistent kernel for 192 <= K <= 384 (option GEMM_ASTAT, default 1), the tiled LDS-DMA kernel otherwise."""
    from vtx import ops
    if ops.pp_ok(N, K, M):
        return f"gemm_pp_kernel<{ops.pp_wmf(M, N)}, false, 0>"
    return f"gemm_astat_kernel<{K // 64}, false, ...>" if (astat and 192 <= K <= 384) else BIG128


@pytest.fixture(params=[1, 0], ids=["astat", "tiled"])
def astat(request):
    from vtx import options
    with options.override(GEMM_ASTAT=request.param):
        yield request.param


# ------------------------------------------------------------------ the 128-row LDS-DMA GEMM with every fused epilogue
@pytest.mark.parametrize("M,N,K,T", [(25088, 1536, 384, 196),      # Swin-S stage-3 fc1 / fc2-dgrad, B = 128
                                     (50432, 1536, 384, 197),      # ViT-S/16 fc1 / fc2-dgrad, B = 256
                                     (100352, 768, 192, 784)])     # Swin-S stage-2 fc1 / fc2-dgrad
def test_glds128_silu_and_dsilu_epilogues_vs_oracle(M, N, K, T, astat):
    from vtx import ops
    d = dev()
    assert ops.gemm_kernel_name(BF, N, 0, K=K, M=M) == _bench_kernel(K, astat), "this shape must dispatch to the benchmark's kernel"
    x = _mk((M, K), 101, BF)
    w1 = _mk((N, K), 102, BF, 0.05)
    b1 = _mk((N,), 103, torch.float32, 0.1)
    # forward: z = x W1^T + b1 (saved), h = silu(rounded z)
    h, z = ops.gemm(x.to(d), w1.to(d), 0, bias=b1.to(d), act=ops.ACT_SILU, want_aux=True)
    zr = _mm64(x, w1) + b1.double()
    check(f"glds128 fc1 z {M}x{N}x{K}", z, zr, TOL[BF]["out"])
    zq = z.cpu().double()
    check(f"glds128 fc1 silu(z) {M}x{N}x{K}", h, R.silu(zq), TOL[BF]["out"])
    del zr
    # backward of fc2 through DropPath and SiLU: dz = s[row] * (dy W2) * silu'(z)  -- N, K as above (the transposed problem)
    B = M // T
    keep = (torch.rand(B, generator=torch.Generator().manual_seed(104)) < 0.8).float() / 0.8
    dy = _mk((M, K), 105, BF)
    w2t = _mk((N, K), 106, BF, 0.05)              # the transposed weight copy [in = N][out = K] the dgrad kernel reads
    assert ops.gemm_kernel_name(BF, N, 0, K=K, M=M) == _bench_kernel(K, astat)
    dz = ops.gemm(dy.to(d), w2t.to(d), 0, act=ops.ACT_DSILU, aux_in=z, rowscale=keep.to(d), rows_per_scale=T)
    s = torch.sigmoid(zq)
    dzr = keep.double().repeat_interleave(T)[:, None] * _mm64(dy, w2t) * (s * (1 + zq * (1 - s)))
    check(f"glds128 fc2-dgrad dsilu+droppath {M}x{N}x{K}", dz, dzr, TOL[BF]["out"])


@pytest.mark.parametrize("M,N,K,T", [(50432, 384, 1536, 197),      # ViT-S/16 fc2 + DropPath + residual, B = 256
                                     (50432, 384, 384, 197)])      # ViT-S/16 attention projection
def test_glds128_droppath_residual_epilogue_vs_oracle(M, N, K, T, astat):
    from vtx import ops
    d = dev()
    assert ops.gemm_kernel_name(BF, N, 0, K=K, M=M) == _bench_kernel(K, astat, M, N)
    assert K != 1536 or "gemm_pp_kernel" in ops.gemm_kernel_name(BF, N, 0, K=K, M=M)    # (ViT-S/16 fc2: the two-group kernel)
    hh = _mk((M, K), 111, BF)
    w = _mk((N, K), 112, BF, 0.05)
    b = _mk((N,), 113, torch.float32, 0.1)
    res = _mk((M, N), 114, BF)
    keep = (torch.rand(M // T, generator=torch.Generator().manual_seed(115)) < 0.9).float() / 0.9
    y = ops.gemm(hh.to(d), w.to(d), 0, bias=b.to(d), resid=res.to(d), rowscale=keep.to(d), rows_per_scale=T)
    yr = res.double() + keep.double().repeat_interleave(T)[:, None] * (_mm64(hh, w) + b.double())
    check(f"glds128 bias+droppath+residual {M}x{N}x{K}", y, yr, TOL[BF]["out"])


def test_glds_tile_and_wave_variants_are_bitwise_identical():
    """64- vs 128-row tiles, 2 x 2 vs 2 x 4 vs 4 x 2 waves and the shared vs wave-private epilogue staging only change which
    wave / lane owns an output element, not the order its products are summed in: the epilogues must agree bit for bit."""
    from vtx import ops, options
    d = dev()
    M, N, K, T = 25088, 1536, 384, 196
    x, w, b = _mk((M, K), 121, BF, device=d), _mk((N, K), 122, BF, 0.05, device=d), _mk((N,), 123, torch.float32, 0.1, device=d)
    dy, res = _mk((M, K), 124, BF, device=d), _mk((M, N), 125, BF, device=d)
    keep = ((torch.rand(M // T, device=d) < 0.8).float() / 0.8)

    def run():
        h, z = ops.gemm(x, w, 0, bias=b, act=ops.ACT_SILU, want_aux=True)
        dz = ops.gemm(dy, w, 0, act=ops.ACT_DSILU, aux_in=z, rowscale=keep, rows_per_scale=T)
        y = ops.gemm(x, w, 0, bias=b, resid=res, rowscale=keep, rows_per_scale=T)
        return h, z, dz, y

    with options.override(GEMM_ASTAT=0):
        base = run()
        assert options.get("GLDS_EPI") == 1 and ops.gemm_kernel_name(BF, N, 0, K=K, M=M) == BIG128
    # (round 4) the A-stationary persistent kernel: plain product with permuted weight rows, epilogue straight from the accumulators
    assert options.get("GEMM_ASTAT") == 1 and ops.gemm_kernel_name(BF, N, 0, K=K, M=M).startswith("gemm_astat_kernel<6, false")
    for a, g, name in zip(base, run(), ("h", "z", "dz", "y")):
        assert torch.equal(a, g), f"{name} differs between the tiled and the A-stationary kernel"
    for kw in (dict(GLDS_BM=64), dict(GLDS_BM=128), dict(GLDS_EPI=0), dict(GLDS_EPI=0, GLDS_BM=64), dict(GLDS_EPI=0, GLDS_BM=128),
               dict(GLDS_EPI=0, GLDS_BM=128, GLDS_WAVES=4), dict(GLDS_EPI=0, GLDS_BM=64, GLDS_WAVES=4)):
        with options.override(GEMM_ASTAT=0, **kw):
            assert ops.gemm_kernel_name(BF, N, 0, K=K, M=M) != BIG128 or kw == dict(GLDS_BM=128)
            got = run()
        for a, g, name in zip(base, got, ("h", "z", "dz", "y")):
            assert torch.equal(a, g), f"{name} differs under {kw}"


@pytest.mark.parametrize("M,N,K,T", [(25088, 384, 1536, 196),      # Swin-S stage-3 fc2 forward / fc1 dgrad, B = 128 (224 tiles of 224 x 192)
                                     (50432, 384, 1152, 197),      # ViT-S/16 qkv dgrad, B = 256 (two rounds)
                                     (111 * 196, 384, 1536, 196),  # a compacted stage-3 row count (192-row tiles, ragged last tile)
                                     (6272, 768, 3072, 49),        # Swin-S stage-4 fc2 forward (128-row tiles, four column tiles per panel)
                                     (100352, 192, 768, 784),      # Swin-S stage-2 fc2 forward (one column tile)
                                     (1000, 576, 128, 50),         # two k-tiles: the ring never wraps
                                     (25088, 256, 1024, 196),      # (round 5) 256-column tiles: Twins-SVT-S stage-3 fc2 forward / fc1 dgrad
                                     (6272 + 49, 512, 2048, 49),   # two 256-column tiles per panel, ragged last row tile
                                     (100352, 128, 1024, 784),     # 128-column tiles: PVT-Small stage-2 fc2 forward
                                     (3000, 640, 192, 60)])        # N = 640 = 5 x 128, three k-tiles
def test_two_group_gemm_is_bitwise_the_tiled_kernel_and_matches_the_oracle(M, N, K, T):
    """Round 5: gemm_pp_kernel (BM x 192 tiles, one workgroup per CU, two wave groups half a k-step apart, ring of three 64-deep
    k-tiles) accumulates the same products in the same k order and applies the same epilogue expression as the tiled kernels:
    bit-identical for every epilogue and forced tile height; one epilogue against the fp64 oracle."""
    from vtx import ops, options
    d = dev()
    x, w, b = _mk((M, K), 161, BF, device=d), _mk((N, K), 162, BF, 0.05, device=d), _mk((N,), 163, torch.float32, 0.1, device=d)
    res, z = _mk((M, N), 164, BF, device=d), _mk((M, N), 165, BF, device=d)
    keep = ((torch.rand((M + T - 1) // T, device=d, generator=torch.Generator(device=d).manual_seed(166)) < 0.8).float() / 0.8)

    def run():
        y = ops.gemm(x, w, 0, bias=b, resid=res, rowscale=keep, rows_per_scale=T)
        h, zz = ops.gemm(x, w, 0, bias=b, act=ops.ACT_SILU, want_aux=True)
        dz = ops.gemm(x, w, 0, act=ops.ACT_DSILU, aux_in=z, rowscale=keep, rows_per_scale=T)
        return y, h, zz, dz, ops.gemm(x, w, 0)

    with options.override(GEMM_PP=0):
        assert "gemm_pp" not in ops.gemm_kernel_name(BF, N, 0, K=K, M=M)
        base = run()
    for wmf in (0, 4, 5, 6, 7):
        with options.override(GEMM_PP=100 + wmf if wmf else 2):
            kn = ops.gemm_kernel_name(BF, N, 0, K=K, M=M)
            assert ("gemm_pp_kernel" if N % 192 == 0 else f"gemm_ppn_kernel<{min(wmf, 5 if N % 256 == 0 else 7) or ops.pp_wmf(M, N)}") in kn, kn
            got = run()
        for a, g, name in zip(base, got, ("y", "h", "z", "dz", "plain")):
            assert torch.equal(a, g), f"{name} differs between the tiled and the two-group kernel (tile height {32 * wmf or 'auto'})"
    if M <= 30000:
        yr = res.cpu().double() + keep.cpu().double().repeat_interleave(T)[:M, None] * (_mm64(x.cpu(), w.cpu()) + b.cpu().double())
        check(f"two-group gemm bias+droppath+residual {M}x{N}x{K}", base[0], yr, TOL[BF]["out"])


@pytest.mark.parametrize("M,N,K,T", [(130 * 197, 1152, 384, 197), (83 * 196, 1536, 384, 196), (301 * 49, 768, 192, 49)])
def test_a_stationary_gemm_with_a_partial_last_strip_is_bitwise_the_tiled_kernel(M, N, K, T):
    """Round 4: rows past M in the last 128-row strip are computed as copies of the last row (same bits to the same address), so that
    every lane stores every epilogue vector (the kernel's counted waits rely on that); M = the row count of a compacted branch is any
    multiple of the tokens per sample."""
    from vtx import ops, options
    d = dev()
    assert M % 128 != 0
    x, w, b = _mk((M, K), 131, BF, device=d), _mk((N, K), 132, BF, 0.05, device=d), _mk((N,), 133, torch.float32, 0.1, device=d)
    dy, res = _mk((M, K), 134, BF, device=d), _mk((M, N), 135, BF, device=d)
    keep = ((torch.rand(M // T, device=d) < 0.8).float() / 0.8)

    def run():
        q = ops.gemm(x, w, 0, bias=b)
        h, z = ops.gemm(x, w, 0, bias=b, act=ops.ACT_SILU, want_aux=True)
        dz = ops.gemm(dy, w, 0, act=ops.ACT_DSILU, aux_in=z, rowscale=keep, rows_per_scale=T)
        y = ops.gemm(x, w, 0, bias=b, resid=res, rowscale=keep, rows_per_scale=T)
        return q, h, z, dz, y

    with options.override(GEMM_ASTAT=0):
        base = run()
    with options.override(GEMM_ASTAT=2):
        assert ops.gemm_kernel_name(BF, N, 0, K=K, M=M).startswith("gemm_astat_kernel")
        got = run()
    for a, g, name in zip(base, got, ("q", "h", "z", "dz", "y")):
        assert torch.equal(a, g), f"{name} differs"


@pytest.mark.parametrize("M,N,K", [(100352, 576, 192), (128 * 49 * 4 + 77, 576, 192), (25088 + 5, 320, 384), (12800, 448, 256), (3000, 832, 320)])
def test_a_stationary_gemm_with_a_ragged_last_column_tile_is_bitwise_the_tiled_kernel(M, N, K):
    """Round 5: N % 128 == 64 (Swin-S stage-2 qkv forward, N = 576).  In the last column tile of a strip the second wave column works on
    the first one's 64 columns again (same weight rows, same bits to the same addresses), so every wave still stores every vector;
    plain and bias-only launches only -- the other kinds stay on the tiled kernel."""
    from vtx import ops, options
    d = dev()
    assert N % 128 == 64
    x, w, b = _mk((M, K), 191, BF, device=d), _mk((N, K), 192, BF, 0.05, device=d), _mk((N,), 193, torch.float32, 0.1, device=d)
    res = _mk((M, N), 195, BF, device=d)

    def run():
        return ops.gemm(x, w, 0, bias=b), ops.gemm(x, w, 0), ops.gemm(x, w, 0, bias=b, resid=res)

    with options.override(GEMM_ASTAT=0, GEMM_PP=0):
        assert not ops.gemm_kernel_name(BF, N, 0, K=K, M=M).startswith("gemm_astat_kernel")
        base = run()
    with options.override(GEMM_ASTAT=2, GEMM_PP=0):
        assert ops.gemm_kernel_name(BF, N, 0, K=K, M=M).startswith("gemm_astat_kernel")
        assert not ops.gemm_kernel_name(BF, N, 0, K=K, M=M, vec=True).startswith("gemm_astat_kernel")
        got = run()
    for a, g, name in zip(base, got, ("bias", "plain", "bias + residual (tiled kernel in both runs)")):
        assert torch.equal(a, g), f"{name} differs"
    if M == 100352:
        assert ops.gemm_kernel_name(BF, N, 0, K=K, M=M).startswith("gemm_astat_kernel<3"), "the stage-2 qkv forward must take this kernel by default"
    check(f"astat ragged N {M}x{N}x{K}", got[0], _mm64(x.cpu(), w.cpu()) + b.cpu().double(), TOL[BF]["out"])


@pytest.mark.parametrize("M,N,K,T", [(66395, 288, 96, 49), (66395, 96, 96, 49), (66395, 384, 96, 49),
                                     (40100, 128, 64, 100), (33100, 256, 128, 100), (36100, 512, 64, 100),
                                     (33100, 1024, 128, 100), (33100, 2048, 64, 100)])   # (column chunks: 2, 4)
def test_weight_resident_streaming_gemm_vs_oracle_and_the_tiled_kernels(M, N, K, T):
    """Round 3: bf16 GEMMs with a short contraction (K = 64 / 96 / 128: Swin-S stage 1, PVT stages 1-2) over >= 32 768 rows
    take gemm_skinny_kernel (whole weight resident in LDS, A streamed into MFMA registers, transposed product, 16-byte
    epilogue vectors).  Every fused epilogue vs the fp64 oracle, and bit for bit against the tiled kernels (same products,
    same k order inside the MFMA) -- at row counts that are no multiple of the 32-row wave block."""
    from vtx import ops, options
    d = dev()
    assert ops.gemm_kernel_name(BF, N, 0, K=K, M=M) == f"gemm_skinny_kernel<{K // 32}>"
    assert "skinny" not in ops.gemm_kernel_name(BF, N, 0, K=K, M=M, vec=True)      # (residual / z epilogues: tiled by default)
    x = _mk((M, K), 151, BF)
    w = _mk((N, K), 152, BF, 0.1)
    b = _mk((N,), 153, torch.float32, 0.1)
    res = _mk((M, N), 154, BF)
    zin = _mk((M, N), 155, BF)
    keep = (torch.rand(M // T, generator=torch.Generator().manual_seed(156)) < 0.8).float() / 0.8
    xd, wd, bd, rd, zd, kd = (t.to(d) for t in (x, w, b, res, zin, keep))

    def run():
        y0 = ops.gemm(xd, wd, 0, bias=bd)
        h, z = ops.gemm(xd, wd, 0, bias=bd, act=ops.ACT_SILU, want_aux=True)
        dz = ops.gemm(xd, wd, 0, act=ops.ACT_DSILU, aux_in=zd, rowscale=kd, rows_per_scale=T)
        y = ops.gemm(xd, wd, 0, bias=bd, resid=rd, rowscale=kd, rows_per_scale=T)
        g, zg = ops.gemm(xd, wd, 0, bias=bd, act=ops.ACT_GELU, want_aux=True)
        return y0, h, z, dz, y, g, zg

    with options.override(GEMM_SKINNY=2):              # every epilogue on the streaming kernel
        got = run()
    acc = _mm64(x, w)
    ks = keep.double().repeat_interleave(T)[:, None]
    zr = acc + b.double()
    check(f"skinny bias {M}x{N}x{K}", got[0], zr, TOL[BF]["out"])
    check(f"skinny z {M}x{N}x{K}", got[2], zr, TOL[BF]["out"])
    check(f"skinny silu(z) {M}x{N}x{K}", got[1], R.silu(got[2].cpu().double()), TOL[BF]["out"])
    sz = torch.sigmoid(zin.double())
    check(f"skinny dsilu+droppath {M}x{N}x{K}", got[3], ks * acc * (sz * (1 + zin.double() * (1 - sz))), TOL[BF]["out"])
    check(f"skinny bias+droppath+residual {M}x{N}x{K}", got[4], res.double() + ks * zr, TOL[BF]["out"])
    check(f"skinny gelu(z) {M}x{N}x{K}", got[5], torch.nn.functional.gelu(got[6].cpu().double()), TOL[BF]["out"])
    with options.override(GEMM_SKINNY=0):
        assert "skinny" not in ops.gemm_kernel_name(BF, N, 0, K=K, M=M)
        tiled = run()
    for a, t_, name in zip(got, tiled, ("bias", "silu", "z", "dsilu", "resid", "gelu", "z(gelu)")):
        assert torch.equal(a, t_), f"{name}: the streaming kernel differs from the tiled kernel at {(M, N, K)}"


@pytest.mark.parametrize("H,nH", [(14, 12), (56, 3)])
def test_wattn_backward_four_waves_per_problem_matches_the_one_wave_kernel_at_bench_size(H, nH):
    """Round 3: the bf16 window-attention backward shares a problem between the four waves of its workgroup
    (wattn_bwd4_kernel; option WATTN_BWD4).  Same products in the same order: dqkv must equal the one-wave kernel's bit for
    bit at the Swin-S B = 128 geometry (stage 3 and stage 1, shifted windows), the rel_pos gradient up to fp32 summation
    order, and two launches must agree bit for bit (fixed-order gather and reduce)."""
    from oracle import tables
    from vtx import ops, options
    from vtx.tables import mask_regions
    d = dev()
    B, win, L, ntab = 128, 7, 49, 169
    pos_np, mask_np = tables.make_pos_mask((H, H), win, True)
    pos = torch.from_numpy(pos_np).to(d)
    region, ok = mask_regions(torch.from_numpy(mask_np).to(d))
    assert ok
    qkv = _mk((B * H * H, 3 * nH * 32), 141, BF, device=d)
    do = _mk((B * H * H, nH * 32), 142, BF, device=d)
    rel = _mk((ntab, nH), 143, torch.float32, 0.5, device=d)
    swin = (H, H, win, True)
    o, lse = ops.wattn_fwd(qkv, rel, pos, region, B, L, nH, swin)
    assert ops.wattn_bwd_kernel_name(BF, True) == "wattn_bwd4_kernel<true>"
    a = ops.wattn_bwd(qkv, o, do, lse, rel, pos, region, B, L, nH, swin, ntab)
    b = ops.wattn_bwd(qkv, o, do, lse, rel, pos, region, B, L, nH, swin, ntab)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), "the four-wave backward is not deterministic"
    with options.override(WATTN_BWD4=0):
        assert ops.wattn_bwd_kernel_name(BF, True) == "wattn_bwd_kernel<__bf16, true>"
        c = ops.wattn_bwd(qkv, o, do, lse, rel, pos, region, B, L, nH, swin, ntab)
    assert torch.equal(a[0], c[0]), "dqkv of the four-wave kernel differs from the one-wave kernel"
    check(f"wattn drel_pos, four waves vs one wave per problem (H {H})", a[1], c[1].double(), 2e-5)


@pytest.mark.parametrize("B,H,nH,win,shift", [(128, 56, 3, 7, True), (128, 56, 3, 7, False), (128, 14, 12, 7, True), (128, 7, 24, 7, False),
                                              (3, 28, 6, 7, True), (5, 16, 2, 4, True), (2, 20, 3, 5, False), (1, 12, 2, 6, True)])
def test_wattn_forward_four_waves_per_problem_is_bitwise_the_one_wave_kernel(B, H, nH, win, shift):
    """Round 5: the bf16 window-attention forward shares a problem between the four waves of its workgroup (wattn_fwd4_kernel; option
    WATTN_FWD4): one 16-token tile per wave, rotating from problem to problem, K / V through plain LDS images, next-problem prefetch.
    Same products in the same order: o and lse must equal the one-wave kernel's bit for bit, at the Swin-S B = 128 geometries (several
    problems per workgroup: every rotation), at odd problem counts and at windows with padding-only tiles (16, 25, 36 tokens), under
    every WATTN_FAST setting."""
    from oracle import tables
    from vtx import ops, options
    from vtx.tables import mask_regions
    d = dev()
    L, ntab, D = win * win, (2 * win - 1) ** 2, 32
    pos_np, mask_np = tables.make_pos_mask((H, H), win, shift)
    pos = torch.from_numpy(pos_np).to(d)
    region = None
    if shift:
        region, ok = mask_regions(torch.from_numpy(mask_np).to(d))
        assert ok
    qkv = _mk((B * H * H, 3 * nH * D), 181, BF, device=d)
    rel = _mk((ntab, nH), 183, torch.float32, 0.5, device=d)
    swin = (H, H, win, shift)
    nbn = B * (H // win) ** 2
    assert ops.wattn_fwd_kernel_name(BF, shift, nbn).startswith("wattn_fwd4_kernel" if nbn >= 4096 else "wattn_fwd_kernel<__bf16")
    for fast in (3, 0, 1, 2):
        with options.override(WATTN_FAST=fast, WATTN_FWD4=2):
            assert ops.wattn_fwd_kernel_name(BF, shift, nbn) == f"wattn_fwd4_kernel<{'true' if shift else 'false'}>"
            a = ops.wattn_fwd(qkv, rel, pos, region, B, L, nH, swin)
            with options.override(WATTN_FWD4=0):
                assert ops.wattn_fwd_kernel_name(BF, shift, nbn).startswith("wattn_fwd_kernel<__bf16")
                b = ops.wattn_fwd(qkv, rel, pos, region, B, L, nH, swin)
        assert torch.equal(a[0], b[0]), f"o of the four-wave forward differs from the one-wave kernel (WATTN_FAST={fast})"
        assert torch.equal(a[1], b[1]), f"lse of the four-wave forward differs from the one-wave kernel (WATTN_FAST={fast})"
    if B <= 8:
        ref = R.window_attention_core(qkv.view(B, H, H, -1).cpu().double(), rel.cpu().double(), nH, D, win, shift)
        check(f"wattn fwd4 {H}x{H} win {win} h{nH} s{int(shift)}", a[0].view(B, H, H, -1), ref, TOL[BF]["out"] * 1.5)


@pytest.mark.parametrize("dtype", [BF, torch.float32])
@pytest.mark.parametrize("H,nH,shift", [(56, 3, True), (56, 3, False), (14, 12, True), (7, 24, False)])
@pytest.mark.parametrize("fwd4", [2, 0])
def test_wattn_forward_one_row_path_and_uniform_window_branch(H, nH, shift, dtype, fwd4):
    """Round 5 (option WATTN_FAST): bit 1 sends the windows of a shifted layer whose tokens share one region id down the
    unmasked instruction stream -- same bits as the masked stream; bit 0 runs the 49th query of a 7 x 7 window as ONE row
    (product taken as q k^T, 4 scores per lane, P through LDS) -- the other 48 queries bit for bit, the 49th within the
    rounding of the output type of the padded-tile path and of the fp64 oracle."""
    from oracle import tables
    from vtx import ops, options
    from vtx.tables import mask_regions
    d = dev()
    B, win, L, ntab, D = 16, 7, 49, 169, 32
    pos_np, mask_np = tables.make_pos_mask((H, H), win, shift)
    pos = torch.from_numpy(pos_np).to(d)
    region = None
    if shift:
        region, ok = mask_regions(torch.from_numpy(mask_np).to(d))
        assert ok
    qkv = _mk((B, H, H, 3 * nH * D), 171, dtype)
    rel = _mk((ntab, nH), 173, torch.float32, 0.5)
    qd, reld = qkv.to(d), rel.to(d)
    swin = (H, H, win, shift)
    out = {}
    for fast in (0, 1, 2, 3):
        with options.override(WATTN_FAST=fast, WATTN_FWD4=fwd4):
            out[fast] = ops.wattn_fwd(qd, reld, pos, region, B, L, nH, swin)
    assert torch.equal(out[2][0], out[0][0]) and torch.equal(out[2][1], out[0][1]), "uniform-window branch changed bits"
    assert torch.equal(out[3][0], out[1][0]) and torch.equal(out[3][1], out[0 + 1][1])
    # token 48 of every window = rows (y % 7 == 6, x % 7 == 6) after the roll; everything else must not move
    o0, o1 = out[0][0].view(B, H, H, nH * D), out[1][0].view(B, H, H, nH * D)
    sh = win // 2 if shift else 0
    last = torch.zeros(H, H, dtype=torch.bool, device=d)
    idx = torch.arange(H, device=d)
    sel = ((idx - sh) % H) % win == win - 1          # window coordinates are those of the rolled image (roll by -shift)
    last[sel[:, None] & sel[None, :]] = True
    assert torch.equal(o0[:, ~last], o1[:, ~last]), "the one-row path moved a query other than the 49th"
    ref = R.window_attention_core(qkv.double(), rel.double(), nH, D, win, shift)
    tol = TOL[dtype]["out"] * 1.5
    check(f"wattn fwd one-row path {H}x{H} h{nH} s{int(shift)}", out[3][0], ref, tol)
    check(f"wattn fwd 49th query, row vs tile {H}x{H}", o1[:, last], o0[:, last].double(), tol)
    lse0, lse1 = out[0][1].view(-1, L), out[1][1].view(-1, L)
    assert torch.equal(lse0[:, :48], lse1[:, :48])
    check("wattn lse of the 49th query", lse1[:, 48], lse0[:, 48].double(), 1e-5)


# ------------------------------------------------------------------ weight gradients at the real token counts
def _wgrad_ref(dy, x, keep, T, c):
    m = None if keep is None else (keep > 0).double().repeat_interleave(T)[:, None]
    dyd = dy.double() if m is None else m * dy.double()
    sc = 1.0 if keep is None else c
    return sc * (dyd.t() @ x.double()), sc * dyd.sum(0)


@pytest.mark.parametrize("B,T,N,Kin", [(128, 3136, 96, 384),       # Swin-S stage-1 fc2: 401 408 tokens
                                       (128, 196, 384, 1536),      # stage-3 fc2: 25 088 tokens
                                       (128, 196, 1536, 384)])     # stage-3 fc1 (no DropPath on its dy)
def test_wgrad_through_droppath_at_benchmark_size(B, T, N, Kin):
    from vtx import ops
    d = dev()
    M = B * T
    dy, x = _mk((M, N), 131, BF), _mk((M, Kin), 132, BF)
    c = 1.0 / 0.7
    keep = (torch.rand(B, generator=torch.Generator().manual_seed(133)) < 0.7).float() * c
    use = None if N == 1536 else keep
    assert ops.wgrad_kernel_name(BF, N, Kin, True) == "wgrad_glds_kernel<64, 2, 8, false>"
    dW, db = ops.wgrad(dy.to(d), x.to(d), rowscale=None if use is None else use.to(d), rows_per_scale=T,
                       scale_const=c if use is not None else 0.0)
    rW, rb = _wgrad_ref(dy, x, use, T, c)
    check(f"wgrad dW {M}x{N}x{Kin}", dW, rW, 2e-5)
    check(f"wgrad db {M}x{N}x{Kin}", db, rb, 2e-5)


def _layer_jobs(B, T, C, ff, d, seed=140):
    M = B * T
    c = 1.0 / 0.7
    g = torch.Generator().manual_seed(seed)
    s1 = ((torch.rand(B, generator=g) < 0.7).float() * c)
    s2 = ((torch.rand(B, generator=g) < 0.7).float() * c)
    mk = lambda n, k: _mk((M, n), seed + k, BF)
    dy, h, dz, ln2, dx1, o, dqkv, ln1 = mk(C, 1), mk(ff, 2), mk(ff, 3), mk(C, 4), mk(C, 5), mk(C, 6), mk(3 * C, 7), mk(C, 8)
    cpu = [(dy, h, True, s2), (dz, ln2, True, None), (dx1, o, True, s1), (dqkv, ln1, True, None)]
    gpu = [(a.to(d), b.to(d), wb, None if s is None else s.to(d)) for a, b, wb, s in cpu]
    return cpu, gpu, c


@pytest.mark.parametrize("B,T,C,ff", [(128, 196, 384, 1536),       # Swin-S stage 3 (18 of the 24 layers): 108 tiles x 4 slices
                                      (128, 3136, 96, 384),        # stage 1: ragged 96-wide tiles, 51 slices
                                      (256, 197, 384, 1536),       # ViT-S/16 B = 256
                                      (37, 197, 384, 1536),        # 7 289 tokens: slices that end inside a 32-token k-step of the wide-tile kernel
                                      (128, 784, 192, 768),        # Swin-S stage 2: 128 x 192 tiles, N = 576 / 192 end inside a row tile
                                      (128, 196, 320, 1280),       # PVT-Small stage 3: 128 x 320 tiles (third x panel half used), N = 320 / 960 ragged
                                      (128, 196, 256, 1024),       # Twins-SVT-S stage 3: 128 x 256 tiles
                                      (23, 49, 512, 2048),         # 512-wide stage at an odd token count: 96 tiles x 2 slices = 75 %: stays on 128 x 128
                                      (128, 49, 768, 3072)])       # Swin-S stage 4: 144 tiles of 384 columns = 56 % -> 216 tiles of 256 columns (84 %)
def test_grouped_layer_wgrads_vs_oracle_and_switches(B, T, C, ff):
    """The grouped launch of a layer's four weight gradients (fc2 and proj through DropPath) vs fp64; the grouped launch vs
    one launch per problem; 8 vs 4 waves: dW bit-identical; rerun: bit-identical (deterministic)."""
    from vtx import ops, options
    d = dev()
    cpu, gpu, c = _layer_jobs(B, T, C, ff, d)
    assert ops.wgrad_group_ok(gpu, T, c)
    # (128 x 256 tiles are opt-in, WGRAD_WIDE bit 3: no gain inside the Twins-SVT-S step; the C = 256 case runs them here)
    import contextlib
    j4 = options.override(WGRAD_WIDE=9) if C == 256 else contextlib.nullcontext()
    with j4:
        _grouped_wgrad_case(B, T, C, ff, d, cpu, gpu, c)


def _grouped_wgrad_case(B, T, C, ff, d, cpu, gpu, c):
    from vtx import ops, options
    res = ops.wgrad_group(gpu, T, c)
    for (dy, x, _, s), (dW, db), name in zip(cpu, res, ("fc2", "fc1", "proj", "qkv")):
        rW, rb = _wgrad_ref(dy, x, s, T, c)
        check(f"grouped wgrad dW {name} B{B} T{T} C{C}", dW, rW, 2e-5)
        check(f"grouped wgrad db {name} B{B} T{T} C{C}", db, rb, 2e-5)
    again = ops.wgrad_group(gpu, T, c)
    for (a, ab), (g, gb) in zip(res, again):
        assert torch.equal(a, g) and torch.equal(ab, gb), "grouped wgrad differs on a rerun"
    # C = 384 layers take the 128 x 384 tiles (one workgroup per CU, 7 slices); the 128 x 128 kernel (4 slices) sums the same
    # products in another slice partition
    wide, wj = ops.wgrad_wide_tiles([(j[0].shape[1], j[1].shape[1]) for j in gpu], want_j=True)
    # (the 512-wide group at 1 127 tokens: 96 tiles of 128 x 256 fill 192 of 256 CUs -- under the 85 % rule, it stays on 128 x 128)
    assert wj == {384: 6, 320: 5, 256: 4, 192: 3, 768: 4}.get(C, 0), (C, wide, wj)
    wide = wide > 0
    if wide and wj < 6:
        # the round-4 rule (whole 128 x 384 tiles only) leaves these widths on 128 x 128 tiles
        with options.override(WGRAD_WIDE=5):
            assert ops.wgrad_wide_tiles([(j[0].shape[1], j[1].shape[1]) for j in gpu]) == 0
        with options.override(WGRAD_WIDE=1):
            assert (ops.wgrad_wide_tiles([(j[0].shape[1], j[1].shape[1]) for j in gpu]) > 0) == (wj != 4 or C == 768)
    if wide and wj == 6:
        # option 1 (lockstep multiplying waves) vs the default 2 (two wave groups half a k-step apart): the same products in the same order
        with options.override(WGRAD_WIDE=3 - options.get("WGRAD_WIDE")):
            other = ops.wgrad_group(gpu, T, c)
        for (a, ab), (g, gb) in zip(other, res):
            assert torch.equal(a, g) and torch.equal(ab, gb), "wide-tile weight gradient: lockstep and two-group loops differ"
    with options.override(WGRAD_WIDE=0):
        narrow = ops.wgrad_group(gpu, T, c)
        again = ops.wgrad_group(gpu, T, c)
        with options.override(WG_WAVES=4):
            w4 = ops.wgrad_group(gpu, T, c)
    for (a, ab), (g, gb), name in zip(narrow, res, ("fc2", "fc1", "proj", "qkv")):
        if wide:
            check(f"grouped wgrad dW {name}, 128 x 64 J vs 128 x 128 tiles", g, a, 3e-6)
            check(f"grouped wgrad db {name}, 128 x 64 J vs 128 x 128 tiles", gb, ab, 3e-6)
        else:
            assert torch.equal(a, g) and torch.equal(ab, gb)
    for (a, ab), (g, gb) in zip(narrow, again):
        assert torch.equal(a, g) and torch.equal(ab, gb), "grouped wgrad (128 x 128 tiles) differs on a rerun"
    res = narrow
    # 4 waves: every dW element is summed in the same order (bit-identical); the bias gradient's row groups are
    # 16 instead of 32 per workgroup, i.e. a different (still fixed) summation order
    for (a, ab), (g, gb) in zip(res, w4):
        assert torch.equal(a, g), "grouped wgrad dW differs with 4 waves"
        check("grouped wgrad db, 4 vs 8 waves", gb, ab, 2e-6)


# ------------------------------------------------------------------ persistent window attention at B = 128
@pytest.mark.parametrize("H,nH,shift", [(56, 3, True), (56, 3, False), (14, 12, True)])
def test_window_attention_at_benchmark_batch(H, nH, shift):
    """B = 128: 8 192 (image, window) pairs per head at stage 1 (> 1 problem per persistent wave, several rounds), 512 at
    stage 3 -- forward, dqkv and the rel_pos gradient vs the fp64 oracle."""
    from oracle import tables
    from vtx import ops
    from vtx.tables import mask_regions
    d = dev()
    B, D, win = 128, 32, 7
    L, ntab = win * win, (2 * win - 1) ** 2
    qkv = _mk((B, H, H, 3 * nH * D), 161, BF)
    do = _mk((B, H, H, nH * D), 162, BF)
    rel = _mk((ntab, nH), 163, torch.float32, 0.5)
    pos_np, mask_np = tables.make_pos_mask((H, H), win, shift)
    pos = torch.from_numpy(pos_np).to(d)
    region = None
    if shift:
        region, ok = mask_regions(torch.from_numpy(mask_np).to(d))
        assert ok
    swin = (H, H, win, shift)
    qd, dod, reld = qkv.to(d), do.to(d), rel.to(d)
    o, lse = ops.wattn_fwd(qd, reld, pos, region, B, L, nH, swin)
    dqkv, drel = ops.wattn_bwd(qd, o, dod, lse, reld, pos, region, B, L, nH, swin, ntab)
    dqkv2, drel2 = ops.wattn_bwd(qd, o, dod, lse, reld, pos, region, B, L, nH, swin, ntab)
    assert torch.equal(drel, drel2) and torch.equal(dqkv, dqkv2), "not deterministic"
    # oracle in chunks of 16 images (fp64 autograd over the full batch would hold ~10 GB of scores at stage 1)
    orf, dqr = [], []
    drr = torch.zeros(ntab, nH, dtype=torch.float64)
    for i in range(0, B, 16):
        qr = qkv[i:i + 16].double().requires_grad_(True)
        rr = rel.double().requires_grad_(True)
        oo = R.window_attention_core(qr, rr, nH, D, win, shift)
        gq, gr = torch.autograd.grad(oo, [qr, rr], do[i:i + 16].double())
        orf.append(oo.detach()); dqr.append(gq); drr += gr
    tag = f"B128 {H}x{H} h{nH} s{int(shift)}"
    check(f"wattn fwd {tag}", o, torch.cat(orf), TOL[BF]["out"] * 1.5)
    check(f"wattn dqkv {tag}", dqkv, torch.cat(dqr), 1e-2)
    check(f"wattn drel_pos {tag}", drel, drr, 1e-2)


# ------------------------------------------------------------------ side-stream (deferred) weight gradients
def test_train_step_side_stream_wgrads_are_bitwise_the_single_stream_step():
    """vtx.train_step runs the layers' grouped weight gradients on a second HIP stream, joined once after backward; the
    parameters after two steps must equal those of the single-stream run bit for bit (every kernel is deterministic,
    the event graph fixes all cross-stream orderings)."""
    from models import SwinTransformer
    from vtx import functional as VF
    from vtx.optim import FusedAdamW
    from vtx.train_step import MixLoss, make_param_groups, train_step
    d = dev()
    cfg = dict(image_size=(224, 224), n_class=1000, depths=(2, 2, 2, 2), dims=(96, 192, 384, 768), dim_head=32,
               n_heads=(3, 6, 12, 24), dim_ffs=(384, 768, 1536, 3072), window_size=7, drop_path=0.2)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(16, 3, 224, 224, generator=gen).to(d)
    l1 = torch.randint(0, 1000, (16,), generator=gen).to(d)
    batch = (x, l1, l1.roll(1), torch.rand(16, generator=gen).to(d))

    def run(side):
        torch.manual_seed(0)
        model = SwinTransformer(**cfg).to(d).train()
        opt = FusedAdamW(make_param_groups(model.named_parameters(), 0.05, "vit"), lr=1e-3)
        old = VF._SIDE_ENABLED
        VF._SIDE_ENABLED = side
        try:
            torch.manual_seed(1)                      # DropPath masks
            for _ in range(2):
                loss = train_step(model, MixLoss(0.1), opt, batch, clip_grad_norm=5.0)
        finally:
            VF._SIDE_ENABLED = old
        torch.cuda.synchronize()
        return loss, [p.detach().clone() for p in model.parameters()]

    la, pa = run(True)
    lb, pb = run(False)
    assert torch.equal(la, lb)
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.equal(a, b), f"parameter {i} differs between the side-stream and the single-stream step"


# ------------------------------------------------------------------ one C call per layer (csrc/layer.hip)
def _layer_io(model, x, bf16, seed, side=True):
    """logits-free probe of a whole model: output + every parameter gradient of one forward / backward."""
    from vtx import functional as VF
    model.zero_grad(set_to_none=True)
    torch.manual_seed(seed)                                  # the DropPath draws
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        out = model(x)
    with VF.deferred_wgrad(side):        # (side stream: only where every parameter gets ONE gradient per backward)
        out.float().square().mean().backward()
    return out.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("bf16", [True, False])
@pytest.mark.parametrize("family", ["swin", "vit", "vit_multicrop"])
def test_one_call_per_layer_is_bitwise_the_call_by_call_path(family, bf16, monkeypatch):
    """vtx_layer_fwd / vtx_layer_bwd enqueue the launches of functional.TransformerLayerFn from one descriptor (window
    attention with / without shift and DropPath, global attention; weight gradients on the side stream): outputs and all
    parameter gradients must equal the call-by-call path bit for bit, in bf16 and in the fp32 parity mode."""
    from vtx import functional as VF
    d = dev()
    torch.manual_seed(31)
    if family == "swin":
        from models import SwinTransformer
        model = SwinTransformer(image_size=(224, 224), n_class=16, depths=(2, 2, 2, 2), dims=(64, 128, 256, 512), dim_head=32,
                                n_heads=(2, 4, 8, 16), dim_ffs=(256, 512, 1024, 2048), window_size=7, drop_path=0.2)
        for m in model.modules():
            if hasattr(m, "rel_pos"):
                torch.nn.init.normal_(m.rel_pos.weight, std=0.3)
        x = torch.randn(6, 3, 224, 224, device=d)
    else:
        from models import VisionTransformer
        from vtx.nn import Linear
        model = VisionTransformer(Linear(128, 16), 224, 16, 3, 128, 2, 512, 0.0, 0.0, 0.0, 0.1)
        x = torch.randn(5, 3, 224, 224, device=d)
        if family == "vit_multicrop":
            x = [x, torch.randn(5, 3, 96, 96, device=d), torch.randn(5, 3, 96, 96, device=d)]
    model.to(d).train()
    calls = []
    real = VF.TransformerLayerFn._forward_one_call
    monkeypatch.setattr(VF.TransformerLayerFn, "_forward_one_call", staticmethod(lambda *a, **k: (calls.append(1), real(*a, **k))[1]))
    monkeypatch.setattr(VF, "_LAYER_CALL", True)
    monkeypatch.setattr(VF, "_layer_perms", lambda *a: None)     # (stochastic-depth compaction has its own test below)
    side = family != "vit_multicrop"
    out_a, g_a = _layer_io(model, x, bf16, 77, side)
    assert len(calls) >= (3 if bf16 else 0), "the one-call path did not run"     # (fp32: grouped weight gradients are bf16-only)
    n = len(calls)
    monkeypatch.setattr(VF, "_LAYER_CALL", False)
    out_b, g_b = _layer_io(model, x, bf16, 77, side)
    assert len(calls) == n, "VTX_LAYER_CALL = 0 must take the call-by-call path"
    assert torch.equal(out_a, out_b)
    assert g_a.keys() == g_b.keys() and len(g_a) > 20
    for k in g_a:
        assert torch.equal(g_a[k], g_b[k]), f"{family}: gradient of {k} differs between the one-call and the call-by-call layer"


def test_one_call_layer_under_no_grad_and_shared_param_backward():
    """The one-call layer without a graph (teacher / evaluation: no pre-activation kept) gives the training forward's bits,
    and inside shared_param_backward() (DINO) its second gradient is accumulated in the reduce launch -- bitwise autograd's sum."""
    from models import VisionTransformer
    from vtx import functional as VF
    from vtx.nn import Linear
    d = dev()
    torch.manual_seed(33)
    model = VisionTransformer(Linear(384, 16), 224, 16, 2, 384, 6, 1536, 0.0, 0.0, 0.0, 0.0).to(d).train()
    crops = [torch.randn(16, 3, 224, 224, device=d), torch.randn(32, 3, 96, 96, device=d)]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ref = model(crops)
        with torch.no_grad():
            ng = model(crops)
    assert torch.equal(ref, ng)

    def grads(shared):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = model(crops).float().square().mean()
        with VF.shared_param_backward(shared):
            loss.backward()
            if shared:
                assert len(VF._shared_grads) == 2 * 12
        return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    a, b = grads(False), grads(True)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("family,p", [("swin", 0.45), ("swin", 0.1), ("vit", 0.4), ("vit_multicrop", 0.3), ("swin_astat", 0.45), ("swin_fwd4", 0.45),
                                      ("swin_wide", 0.45)])
def test_stochastic_depth_compaction_matches_the_compute_and_scale_path(family, p, monkeypatch, request):
    """Round 3: with host-drawn DropPath masks every branch of a Swin layer runs over its KEPT samples only (row-mapped
    LayerNorm / LDS-DMA GEMMs / window attention, copy-only tiles for the dropped samples, weight gradients skipping their
    rows: csrc/layer.hip) instead of being computed for all samples and multiplied by 0.  Same masks, same per-row math:
    the output (a per-row quantity) must be BIT-identical to the compute-and-scale path; LayerNorm gamma / beta and
    rel_pos gradients are summed over a different partition of the rows
    (1e-5).  Also: some branch must actually have been compacted, and nothing may be NaN although the dropped samples'
    activations are never written."""
    from models import SwinTransformer, VisionTransformer
    from vtx import functional as VF
    from vtx.nn import Linear
    d = dev()
    torch.manual_seed(41)
    nb = 10
    if family == "swin_astat":
        # (round 4) batch 32: stage 3 has 49 whole 128-row strips; GEMM_ASTAT = 2 sends its K = 384 GEMMs to the A-stationary kernel at
        # any row count -- row-mapped (sample table in LDS, copy-only rows by the request waves) in the compacted run, with
        # DropPath scales and every row computed in the call-by-call run
        from vtx import options
        prev = options.get("GEMM_ASTAT")
        request.addfinalizer(lambda: options.set("GEMM_ASTAT", prev))
        options.set("GEMM_ASTAT", 2)
        family, nb = "swin", 32
    fwd4 = family == "swin_fwd4"
    if fwd4:
        # (round 5) the four-wave window-attention forward at any problem count (the benchmark reaches it at stage 1 only): row-mapped
        # in the compacted run; afterwards the one-wave kernel must give the same bits
        from vtx import options
        prev4 = options.get("WATTN_FWD4")
        request.addfinalizer(lambda: options.set("WATTN_FWD4", prev4))
        options.set("WATTN_FWD4", 2)
        family = "swin"
    dims, heads, ffs = (64, 128, 384, 768), (2, 4, 12, 24), (256, 512, 1536, 3072)
    if family == "swin_wide":
        # (round 5) widths 256 / 640: the compacted layers' grouped weight gradients run the ROW-MAPPED wide-tile kernel on 128 x 256
        # (opt-in: WGRAD_WIDE=9) / 128 x 320 tiles, the call-by-call run the plain one through the DropPath masks
        from vtx import ops, options
        prevw = options.get("WGRAD_WIDE")
        request.addfinalizer(lambda: options.set("WGRAD_WIDE", prevw))
        options.set("WGRAD_WIDE", 9)
        assert ops.wgrad_wide_tiles([(256, 1024), (1024, 256), (256, 256), (768, 256)], want_j=True)[1] == 4
        assert ops.wgrad_wide_tiles([(640, 2560), (2560, 640), (640, 640), (1920, 640)], want_j=True)[1] == 5
        dims, heads, ffs = (64, 128, 256, 640), (2, 4, 8, 20), (256, 512, 1024, 2560)
        family = "swin"
    if family == "swin":
        model = SwinTransformer(image_size=(224, 224), n_class=16, depths=(1, 1, 3, 2), dims=dims, dim_head=32,
                                n_heads=heads, dim_ffs=ffs, window_size=7, drop_path=p).to(d).train()
        for m in model.modules():
            if hasattr(m, "rel_pos"):
                torch.nn.init.normal_(m.rel_pos.weight, std=0.3)
        x = torch.randn(nb, 3, 224, 224, device=d)
    else:                                                    # global attention (the bf16 fast path takes the sample order too)
        model = VisionTransformer(Linear(384, 16), 224, 16, 4, 384, 6, 1536, 0.0, 0.0, 0.0, p).to(d).train()
        x = torch.randn(10, 3, 224, 224, device=d)
        if family == "vit_multicrop":
            x = [x, torch.randn(10, 3, 96, 96, device=d)]
    side = family != "vit_multicrop"
    used = []
    real = VF._layer_perms
    monkeypatch.setattr(VF, "_layer_perms", lambda *a: (used.append(real(*a)), used[-1])[1])
    monkeypatch.setattr(VF, "_LAYER_CALL", True)
    monkeypatch.setattr(VF, "_COMPACT_MIN_PCT", 0)             # (compact whenever anything is dropped: the mechanism is under test)
    model._vtx_dp_compaction = True
    # poison the allocator's free memory: what compaction leaves unwritten must never be read
    junk = torch.full((1 << 28,), float("nan"), device=d, dtype=torch.bfloat16)
    del junk
    out_a, g_a = _layer_io(model, x, True, 91, side)
    assert sum(u is not None and (u[0][1] < nb or u[1][1] < nb) for u in used) >= 2, "no branch was compacted"
    monkeypatch.setattr(VF, "_LAYER_CALL", False)                      # call-by-call: every sample computed, then scaled
    out_b, g_b = _layer_io(model, x, True, 91, side)
    assert torch.isfinite(out_a).all() and all(torch.isfinite(v).all() for v in g_a.values())
    assert torch.equal(out_a, out_b)
    exact = 0
    for k in g_a:
        if torch.equal(g_a[k], g_b[k]):
            exact += 1
        else:
            check(f"compaction: d {k}", g_a[k], g_b[k], 1e-5)
    assert exact >= 10, "the stages without compaction (and the stem / head) must still agree bit for bit"
    if nb == 32:
        from vtx import options
        options.set("GEMM_ASTAT", 0)                                     # ... and the tiled kernels give the same bits
        out_c, _ = _layer_io(model, x, True, 91, side)
        assert torch.equal(out_a, out_c)
    if fwd4:
        from vtx import options
        options.set("WATTN_FWD4", 0)
        out_c, g_c = _layer_io(model, x, True, 91, side)
        assert torch.equal(out_b, out_c) and all(torch.equal(g_b[k], g_c[k]) for k in g_b), "four-wave forward != one-wave forward in the model"


def test_forward_feature_without_a_weight_scope_runs_uncompacted_and_backward_works(monkeypatch):
    """`VisionTransformer.forward_feature()` is a public method of the reference (vit.py:139-151).  Called directly
    there is no weight_scope, hence no transposed bf16 weight copies, which the compacted backward needs: the layers must run
    uncompacted (same numbers as forward(): compaction never changes a per-row result) instead of failing in backward."""
    from models import VisionTransformer
    from vtx import functional as VF
    d = dev()
    torch.manual_seed(3)
    model = VisionTransformer(None, 224, 16, 3, 384, 6, 1536, 0.0, 0.0, 0.0, 0.45).to(d).train()
    assert model._vtx_dp_compaction
    monkeypatch.setattr(VF, "_COMPACT_MIN_PCT", 0)
    monkeypatch.setattr(VF, "_LAYER_CALL", True)
    x = torch.randn(10, 3, 224, 224, device=d)

    def run(direct):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(17)                                  # same DropPath draws
        with torch.autocast("cuda", dtype=torch.bfloat16):
            f = model.forward_feature(x) if direct else model(x)
        f.float().square().sum().backward()
        return f.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()}

    f_direct, g_direct = run(True)                             # raised VtxError("vtx_layer_bwd: ... shape") in backward before
    f_model, g_model = run(False)
    assert torch.isfinite(f_direct).all() and all(torch.isfinite(v).all() for v in g_direct.values())
    assert torch.equal(f_direct, f_model)
    for k in g_direct:
        check(f"forward_feature vs forward: d {k}", g_direct[k], g_model[k], 1e-5)


# ------------------------------------------------------------------ option-off vs option-on of two round-4 switches
@pytest.mark.parametrize("B,H,W,C,r", [(2, 56, 56, 64, 7), (2, 28, 28, 128, 7), (1, 14, 14, 256, 7)])
def test_twins_subsampling_through_lds_is_bitwise_the_elementwise_kernel(B, H, W, C, r):
    """TWINS_SUB_LDS = 1 (gather / scatter staged through LDS where the geometry gives 8- / 16-byte chunks) vs 0 (element-wise): a
    permutation and its inverse (+ accumulate) -- identical bits."""
    from vtx import ops, options
    d = dev()
    x = _mk((B, H, W, C), 701, BF).to(d)
    g = _mk((B * (H // r) * (W // r), C * r * r), 702, BF).to(d)
    base = _mk((B, H, W, C), 703, BF).to(d)
    res = {}
    for v in (1, 0):
        with options.override(TWINS_SUB_LDS=v):
            f = ops.twins_subsample_fwd(x, B, H, W, C, r)
            dx = torch.empty_like(x)
            ops.twins_subsample_bwd(g, dx, B, H, W, C, r)
            acc = base.clone()
            ops.twins_subsample_bwd(g, acc, B, H, W, C, r, accumulate=True)
            torch.cuda.synchronize()
            res[v] = (f, dx, acc)
    for a, b in zip(res[1], res[0]):
        assert torch.equal(a, b), "TWINS_SUB_LDS changes bits"


def test_split_k_dgrad_of_the_dino_output_layer_vs_the_plain_gemm_and_fp64(monkeypatch):
    """VTX_DGRAD_SPLITK (functional.dgrad: a 65 536-long contraction with 20 output tiles runs as a split-K launch): against the
    plain GEMM path (another fixed summation partition) and against fp64."""
    from vtx import functional as VF
    d = dev()
    gen = torch.Generator().manual_seed(704)
    dy = torch.randn(640, 65536, generator=gen).to(BF)
    w = (0.02 * torch.randn(65536, 256, generator=gen))
    wp = (w.to(BF).to(d), None)
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(VF, "_DGRAD_SPLITK", flag)
        out[flag] = VF.dgrad(dy.to(d), wp, BF)
        again = VF.dgrad(dy.to(d), wp, BF)
        assert torch.equal(out[flag], again), "dgrad is not deterministic"
    ref = dy.double() @ w.to(BF).double()
    check("DINO output-layer dgrad, split-K vs fp64", out[True], ref, 4e-3)
    check("DINO output-layer dgrad, plain GEMM vs fp64", out[False], ref, 4e-3)
    check("DINO output-layer dgrad, split-K vs plain GEMM", out[True], out[False].double(), 6e-3)
