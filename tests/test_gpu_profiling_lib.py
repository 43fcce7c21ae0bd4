"""-m gpu, ONLY with VITG_PROFILING_LIB=1 and libvideoitg_hip_prof.so built (videoitg_amd/csrc/build.sh --profiling): the research
GEMM variants of round 1 (fragment-packed weights, persistent kernel with deferred epilogue) stay bit-identical to the shipped
kernels.  Not part of the product test matrix: the product library does not contain these entry points."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.research]      # (tests/conftest.py deselects the module unless VITG_PROFILING_LIB=1)

DTYPES = [torch.bfloat16, torch.float16]
EPS16 = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


@pytest.fixture(scope="module")
def ops():
    from videoitg_amd import ops as o
    return o


def rnd(shape, seed, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).float()


# ---- static-weight fast path: fragment-packed weights (gemm_wf.hip) ---------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_pack_weight_fragments_layout(ops, dtype):
    """block (tile, kstep), chunk l = W[tile*32 + (l & 31)][kstep*16 + 8*(l >> 5) : +8]; rows past N are zero; N rounds up to 64"""
    N, K = 200, 96
    w = rnd((N, K), 21, dtype=dtype).to(dtype)
    got = ops.pack_weight_fragments(w.cuda()).cpu()
    np_ = 256                                                                   # ceil(200 / 64) * 64
    wp = F.pad(w, (0, 0, 0, np_ - N))
    want = wp.view(np_ // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)
    assert got.numel() == want.numel() and torch.equal(got.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(512, 512, 64), (512, 512, 128), (640, 320, 192), (777, 520, 256), (1024, 768, 1152), (300, 1152, 4352),
                                   (2048, 3456, 1152), (200, 512, 128), (512, 200, 128)])
def test_gemm_fragment_packed_weights_store16_bit_identical(ops, dtype, M, N, K):
    """same MFMA, same accumulation order as the LDS kernel: identical bits (shapes the fast path does not cover fall back to it)"""
    a, w, b = rnd((M, K), 1, dtype=dtype).to(dtype).cuda(), rnd((N, K), 2, K ** -0.5, dtype=dtype).to(dtype).cuda(), rnd((N,), 3, 0.1, torch.float32).cuda()
    wf = ops.pack_weight_fragments(w)
    ops.set_option("gemm_cfg", 6)                                               # the kernel with the same k-order and wave tile
    ops.set_option("gemm_mi", 32)                                               # (its 32x32x16 form: the 16x16x32 one starts its accumulators at the bias)
    try:
        ref = ops.gemm(a, w, torch.empty((M, N), dtype=dtype, device="cuda"), mode=ops.GEMM_STORE16, bias=b, act=ops.ACT_GELU_TANH)
    finally:
        ops.set_option("gemm_cfg", 0)
        ops.set_option("gemm_mi", 0)
    got = ops.gemm(a, w, torch.full((M, N), float("nan"), dtype=dtype, device="cuda"), mode=ops.GEMM_STORE16, bias=b, act=ops.ACT_GELU_TANH, wf=wf)
    assert torch.equal(got, ref)
    want = F.gelu(a.float() @ w.float().t() + b, approximate="tanh")
    assert (got.float() - want).abs().max().item() <= EPS16[dtype] * want.abs().max().item() + 1e-3


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_fragment_packed_weights_f32_and_swiglu(ops, dtype):
    M, N, K = 1000, 1152, 4352
    a, w, b = rnd((M, K), 4, dtype=dtype).to(dtype).cuda(), rnd((N, K), 5, K ** -0.5, dtype=dtype).to(dtype).cuda(), rnd((N,), 6, 0.1, torch.float32).cuda()
    x = rnd((M, N), 8, 1.0, torch.float32).cuda()
    want = a.float() @ w.float().t() + b + x
    out = x.clone()
    ops.gemm(a, w, out, mode=ops.GEMM_F32, bias=b, res=out, wf=ops.pack_weight_fragments(w))
    assert (out - want).abs().max().item() < 2e-4 * math.sqrt(K)
    M, F2, K = 1300, 1024, 512
    a, gu = rnd((M, K), 9, dtype=dtype).to(dtype).cuda(), rnd((2 * F2, K), 10, K ** -0.5, dtype=dtype).to(dtype).cuda()
    ref = ops.gemm(a, gu, torch.empty((M, F2), dtype=dtype, device="cuda"), mode=ops.GEMM_SWIGLU)
    got = ops.gemm(a, gu, torch.empty((M, F2), dtype=dtype, device="cuda"), mode=ops.GEMM_SWIGLU, wf=ops.pack_weight_fragments(gu))
    assert (got.float() - ref.float()).abs().max().item() <= EPS16[dtype] * ref.float().abs().max().item()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(512, 512, 768), (4096, 1024, 1152), (70000, 3456, 1152), (12860, 4608, 3584), (5000, 520, 832), (300, 264, 768)])
def test_gemm_persistent_deferred_epilogue_bit_identical(ops, dtype, M, N, K):
    """cfg 13: one block per CU walks many tiles and stores tile i underneath the main loop of tile i+1 - same bits as cfg 6"""
    a, w, b = rnd((M, K), 31, dtype=dtype).to(dtype).cuda(), rnd((N, K), 32, K ** -0.5, dtype=dtype).to(dtype).cuda(), rnd((N,), 33, 0.1, torch.float32).cuda()
    outs = {}
    for cfg in (6, 13):
        ops.set_option("gemm_cfg", cfg)
        ops.set_option("gemm_mi", 32)                                           # (cfg 6 in its 32x32x16 form, the one cfg 13 was derived from: bias added in the epilogue)
        try:
            outs[cfg] = [ops.gemm(a, w, torch.full((M, N), float("nan"), dtype=dtype, device="cuda"), mode=ops.GEMM_STORE16, bias=b, act=act)
                         for act in (ops.ACT_NONE, ops.ACT_GELU_TANH, ops.ACT_GELU_ERF)]
        finally:
            ops.set_option("gemm_cfg", 0)
            ops.set_option("gemm_mi", 0)
    for got, ref in zip(outs[13], outs[6]):
        assert torch.isfinite(got.float()).all()
        assert torch.equal(got, ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,F2,K", [(1300, 1024, 768), (12860, 2048, 3584), (600, 320, 832)])
def test_gemm_persistent_deferred_epilogue_swiglu_bit_identical(ops, dtype, M, F2, K):
    a, gu = rnd((M, K), 41, dtype=dtype).to(dtype).cuda(), rnd((2 * F2, K), 42, K ** -0.5, dtype=dtype).to(dtype).cuda()
    outs = {}
    for cfg in (6, 13):
        ops.set_option("gemm_cfg", cfg)
        try:
            outs[cfg] = ops.gemm(a, gu, torch.full((M, F2), float("nan"), dtype=dtype, device="cuda"), mode=ops.GEMM_SWIGLU)
        finally:
            ops.set_option("gemm_cfg", 0)
    assert torch.isfinite(outs[13].float()).all() and torch.equal(outs[13], outs[6])


# ---- LayerNorm / RMSNorm folded into the GEMMs around it (vitg_gemm_norm_*) --------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,res_mod", [(1024, 1152, 1152, 0), (2048, 1152, 4352, 0), (300, 3584, 3584, 0), (1458, 1152, 640, 729), (700, 256, 256, 0)])
def test_fused_norm_producer(ops, dtype, M, N, K, res_mod):
    """out must equal the plain fp32-residual GEMM bit for bit; xh = round16(out * gamma); finalised statistics = the row mean /
    rstd of out (LayerNorm) and rsqrt(mean(out^2) + eps) (RMSNorm) to fp32 accuracy"""
    cols = ops.gemm_norm_plan(M, N, K)
    assert cols in (64, 96) and N % cols == 0
    a, w = rnd((M, K), 51, dtype=dtype).to(dtype).cuda(), (rnd((N, K), 52, dtype=dtype) * K ** -0.5).to(dtype).cuda()
    bias, gamma = rnd((N,), 53).cuda(), (1 + 0.1 * rnd((N,), 54)).cuda()
    res = (rnd((res_mod or M, N), 55) * 3 + 0.5).cuda()
    want = torch.empty((M, N), device="cuda") if res_mod else res.clone()
    ops.gemm(a, w, want, mode=ops.GEMM_F32, bias=bias, res=res if res_mod else want, res_mod=res_mod)
    out = torch.empty((M, N), device="cuda") if res_mod else res.clone()
    xh = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
    P = N // cols
    part = torch.empty(M * P * 2, device="cuda")
    ops.gemm_norm_producer(a, w, out, res=res if res_mod else out, next_gamma=gamma, xh=xh, partials=part, bias=bias, res_mod=res_mod)
    assert torch.equal(out, want)
    want_xh = (out * gamma)
    assert ((xh.float() - want_xh).abs() <= EPS16[dtype] * want_xh.abs() + 1e-7).all()      # one rounding to 16 bit, nothing more
    for rms in (False, True):
        stats = torch.empty(M * 2, device="cuda")
        ops.norm_stats_finalize(part, M, P, cols, 1e-6, rms, stats)
        st = stats.view(M, 2).double().cpu()
        o = out.double().cpu()
        if rms:
            assert (st[:, 0] == 0).all()
            assert torch.allclose(st[:, 1], torch.rsqrt((o * o).mean(-1) + 1e-6), rtol=2e-6, atol=0)
        else:
            assert torch.allclose(st[:, 0], o.mean(-1), rtol=0, atol=3e-6 * o.abs().max().item())
            assert torch.allclose(st[:, 1], torch.rsqrt(o.var(-1, unbiased=False) + 1e-6), rtol=5e-6, atol=0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,act", [(1024, 3456, 1152, 0), (1000, 4352, 1152, 1), (300, 4608, 3584, 0), (512, 384, 256, 1)])
def test_fused_norm_consumer_layernorm(ops, dtype, M, N, K, act):
    """epilogue(rstd * (xh @ W^T - mean * s) + c) == Linear(LayerNorm(x)) up to the 16-bit rounding of xh = x * gamma"""
    g = torch.Generator().manual_seed(61)
    x = torch.randn((M, K), generator=g) * 2 + 0.3 * torch.randn((M, 1), generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    w = (torch.randn((N, K), generator=g) * K ** -0.5).to(dtype)
    b = 0.1 * torch.randn(N, generator=g)
    xh = (x * gamma).to(dtype)
    mean, rstd = x.mean(-1), torch.rsqrt(x.var(-1, unbiased=False) + 1e-6)
    stats = torch.stack([mean, rstd], 1).contiguous().cuda()
    wf = w.float()
    colsum, c = (wf @ gamma).cuda(), (wf @ beta + b).cuda()
    out = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
    ops.gemm_norm_consumer(xh.cuda(), w.cuda(), out, mode=ops.GEMM_STORE16, stats=stats.view(-1), colsum=colsum, bias=c,
                           act=ops.ACT_GELU_TANH if act else ops.ACT_NONE)
    # reference with the SAME rounded operand (what the kernel is asked to compute), then against the true LayerNorm path
    ref = rstd[:, None] * (xh.float() @ wf.t() - mean[:, None] * colsum.cpu()) + c.cpu()
    true = torch.nn.functional.layer_norm(x, (K,), gamma, beta, 1e-6) @ wf.t() + b
    if act:
        from oracle.videoitg_oracle import gelu_tanh
        ref, true = gelu_tanh(ref), gelu_tanh(true)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= EPS16[dtype] * ref.abs().max().item() + 2e-3
    assert (got - true).abs().max().item() <= 6 * EPS16[dtype] * max(1.0, true.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
def test_fused_norm_consumer_rmsnorm_swiglu(ops, dtype):
    M, F_, K = 600, 1024, 512
    g = torch.Generator().manual_seed(62)
    x = torch.randn((M, K), generator=g) * 3
    gamma = 1 + 0.1 * torch.randn(K, generator=g)
    gate, up = (torch.randn((F_, K), generator=g) * K ** -0.5).to(dtype), (torch.randn((F_, K), generator=g) * K ** -0.5).to(dtype)
    gu = torch.cat([gate.view(F_ // 32, 1, 32, K), up.view(F_ // 32, 1, 32, K)], 1).reshape(2 * F_, K)
    xh = (x * gamma).to(dtype)
    rstd = torch.rsqrt((x * x).mean(-1) + 1e-6)
    stats = torch.stack([torch.zeros(M), rstd], 1).contiguous().cuda()
    out = torch.empty((M, F_), dtype=dtype, device="cuda")
    ops.gemm_norm_consumer(xh.cuda(), gu.cuda(), out, mode=ops.GEMM_SWIGLU, stats=stats.view(-1))
    h = xh.float() * rstd[:, None]
    want = torch.nn.functional.silu(h @ gate.float().t()) * (h @ up.float().t())
    assert (out.float().cpu() - want).abs().max().item() <= 2 * EPS16[dtype] * want.abs().max().item() + 2e-3


@pytest.mark.parametrize("name", ["depth16_224", "c1_64x384"])
def test_fused_norm_forward_matches_unfused_and_oracle(golden_dir, name, monkeypatch):
    """LayerNorm / RMSNorm folded into the neighbouring GEMMs (engine.fuse_norm) at full depth and width: same scores as the
    stand-alone-norm path to well inside the tolerance, and the 1e-3 gate against the complete oracle pass still holds (fp16)."""
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig
    from videoitg_amd.engine import SelectorEngine
    from videoitg_amd.synth_device import packed_random_weights
    import json
    path = os.path.join(golden_dir, f"full_{name}.json")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    rec = json.load(open(path))
    cfg = ITGConfig()
    monkeypatch.setenv("VITG_FUSE_NORM", "1")            # the load-time constants of the fused consumers are only packed when the feature is on
    eng = SelectorEngine(packed_random_weights(cfg, torch.float16, "cuda:0", seed=rec["weight_seed"]))
    assert eng.w.vit[0].qkv_cs is not None
    ids, pix = synth.workload_inputs(cfg, rec["frames"], rec["resolution"], rec["prompt_len"], rec["pixel_seed"])
    pix = pix.half().cuda()
    outs = {}
    for fuse in (False, True):
        eng.fuse_norm = fuse
        outs[fuse] = eng.forward(ids, None, [pix])[0].view(-1).float().cpu()
    assert sum(1 for v in eng._normplan.values() if v > 0) >= 3                          # the fused kernels really ran
    want = torch.tensor(rec["logits"])
    sg = torch.sigmoid
    d_uf = (sg(outs[True]) - sg(outs[False])).abs().max().item()
    d_or = (sg(outs[True]) - sg(want)).abs().max().item()
    print(f"fused norm {name}: max|dscore| vs unfused {d_uf:.2e}, vs oracle {d_or:.2e} (unfused vs oracle {(sg(outs[False]) - sg(want)).abs().max().item():.2e})")
    assert d_uf <= 5e-4 and d_or <= 1e-3
    eng.release_workspace()
