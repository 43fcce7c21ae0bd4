"""-m gpu, ONLY with VITG_PROFILING_LIB=1 and libvideoitg_hip_prof.so built (videoitg_amd/csrc/build.sh --profiling): the research
GEMM variants of round 1 (fragment-packed weights, persistent kernel with deferred epilogue) stay bit-identical to the shipped
kernels.  Not part of the product test matrix: the product library does not contain these entry points."""
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("VITG_PROFILING_LIB", "0") in ("", "0"), reason="profiling library not selected")]

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
    try:
        ref = ops.gemm(a, w, torch.empty((M, N), dtype=dtype, device="cuda"), mode=ops.GEMM_STORE16, bias=b, act=ops.ACT_GELU_TANH)
    finally:
        ops.set_option("gemm_cfg", 0)
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
        try:
            outs[cfg] = [ops.gemm(a, w, torch.full((M, N), float("nan"), dtype=dtype, device="cuda"), mode=ops.GEMM_STORE16, bias=b, act=act)
                         for act in (ops.ACT_NONE, ops.ACT_GELU_TANH, ops.ACT_GELU_ERF)]
        finally:
            ops.set_option("gemm_cfg", 0)
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


