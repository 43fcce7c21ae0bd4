"""CPU: host-side logic added in round 2 - prompt buckets / graph layout keys, the split-K plan, the oracle-side fp8 quantiser,
the torch.ops registration (schemas, Meta implementations, no CPU kernels)."""
import pytest
import torch


def test_prompt_bucket_and_layout_key():
    from videoitg_amd.engine import GraphedForward, prompt_bucket
    assert [prompt_bucket(n, 32) for n in (0, 1, 32, 33, 64, 65)] == [32, 32, 32, 64, 64, 96]
    pix = torch.zeros(8, 3, 56, 56)
    key = lambda ids: GraphedForward.layout_key(None, torch.tensor(ids), pix, 32)
    assert key([-200] + [5] * 7) == key([-200] + [9] * 32) == (0, 32, (8, 3, 56, 56))        # same bucket, token VALUES do not matter
    assert key([-200] + [5] * 33)[1] == 64
    assert key([1, 2, -200, 5])[0] == 2                                                      # text before <image> is part of the key
    with pytest.raises(ValueError):
        key([5, 6, 7])
    with pytest.raises(ValueError):
        key([-200, 5, -200])


def test_splitk_plan_only_for_few_tiles_and_long_k():
    from videoitg_amd import ops
    # no GPU here: the plan falls back to 256 CUs
    # round 5: the few-row shapes of a context-parallel rank plan with 320-row tiles - 84 (70) tiles x 3 partials = one round of 252 (210) blocks each
    # running a third of K, where 98 tiles of 256 rows x 2 ran half of it
    assert ops.gemm_splitk_plan(1660, 3584, 18944) == (3, 3 * 1660 * 3584 * 4)     # down projection of one of 8 ranks
    assert ops.gemm_splitk_plan(1660, 3584, 3584)[0] == 3                          # o projection
    assert ops.gemm_splitk_plan(1600, 3584, 18944)[0] == 3                         # the seven ranks without the prompt
    assert ops.gemm_splitk_plan(3260, 3584, 18944)[0] == 1                         # four ranks: 13 x 14 tiles of 256 rows already fill 182 of 256 CUs
    assert ops.gemm_splitk_plan(12860, 3584, 18944)[0] == 1                        # the single-GPU shape fills the chip
    assert ops.gemm_splitk_plan(1660, 3584, 1152)[0] == 1                          # short K: not worth a reduction pass
    assert ops.gemm_splitk_plan(100, 3584, 18944)[0] == 1                          # below one tile row: the 128-tile kernel's case
    assert ops.gemm_splitk_plan(512, 512, 65536)[0] == 4                           # capped at 4 partials


def test_q8_rows_is_ocp_e4m3_with_per_row_scale():
    from oracle.quant_model import q8_rows
    x = torch.tensor([[0.0, 1.0, -448.0, 3.3, 1e-3], [0.0, 0.0, 0.0, 0.0, 0.0], [1e-6, 2e-6, -3e-6, 0.0, 5e-7]])
    q = q8_rows(x)
    assert q[0, 2] == -448.0 and q[0, 1] == 1.0 and q[0, 0] == 0.0                 # scale 1: the grid is e4m3 itself
    assert q[0, 3] in (3.25, 3.5) and abs(q[0, 3] - 3.3) <= 0.125                   # 3 mantissa bits
    assert (q[1] == 0).all()                                                        # all-zero row: scale 1, stays zero
    assert torch.allclose(q[2], x[2], rtol=2 ** -4, atol=0)                          # tiny rows are scaled up first: relative error only
    big = torch.randn(64, 512) * torch.logspace(-3, 3, 64)[:, None]
    assert ((q8_rows(big) - big).abs() <= big.abs() * 2 ** -4 + big.abs().amax(-1, keepdim=True) * 2 ** -9 * 1.01).all()


def test_torch_ops_registered_with_schemas_and_no_cpu_kernels():
    from videoitg_amd import torch_ops
    for name in torch_ops.OP_NAMES:
        op = getattr(torch.ops.videoitg, name)
        assert op.default._schema.name == f"videoitg::{name}"
    assert "Tensor(a!) out" in str(torch.ops.videoitg.gemm.default._schema)
    assert "Tensor? kv_len=None" in str(torch.ops.videoitg.attention.default._schema)
    x = torch.zeros(4, 64)
    with pytest.raises(NotImplementedError):
        torch.ops.videoitg.rmsnorm(x, torch.ones(64), torch.zeros(4, 64, dtype=torch.float16), 64, 1e-6)
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        q = torch.ops.videoitg.patchify(torch.empty(2, 3, 56, 56, device="cuda"), 14, 640, 1)
        assert q.shape == (32, 640) and q.dtype == torch.float16


def test_context_parallel_local_frames_matches_the_plan():
    from videoitg_amd.parallel import ContextParallel, make_plan
    for world in (1, 2, 3, 8):
        for T in (5, 16, 19, 512):
            plan = make_plan(world, T, 25, 0, 60)
            assert [ContextParallel(r, world).local_frames(T) for r in range(world)] == plan.frames
            assert sum(plan.n_local) == plan.total == T * 25 + 60


def test_workspace_buffers_are_name_keyed_views():
    """engine._buf without a GPU: one allocation per name, grown on demand, [:rows] views share storage"""
    from videoitg_amd.engine import SelectorEngine

    class W:
        cfg = None; dtype = torch.float16; vit_dtype = torch.float16; device = torch.device("cpu"); fp8 = False
    eng = SelectorEngine(W())
    a = eng._buf("x", (10, 8), torch.float32, zero=True)
    b = eng._buf("x", (4, 8), torch.float32)
    assert b.data_ptr() == a.data_ptr() and b.shape == (4, 8) and eng.workspace_bytes() == 10 * 8 * 4
    c = eng._buf("x", (20, 8), torch.float32)
    assert c.shape == (20, 8) and eng.workspace_bytes() == 20 * 8 * 4 and len(eng.workspace_tensors()) == 1
    d = eng._buf("s", (7,), torch.float32)
    assert d.shape == (7,) and len(eng.workspace_tensors()) == 2
    # the fp8 selector: parts, single GEMMs, everything (PackedWeights.fp8_on / fp8_part decide per layer GEMM)
    from videoitg_amd.engine import FP8_GEMMS, PackedWeights, normalise_fp8
    assert normalise_fp8("vit") == frozenset(g for g in FP8_GEMMS if g.startswith("vit."))
    assert normalise_fp8("vit,llm") is True and normalise_fp8(list(FP8_GEMMS)) is True
    pw = PackedWeights(cfg=None, dtype=torch.float16, device=torch.device("cpu"), fp8="vit.fc1,llm.gu,llm.down")
    assert pw.fp8_on("vit.fc1") and not pw.fp8_on("vit.fc2") and pw.fp8_part("vit") and pw.fp8_part("llm") and not pw.fp8_on("llm.o")
    assert not PackedWeights(cfg=None, dtype=torch.float16, device=torch.device("cpu"), fp8="llm").fp8_part("vit")
