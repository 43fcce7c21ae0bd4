"""BASELINE.json configs[2] ("512 x 448^2 frames sharded 8 ways ... all-gather of scores") at its OWN workload, on one MI355X.

The reference has no intra-video sharding (/root/reference/lmms_eval/models/videoitg.py:154-177 loads one whole model per rank and
lmms_eval/utils.py:689-695 strides whole documents over ranks), so the correctness of the 8-way split is this repo's alone to prove:
N processes share cuda:0, exchange over gloo (RCCL refuses two ranks on one device) and run `ContextParallel.forward` on the FULL model
(26 + 28 layers, real widths) over their frame shard of the headline video.  The gathered logits must meet the complete fp32 oracle pass
(tests/golden/full_c2_512x448.json, north_star's 1e-3 on the scores and the identical Top-32 set) and the single-GPU forward.  This is
where the kernels only a rank launches meet the oracle: M = 1600 / 1660 rows on the 256-wide tiles with a ragged last row tile, split-K
o / down, K|V-first and Q-only projections, 1 6xx queries x 12 860 gathered keys, 64-frame tower shards pooled with the GLOBAL grid.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
CASE = "c2_512x448"
SCORE_TOL = 1e-3            # north_star
LOGIT_TOL_VS_SINGLE = 4e-3  # sharding changes tile shapes / split-K partitions, i.e. fp32 summation order, nothing else
FP8_POLICY = "llm.gu,llm.down"
FP8_SCORE_TOL = 2.5e-2      # the decoder-MLP fp8 tier's own bound (profiles/r03_fp8_pareto.md), as in tests/test_gpu_fp8.py


def _mod():
    if TOOLS not in sys.path:
        sys.path.insert(0, TOOLS)               # spawn hands sys.path to the children, which import the module by name
    import cp_full_one_gpu as mod
    return mod


_SINGLE = {}


def single_gpu(case):
    """the one-GPU forward of a fixture's video (computed once per case)"""
    if case not in _SINGLE:
        torch.cuda.empty_cache()
        _SINGLE[case] = _mod().single_gpu_logits(case)
    return _SINGLE[case]


# the headline video at 8 / 4 / 2 ranks; the reference-native 384^2 shape (27 x 27 grid, no position-table interpolation) at 8 ranks; the
# LLM-heavy 64-frame video (16 x 16 pooled grid, N = 16 444: 4 096 + rows per rank) at 4 ranks
# (round 6, VERDICT r5 item 6 - the GPU suite's budget: the 4- and 2-rank runs of the headline video went; 4 ranks are covered by c1, 2 / 3 / 4 ranks by the
# toy splits of tests/test_gpu_e2e.py::test_context_parallel_multi_rank_on_one_gpu)
@pytest.mark.parametrize("world,case", [(8, CASE), (8, "b0_512x384"), (4, "c1_64x384")])
def test_context_parallel_full_size_meets_oracle(world, case):
    mod = _mod()
    rec = mod.load_case(case)
    ref1 = single_gpu(case)
    torch.cuda.empty_cache()
    got, forms = mod.run(world, case, port=29560 + world + (0 if case == CASE else 20 if case.startswith("b0") else 40))
    rep = mod.report(case, got, ref1)
    print(f"context-parallel x{world} on {case}, full model: {rep}; rows per rank {forms['n_local']}; split-K shapes {forms['splitk']}")
    assert got.shape == (rec["frames"],) and torch.isfinite(got).all()
    assert rep["max_abs_dscore_vs_oracle"] <= SCORE_TOL, rep
    gap = rec["gap_k_k1"]
    # the headline video: Top-32 set identical (holds because the error is 3e-4 against a rank-32/33 gap of 1.56e-3); the other
    # fixtures: identical wherever the oracle's own rank-K / K+1 gap exceeds twice the tolerance, else at most one frame swapped at the boundary
    # round 6: decidable = gap above twice the fp16 path's stated error ceiling on benign draws (4e-4, tests/test_fullsize_parity.py
    # CEIL) or this run's own error - so the reference-native 512 x 384^2 video (gap 1.35e-3) has its Top-32 set asserted at 8 ranks too
    if case == CASE or gap is None or gap > 2 * max(4e-4, rep["max_abs_dscore_vs_oracle"]):
        assert rep[f"top{rec['k']}_set_match"], rep
    else:
        assert rep[f"top{rec['k']}_kept"] >= rec["k"] - 1, rep
    assert rep["max_abs_dlogit_vs_single_gpu"] <= LOGIT_TOL_VS_SINGLE, rep
    if world == 8 and case == CASE:
        # the few-tile forms really ran on rank 0 (M = 1600 rows): o and down projections on split-K
        ks = {s[0][2] for s in forms["splitk"]}
        assert 3584 in ks and 18944 in ks, forms
        assert forms["n_local"] == [1600] * 7 + [1660]


def test_context_parallel_on_heavy_tailed_weights():
    """round 5: the sharded forward on the heavy-tailed twin of the headline video (tests/golden/full_c2o_512x448.json: massive-activation channels, |x| ~ 10^3
    in both residual streams).  Sharding only re-partitions fp32 sums (tile shapes, split-K), but on these weights every such re-association is amplified like the
    16-bit operand noise is (35x in logits), so the gathered logits are held to 1e-2 of the ONE-GPU forward (measured 2.6e-3; the benign bound is 4e-3) and to the fixture's stated
    bounds against the oracle (tests/test_fullsize_parity.py: HEAVY)."""
    mod = _mod()
    case = "c2o_512x448"
    rec = mod.load_case(case)
    ref1 = single_gpu(case)
    torch.cuda.empty_cache()
    got, forms = mod.run(4, case, port=29575)             # (4 ranks since round 6: what is tested is the amplified re-association, not the rank count)
    rep = mod.report(case, got, ref1)
    print(f"context-parallel x4 on {case} (heavy-tailed weights): {rep}")
    assert got.shape == (rec["frames"],) and torch.isfinite(got).all()
    assert rep["max_abs_dscore_vs_oracle"] <= 5e-3 and rep["max_abs_dlogit_vs_oracle"] <= 1.2e-1, rep
    assert rep[f"top{rec['k']}_kept"] >= rec["k"] - 1, rep
    assert rep["max_abs_dlogit_vs_single_gpu"] <= 1e-2, rep


def test_context_parallel_operand_split_tier_on_the_heavy_tailed_headline_video():
    """round 6: the frame-sharded / context-parallel forward in the OPERAND-SPLIT tier (K|V exchanged as (hi, lo) pairs, 3-term attention against the
    gathered keys) at full size on the heavy-tailed twin of the headline video: two ranks on one GPU against the complete fp32 oracle pass at the tier's own
    bounds (tests/test_gpu_split.py: 1e-4 in score, the oracle's Top-32 although its rank-32/33 gap is 1.5e-4)."""
    mod = _mod()
    case = "c2o_512x448"
    rec = mod.load_case(case)
    torch.cuda.empty_cache()
    got, forms = mod.run(2, case, port=29577, split=True)
    rep = mod.report(case, got)
    print(f"operand-split tier, context-parallel x2 on {case}: {rep}")
    assert got.shape == (rec["frames"],) and torch.isfinite(got).all()
    assert rep["max_abs_dscore_vs_oracle"] <= 1e-4 and rep["max_abs_dlogit_vs_oracle"] <= 4e-3, rep
    assert rep[f"top{rec['k']}_set_match"], rep


def test_context_parallel_full_size_fp8_llm_mlp():
    """configs[4] names the fp8 weight path at 1 AND 8 GPUs: the named fp8 tier (decoder gate/up + down on e4m3, MX hand-over) through the
    8-way context-parallel forward against the oracle at the tier's own bound, and against the same tier on one GPU."""
    mod = _mod()
    torch.cuda.empty_cache()
    got, forms = mod.run(8, CASE, fp8=FP8_POLICY, port=29571)
    single = mod.single_gpu_logits(CASE, fp8=FP8_POLICY)      # (the fp8 tier's own one-GPU forward)
    rep = mod.report(CASE, got, single)
    print(f"context-parallel x8 fp8 {FP8_POLICY}: {rep}")
    assert torch.isfinite(got).all()
    assert rep["max_abs_dscore_vs_oracle"] <= FP8_SCORE_TOL, rep
    # row sharding does not change a row's quantisation (per-token / per-block scales are row-local); what differs is the fp32 summation
    # order in front of a quantiser, which can move an e4m3 code: the bound is the tier's own
    assert rep["max_abs_dscore_vs_single_gpu"] <= FP8_SCORE_TOL, rep


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` as a plain command (no launcher, WORLD_SIZE unset) must start its two ranks itself - the reference's
    equivalent is `accelerate launch --num_processes=8` (/root/reference/scripts/eval_lmms_eval/videomme_grounding.sh:3-5) - and print ONE
    JSON line saying how many ranks took part in the collectives and over which backend.  One-GPU dry run: both ranks on cuda:0 over gloo."""
    import json
    import subprocess
    root = os.path.dirname(TOOLS)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VITG_BENCH_SINGLE_DEVICE="1", VITG_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "tiny", "--frames", "9", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-secondary"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["backend"] == "gloo" and rec["scaling"] == "strong"
    assert rec["value"] > 0 and len(rec["top32"]) == 9
