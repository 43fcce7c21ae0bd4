"""Complete-pass parity at full depth and width (26 SigLIP + 28 Qwen2 layers, the real hidden sizes).

tests/golden/full_<case>.json hold the logits of a complete fp32 ORACLE pass (tests/golden/make_fullsize_golden.py, run in the
build container; the oracle itself is pinned to the reference at small sizes by test_oracle_golden.py).  Weights are
`synth.counter_tensor` - an integer hash that is bit-identical on CPU and GPU - so the GPU rebuilds exactly the model the
oracle saw; no oracle code runs here.

Gate (north_star): max |sigmoid score - oracle| <= 1e-3 and an identical Top-K index set for the HEADLINE dtype fp16 (the
reference's own inference type, eagle/model/builder.py:59).  bf16 is the secondary tier: 8 instead of 11 mantissa bits on every
MFMA operand cost it 1.4e-3 at 512 x 448^2 and 1.7e-3 at 512 x 384^2 (measured, profiles/r02_parity.md), so it is held to a STATED 2.5e-3.  Both bounds
are fixed numbers - neither comes from a model of this implementation's own rounding.
"""
import json
import os

import pytest
import torch

SCORE_TOL = 1e-3
TOL = {torch.float16: SCORE_TOL, torch.bfloat16: 2.5e-3}
CASES = ["depth16_224", "depth16_224_s1", "depth24_280_s2", "c1_64x384", "b0_512x384", "c2_512x448",
         # planted per-frame signal (synth.planted_inputs): the oracle's Top-k is the planted set behind a wide gap
         "depth16p_224", "c2p_512x448"]


def load_case(golden_dir, name):
    path = os.path.join(golden_dir, f"full_{name}.json")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    with open(path) as f:
        return json.load(f)


def test_fixtures_are_wellformed(golden_dir):
    """CPU: every committed full-size fixture is self-consistent (order = stable descending sort of its own scores ...)."""
    seen = 0
    for name in CASES:
        path = os.path.join(golden_dir, f"full_{name}.json")
        if not os.path.exists(path):
            continue
        seen += 1
        r = json.load(open(path))
        lg = torch.tensor(r["logits"])
        assert lg.numel() == r["frames"] and torch.isfinite(lg).all()
        s = torch.sigmoid(lg)
        order = torch.sort(s, descending=True, stable=True).indices.tolist()
        assert order == r["order"]
        assert sorted(order[:r["k"]]) == r["topk"]
        if r["gap_k_k1"] is not None:
            assert abs((s[order[r["k"] - 1]] - s[order[r["k"]]]).item() - r["gap_k_k1"]) < 1e-7
        assert r["oracle"]["seconds"]["total"] > 0
        if r.get("plant"):
            # the planted frames ARE the oracle's Top-k, and the gap to the first unplanted frame is wide in logits (the point of the case)
            from videoitg_amd import synth
            assert r["plant"]["frames"] == synth.planted_frames(r["frames"], r["plant"]["k"], r["plant"]["seed"])
            assert r["topk"] == sorted(r["plant"]["frames"]) and r["k"] == r["plant"]["k"]
            assert (lg[order[r["k"] - 1]] - lg[order[r["k"]]]).item() > 1.0
    assert seen >= 1


def gpu_logits(rec, dtype):
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig
    from videoitg_amd.engine import SelectorEngine
    from videoitg_amd.synth_device import packed_random_weights
    cfg = ITGConfig()
    dev = torch.device("cuda:0")
    eng = SelectorEngine(packed_random_weights(cfg, dtype, dev, seed=rec["weight_seed"]))
    ids, pix = synth.fixture_inputs(cfg, rec)
    got = eng.forward(ids, None, [pix.to(dtype).to(dev)])[0].view(-1)
    torch.cuda.synchronize()
    out = got.float().cpu()
    eng.release_workspace()
    del eng
    torch.cuda.empty_cache()
    return out


def compare(rec, got):
    from videoitg_amd.torch_ops import K
    want = torch.tensor(rec["logits"])
    ds = (torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item()
    dl = (got - want).abs().max().item()
    _, order = K.rank(got.cuda())
    order = order.cpu().tolist()
    k = rec["k"]
    return ds, dl, sorted(order[:k]) == rec["topk"], order


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("name", CASES)
def test_full_depth_matches_oracle(golden_dir, name, dtype):
    rec = load_case(golden_dir, name)
    got = gpu_logits(rec, dtype)
    assert torch.isfinite(got).all()
    ds, dl, same_topk, order = compare(rec, got)
    gap = rec["gap_k_k1"]
    print(f"full {name} {dtype}: max|dscore|={ds:.2e} max|dlogit|={dl:.2e} top{rec['k']} set equal: {same_topk} "
          f"(oracle rank-{rec['k']}/{rec['k'] + 1} gap {gap})")
    tol = TOL[dtype]
    assert ds <= tol, (name, dtype, ds)
    # the Top-K SET is decidable only if the oracle's own rank-K/K+1 gap exceeds twice the score tolerance
    if gap is None or gap > 2 * tol:
        assert same_topk, (name, dtype)
    # every pair of frames the oracle separates by more than 2 x tol must be ordered the same way
    want = torch.sigmoid(torch.tensor(rec["logits"]))
    pos = {f: i for i, f in enumerate(order)}
    ro = rec["order"]
    for a in range(len(ro) - 1):
        for b in range(a + 1, len(ro)):
            if (want[ro[a]] - want[ro[b]]).item() > 2 * tol:
                assert pos[ro[a]] < pos[ro[b]], (ro[a], ro[b])
                break                                     # the nearest separable successor suffices (order is transitive)


@pytest.mark.gpu
def test_counter_weights_identical_on_cpu_and_gpu():
    """the premise of the full-size fixtures: synth.counter_tensor gives the same bits on both devices"""
    from videoitg_amd import synth
    for key, shape, mean, std in [("model.layers.3.mlp.down_proj.weight", (3584, 1024), 0.0, 18944 ** -0.5),
                                  ("model.norm.weight", (3584,), 1.0, 0.02),
                                  ("vision_model.embeddings.position_embedding.weight", (729, 1152), 0.0, 0.02)]:
        a = synth.counter_tensor(0, key, shape, mean, std, "cpu")
        b = synth.counter_tensor(0, key, shape, mean, std, "cuda:0").cpu()
        assert torch.equal(a, b), key
