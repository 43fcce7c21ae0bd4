"""Pin the oracle (oracle/videoitg_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py from /root/reference)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import videoitg_oracle as O
from videoitg_amd import synth
from videoitg_amd.config import ITGConfig

CASES = ["pool_post", "nopool_prepost", "oddgrid", "batch2_pad", "chunk520", "ragged_image", "two_images", "two_images_ragged",
         "heavy_pool_post"]        # (round 6: heavy-tailed weights - synth.HeavyTailPlan on the toy architecture - through the reference itself)
TOL = 2e-5


def load_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"{name}.npz"))
    cfg = ITGConfig.from_dict(json.loads(str(z["cfg"])))
    seed = int(z["seed"])
    T = z["T"].tolist()
    images = [synth.synth_pixel_values(t, cfg.vision.image_size, seed=1234 + i) for i, t in enumerate(T)]
    return z, cfg, seed, T, images


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(golden_dir, name):
    z, cfg, seed, T, images = load_case(golden_dir, name)
    sd = synth.synth_llm_state_dict(cfg, seed)
    vsd = synth.synth_vision_state_dict(cfg, seed)
    if "heavy_tail" in z:                                   # the fixture's own plan (exact edits: power-of-two gains, 16-bit-exact constants)
        plan = synth.HeavyTailPlan.from_json(json.loads(str(z["heavy_tail"])))
        sd, vsd = plan.apply(sd), plan.apply(vsd)
        assert z["stream_absmax"].min() > 500                # massive-activation channels in BOTH residual streams of the reference's forward
    ids = torch.from_numpy(z["input_ids"])
    mask = torch.from_numpy(z["attention_mask"])
    logits, im = O.forward(cfg, sd, vsd, ids, mask, images, return_intermediates=True)
    # A7: tower output (hidden_states[-2]); elementwise TOL + 2 fp32 ulps of the value (the heavy-tailed case carries |x| ~ 10^3: one ulp there is 6e-5)
    heavy = "heavy_tail" in z

    def close(a, b):
        b = torch.from_numpy(b)                              # (benign cases: TOL absolute, as since round 1)
        return bool((a - b).abs().max() <= TOL + (1e-6 * b.abs().max() if heavy else 0.0))
    assert close(im["vit"][0], z["vit_first"])
    assert close(im["vit"][-1], z["vit_last"])
    assert abs(im["vit"].double().abs().sum().item() - z["vit_sum"][1]) < 1e-6 * z["vit_sum"][1] + 1e-3
    # A9: projector shapes and values
    assert [list(v.shape) for v in im["visual"]] == z["visual_shapes"].tolist()
    assert close(im["visual"][0][:32], z["visual0_head"])
    for v, (s, a) in zip(im["visual"], z["visual_sums"]):
        assert abs(v.double().abs().sum().item() - a) < 1e-6 * a + 1e-3
    # A10: splice
    assert list(im["inputs_embeds"].shape) == z["embeds_shape"].tolist()
    assert np.array_equal(im["token_types"].numpy(), z["token_types"])
    assert abs(im["inputs_embeds"].double().abs().sum().item() - z["embeds_sum"][1]) < 1e-6 * z["embeds_sum"][1] + 1e-3
    # A11: hidden after layer 1 and after the final norm (reference hidden_states = embeds, layer outs.., normed)
    ls = z["layer_sums"]
    assert abs(im["layers"][0].double().abs().sum().item() - ls[1][1]) < 2e-6 * ls[1][1]
    assert abs(im["final"].double().abs().sum().item() - ls[-1][1]) < 2e-6 * ls[-1][1]
    # A12: logits
    assert len(logits) == ids.shape[0]                     # one logits tensor per SAMPLE (a sample may hold several <image> tokens)
    for b in range(ids.shape[0]):
        ref = torch.from_numpy(z[f"logits{b}"])
        assert logits[b].shape == ref.shape == (T[b], 1)
        assert (logits[b] - ref).abs().max() < TOL


def test_oracle_degenerate_samples_as_the_reference(golden_dir):
    """helpers.json "forward_errors": what the reference does with a sample without <image> (NaN logits of shape [T,1]) and with
    visual rows that do not divide by images[b].shape[0] (RuntimeError from the head's reshape)."""
    from videoitg_amd.config import tiny_config
    ka = json.load(open(os.path.join(golden_dir, "helpers.json")))["forward_errors"]
    for name, rec in ka.items():
        cfg = tiny_config(**rec["cfg"])
        sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
        images = [synth.synth_pixel_values(t, cfg.vision.image_size, seed=1234 + i) for i, t in enumerate(rec["T"])]
        ids = torch.tensor(rec["ids"])
        if rec["error"]:
            with pytest.raises(RuntimeError):
                O.forward(cfg, sd, vsd, ids, None, images)
        else:
            lg = O.forward(cfg, sd, vsd, ids, None, images)
            assert [list(l.shape) for l in lg] == rec["logits_shapes"]
            assert all(l.isnan().all() for l in lg) == rec["logits_all_nan"]


def test_padded_batch_equals_single(golden_dir):
    """Row of a right-padded batch == the same video alone (key padding masked out)."""
    a = np.load(os.path.join(golden_dir, "pool_post.npz"))
    b = np.load(os.path.join(golden_dir, "batch2_pad.npz"))
    assert np.abs(a["logits0"] - b["logits0"]).max() < 1e-6


def test_rank_frames_stable_ties():
    lg = torch.tensor([0.0, 2.0, 2.0, -1.0, 2.0])
    s, order = O.rank_frames(lg)
    assert order.tolist() == [1, 2, 4, 0, 3]
    assert O.topk_selection(lg, [10, 20, 30, 40, 50], 2) == [20, 30]


def test_pooled_grid_table():
    # SURVEY.md Appendix A table (mlp_proj.py:48,54)
    want = {16: 27, 22: 27, 23: 26, 32: 22, 64: 16, 128: 11, 200: 9, 256: 8, 512: 5, 1024: 4}
    for T, hw in want.items():
        assert O.pooled_grid(T, 27, 16384) == hw
        assert ITGConfig().pooled_grid(T) == hw


def test_oracle_training_mode_pooled_grids(golden_dir):
    """the projector's TRAINING-mode pooling (a random grid per video, mlp_proj.py:51-52).  train_pool.npz holds the
    reference's forward in .train() mode under a fixed torch seed and the grids that seed draws; the oracle with those grids matches."""
    z, cfg, seed, T, images = load_case(golden_dir, "train_pool")
    sd, vsd = synth.synth_llm_state_dict(cfg, seed), synth.synth_vision_state_dict(cfg, seed)
    grids = z["grids"].tolist()
    assert grids == [3, 2]                                              # (not the eval rule's [3, 3])
    import math
    torch.manual_seed(int(z["torch_seed"]))                             # the draw sequence of the reference: one randint per video, in order
    drawn = [min(torch.randint(cfg.vision_min_num, math.floor((cfg.vision_token_num / t) ** 0.5) + 1, (1,)).item(), cfg.vision.grid) for t in T]
    assert drawn == grids
    lg = O.forward(cfg, sd, vsd, torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]), images, pooled_grids=grids)
    for b in range(len(T)):
        assert (lg[b] - torch.from_numpy(z[f"logits{b}"])).abs().max() < TOL


def _structure(golden_dir, seed):
    import structures as ST
    z = np.load(os.path.join(golden_dir, "random_structures.npz"))
    s = ST.draw(seed)
    # the committed fixture holds the inputs the reference saw: the draw rule must still produce them
    assert np.array_equal(z[f"s{seed}_ids"], s["ids"].numpy()) and np.array_equal(z[f"s{seed}_mask"], s["mask"].numpy())
    assert z[f"s{seed}_T"].tolist() == s["T"] and z[f"s{seed}_meta"].tolist() == [s["side"], s["budget"], int(s["two_first"])]
    return z, s, ST


@pytest.mark.parametrize("seed", list(range(16)))
def test_oracle_matches_reference_on_random_batch_structures(golden_dir, seed):
    """Round 6: sixteen seeded batch structures nobody chose (tests/structures.py: 1-3 prompts, left / right padding, random text around the <image> tokens,
    clips of 1-11 frames, two clips in one prompt, pooled 1 x 1 .. 3 x 3 or unpooled, two image sides) through THE REFERENCE
    (tests/golden/make_random_structures.py) - the oracle reproduces its logits to the same 2e-5 as on the hand-written cases."""
    z, s, ST = _structure(golden_dir, seed)
    cfg = ST.config(s)
    sd, vsd = synth.synth_llm_state_dict(cfg, seed), synth.synth_vision_state_dict(cfg, seed)
    logits = O.forward(cfg, sd, vsd, s["ids"], s["mask"], ST.images(s))
    assert len(logits) == s["B"]
    for b in range(s["B"]):
        ref = torch.from_numpy(z[f"s{seed}_logits{b}"])
        assert logits[b].shape == ref.shape
        assert (logits[b] - ref).abs().max() < TOL, s["what"]


@pytest.mark.parametrize("seed", list(range(12)))
def test_oracle_matches_reference_on_random_architectures(golden_dir, seed):
    """Round 6: twelve drawn ARCHITECTURES (tests/structures.py: draw_arch - tower of 1-4 heads, FFN widths that are multiples of 4 only, 2-4 layers, four
    image sides; decoder with GQA groups 1 / 2 / 3 / 7, FFN widths that are multiples of 8 only, 1-3 layers, two rope bases) through THE REFERENCE
    (random_archs.npz): nothing in the oracle is tied to the two architectures the other fixtures use."""
    import structures as ST
    z = np.load(os.path.join(golden_dir, "random_archs.npz"))
    s = ST.draw_arch(seed)
    assert json.loads(str(z[f"a{seed}_arch"])) == json.loads(json.dumps(s["arch"])) and np.array_equal(z[f"a{seed}_ids"], s["ids"].numpy())
    cfg = ST.arch_config(s)
    sd, vsd = synth.synth_llm_state_dict(cfg, seed), synth.synth_vision_state_dict(cfg, seed)
    logits = O.forward(cfg, sd, vsd, s["ids"], s["mask"], ST.arch_images(s))
    ref = torch.from_numpy(z[f"a{seed}_logits0"])
    assert logits[0].shape == ref.shape and (logits[0] - ref).abs().max() < TOL, s["what"]
