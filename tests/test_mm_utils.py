"""Host helpers vs known answers recorded from the reference (tests/golden/helpers.json, written by
tests/golden/make_golden.py running /root/reference/eagle/mm_utils.py and lmms_eval/models/videoitg.py)."""
import json
import os
import types

import pytest
import torch

from videoitg_amd import mm_utils
from videoitg_amd.constants import IMAGE_TOKEN_INDEX, IGNORE_INDEX, DEFAULT_IMAGE_TOKEN


@pytest.fixture(scope="module")
def ka(golden_dir):
    with open(os.path.join(golden_dir, "helpers.json")) as f:
        return json.load(f)


class Tok:
    def __init__(self, bos=None):
        self.bos_token_id = bos

    def __call__(self, s):
        ids = [ord(c) for c in s]
        if self.bos_token_id is not None:
            ids = [self.bos_token_id] + ids
        return types.SimpleNamespace(input_ids=ids)


def test_constants():
    assert (IGNORE_INDEX, IMAGE_TOKEN_INDEX, DEFAULT_IMAGE_TOKEN) == (-100, -200, "<image>")


def test_get_frame_indices(ka):
    assert len(ka["get_frame_indices"]) >= 10
    for e in ka["get_frame_indices"]:
        assert mm_utils.get_frame_indices(*e["args"]) == e["out"], e["args"]


def test_get_frame_indices_survey_values():
    # SURVEY.md §8(a) row A1
    out = mm_utils.get_frame_indices(3000, 30.0, 2, 512)
    assert len(out) == 200 and out[:2] == [0, 15] and out[-1] == 2985
    out = mm_utils.get_frame_indices(18000, 29.97, 2, 512)
    assert out[:6] == [15, 60, 90, 120, 165, 195] and out[-3:] == [17910, 17955, 17985]
    assert len(mm_utils.get_frame_indices(500, 25.0, 2, 512)) == 42


def test_get_seq_frames(ka):
    for e in ka["get_seq_frames"]:
        got = mm_utils.get_seq_frames(e["total"], e["fps"], e["target_fps"], e["num_frames"], e["multiple"])
        assert got == e["out"], e


def test_get_seq_frames_zero_stride_raises():
    with pytest.raises(ValueError):
        mm_utils.get_seq_frames(100, 0.4, 1, 8)       # round(0.4) == 0 -> range() step 0, as in the reference


def test_tokenizer_image_token(ka):
    for e in ka["tokenizer_image_token"]:
        assert mm_utils.tokenizer_image_token(e["prompt"], Tok(e["bos"]), -200) == e["out"], e
    t = mm_utils.tokenizer_image_token("<image>hi\n", Tok(), return_tensors="pt")
    assert t.dtype == torch.long and t.tolist() == [-200, 104, 105, 10]
    with pytest.raises(ValueError):
        mm_utils.tokenizer_image_token("<image>", Tok(), return_tensors="np")


def test_get_model_name_from_path(ka):
    for e in ka["get_model_name_from_path"]:
        assert mm_utils.get_model_name_from_path(e["path"]) == e["out"]


def test_pad_sequence_sides():
    seqs = [torch.tensor([1, 2, 3]), torch.tensor([4])]
    r = mm_utils.pad_sequence(types.SimpleNamespace(padding_side="right"), seqs, True, 0)
    l = mm_utils.pad_sequence(types.SimpleNamespace(padding_side="left"), seqs, True, 0)
    assert r.tolist() == [[1, 2, 3], [4, 0, 0]] and l.tolist() == [[1, 2, 3], [0, 0, 4]]


# ---- the reference's integer helpers on a sweep of seeded random arguments (tests/golden/make_helpers_sweep.py ran THEM; data only) ----------------
def _digest(xs):
    import hashlib
    xs = [int(x) for x in xs]
    return {"n": len(xs), "head": xs[:3], "tail": xs[-3:], "sha1": hashlib.sha1(",".join(map(str, xs)).encode()).hexdigest()}


def _outcome(fn, *args):
    try:
        return _digest(fn(*args))
    except Exception as e:
        return {"error": type(e).__name__}


@pytest.fixture(scope="module")
def sweep(golden_dir):
    with open(os.path.join(golden_dir, "helpers_sweep.json")) as f:
        return json.load(f)


def test_frame_samplers_equal_the_reference_on_a_sweep(sweep):
    """A1 / A1': 400 seeded draws each (frame counts 1 .. 200 000, frame rates incl. 0.4 / 4.99 / 5.01 / 23.976 / 59.94, sample budgets 1 .. 600,
    `multiple` padding): the index list - or the exception type the reference raises (zero stride) - is identical, by SHA-1 of the list"""
    assert len(sweep["get_frame_indices"]) >= 400 and len(sweep["get_seq_frames"]) >= 400
    for e in sweep["get_frame_indices"]:
        assert _outcome(mm_utils.get_frame_indices, *e["args"]) == e["out"], e["args"]
    n_err = 0
    for e in sweep["get_seq_frames"]:
        assert _outcome(mm_utils.get_seq_frames, *e["args"]) == e["out"], e["args"]
        n_err += "error" in e["out"]
    assert n_err >= 10                                   # the sweep does reach the reference's failure mode


def test_prompt_splice_and_padding_equal_the_reference_on_a_sweep(sweep):
    """A4 / A5: 200 prompts of random text around 0-3 <image> tokens (with / without BOS), 60 ragged batches padded on either side"""
    for e in sweep["tokenizer_image_token"]:
        assert _outcome(mm_utils.tokenizer_image_token, e["prompt"], Tok(e["bos"]), -200) == e["out"], e["prompt"]
    assert len(sweep["pad_sequence"]) >= 60
    for e in sweep["pad_sequence"]:
        got = mm_utils.pad_sequence(types.SimpleNamespace(padding_side=e["side"]), [torch.tensor(s) for s in e["seqs"]], True, e["pad"])
        assert got.tolist() == e["out"], e
