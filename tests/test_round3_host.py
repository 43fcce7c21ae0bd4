"""Host-side pieces added in round 3 (no GPU)."""
import importlib.util
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _infer():
    spec = importlib.util.spec_from_file_location("vitg_infer", os.path.join(ROOT, "infer.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_save_selected_frames_names_and_count(tmp_path):
    """infer.py:81-94 of the reference: frame_<rank>_idx<original index>.jpg, one JPEG per selected frame"""
    from PIL import Image
    m = _infer()
    sel = [3, 40, 41, 977]
    m.save_selected_frames("synthetic", sel, save_dir=str(tmp_path / "vis"))
    names = sorted(os.listdir(tmp_path / "vis"))
    assert names == [f"frame_{i:03d}_idx{f}.jpg" for i, f in enumerate(sel)]
    im = Image.open(tmp_path / "vis" / names[0])
    assert im.format == "JPEG" and im.size == (160, 120)


def test_fp8_selector_is_normalised():
    from videoitg_amd.engine import normalise_fp8
    import numpy as np
    assert normalise_fp8(True) is True and normalise_fp8(1) is True and normalise_fp8(np.bool_(True)) is True
    assert normalise_fp8(False) is False and normalise_fp8(0) is False and normalise_fp8(None) is False
    assert normalise_fp8("llm") == frozenset({"llm.qkv", "llm.o", "llm.gu", "llm.down"}) and normalise_fp8("vit.fc1") == frozenset({"vit.fc1"})
    for bad in ("both", "tower", 1.5, [1]):
        with pytest.raises(ValueError):
            normalise_fp8(bad)


def test_graph_cache_serves_only_what_a_capture_can(monkeypatch):
    """a prompt whose 32-token bucket passes tokenizer_model_max_length, or a prompt with no / two <image> tokens, must go to
    the eager forward instead of raising out of the graph cache"""
    from videoitg_amd.config import tiny_config
    from videoitg_amd.engine import GraphCache

    class Eng:                                   # layout decisions need the config only
        cfg = tiny_config(vision_token_num=72, tokenizer_model_max_length=100)
    gc = GraphCache.__new__(GraphCache)
    gc.eng, gc.bucket = Eng(), 32
    pix = torch.zeros(8, 3, 56, 56)              # 8 frames, 4x4 grid pooled to 3x3: 72 visual tokens
    assert not gc.serves(torch.tensor([-200, 5, 6, 7]), pix)                   # 72 visual + one 32-token bucket = 104 > 100 -> eager forward
    Eng.cfg = tiny_config(vision_token_num=72, tokenizer_model_max_length=104)
    assert gc.serves(torch.tensor([-200, 5, 6, 7]), pix)
    Eng.cfg = tiny_config(vision_token_num=72, tokenizer_model_max_length=None)
    assert gc.serves(torch.tensor([-200, 5, 6, 7]), pix)
    assert not gc.serves(torch.tensor([5, 6, 7]), pix)
    assert not gc.serves(torch.tensor([-200, 5, -200, 7]), pix)


def test_planted_inputs_are_a_brightness_step_on_the_seeded_frames():
    """synth.planted_inputs (the decidable Top-k case for the lower-precision tiers): k seeded frames bright, the rest dark, same prompt and
    same base pixels as workload_inputs, values exactly representable in bf16 and fp16; fixture_inputs rebuilds both kinds of fixture"""
    import torch
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig
    cfg = ITGConfig()
    ids0, pix0 = synth.workload_inputs(cfg, 12, 28, 7, 5)
    ids, pix, planted = synth.planted_inputs(cfg, 12, 28, 7, 5, k=3, plant_seed=99)
    assert torch.equal(ids, ids0) and planted == synth.planted_frames(12, 3, 99) and len(set(planted)) == 3 and planted == sorted(planted)
    assert synth.planted_frames(12, 3, 99) != synth.planted_frames(12, 3, 100)
    assert torch.equal(pix, pix.to(torch.bfloat16).float()) and torch.equal(pix, pix.half().float())
    for t in range(12):
        want = synth.round_16(pix0[t] * 0.5 + (0.5 if t in planted else -0.5))
        assert torch.equal(pix[t], want)
    assert pix[planted].mean() > 0.3 and pix[[t for t in range(12) if t not in planted]].mean() < -0.3
    rec = dict(frames=12, resolution=28, prompt_len=7, pixel_seed=5, plant=dict(k=3, seed=99, frames=planted))
    a, b = synth.fixture_inputs(cfg, rec)
    assert torch.equal(a, ids) and torch.equal(b, pix)
    rec.pop("plant")
    a, b = synth.fixture_inputs(cfg, rec)
    assert torch.equal(b, pix0)


def test_q_prescale_touches_only_the_q_rows_of_the_tower():
    """PackedWeights.vit_q_prescaled folds head_dim^-0.5 * log2(e) into the q rows and the q bias of every tower layer, in fp32 before the one
    rounding to 16 bits; k / v rows, the other GEMMs and the decoder are untouched"""
    import math
    import torch
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.engine import PackedWeights
    cfg = tiny_config()
    vsd = synth.synth_vision_state_dict(cfg, 0)
    packs = {}
    for qp in (True, False):
        pw = PackedWeights(cfg=cfg, dtype=torch.float16, device=torch.device("cpu"), vit_q_prescaled=qp)
        pw.pack_vision_layer(vsd, 0)
        packs[qp] = pw.vit[0]
    Dv = cfg.vision.hidden_size
    c = cfg.vision.head_dim ** -0.5 * math.log2(math.e)
    p = "vision_model.encoder.layers.0.self_attn."
    want_q = (vsd[p + "q_proj.weight"].float() * c).half()
    assert torch.equal(packs[True].qkv_w[:Dv, :Dv], want_q)
    assert torch.equal(packs[False].qkv_w[:Dv, :Dv], vsd[p + "q_proj.weight"].half())
    assert torch.equal(packs[True].qkv_w[Dv:], packs[False].qkv_w[Dv:])
    assert torch.allclose(packs[True].qkv_b[:Dv], vsd[p + "q_proj.bias"].float() * c) and torch.equal(packs[True].qkv_b[Dv:], packs[False].qkv_b[Dv:])
    for n in ("out_w", "fc1_w", "fc2_w", "out_b", "fc1_b", "fc2_b"):
        assert torch.equal(getattr(packs[True], n), getattr(packs[False], n))
