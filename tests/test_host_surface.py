"""CPU: host-side behaviour of the operator surface (builder / model object / config) — no GPU arithmetic."""
import json
import os

import numpy as np
import pytest
import torch

from videoitg_amd import synth
from videoitg_amd.config import ITGConfig, tiny_config
from videoitg_amd.image_processor import SiglipImageProcessor
from videoitg_amd.model.builder import (ByteTokenizer, load_pretrained_model, normalise_vision_keys, split_checkpoint)
from videoitg_amd.model.grounding_qwen2 import EagleQwenG


def test_config_defaults_and_pooled_grid():
    cfg = ITGConfig()
    assert (cfg.head_dim, cfg.vision.head_dim, cfg.vision.grid, cfg.vision.num_patches) == (128, 72, 27, 729)
    assert cfg.num_vit_layers_needed() == 26                       # hidden_states[-2] of a 27-layer tower
    assert cfg.pooled_grid(512) == 5 and cfg.pooled_grid(64) == 16 and cfg.pooled_grid(16) == 27


def test_config_from_pretrained_applies_reference_defaults(tmp_path):
    d = {"hidden_size": 256, "num_attention_heads": 2, "num_key_value_heads": 1, "vision_token_num": 72}
    (tmp_path / "config.json").write_text(json.dumps(d))
    cfg = ITGConfig.from_pretrained(str(tmp_path))
    assert cfg.mm_use_4_vision_tokens is True                      # eagle_archv1.py:223 default when the key is absent
    with pytest.raises(ValueError):                                # ... which is not part of the selector path
        EagleQwenG(cfg, {}, {})


def test_model_object_rejects_unknown_projector_and_cpu_forward():
    cfg = tiny_config(mm_projector_type="linear")
    with pytest.raises(ValueError):                                # multimodal_projector/builder.py:69
        EagleQwenG(cfg, {}, {})
    m = EagleQwenG(tiny_config(), {}, {}, device="cpu")
    assert m.half() is m and m.eval() is m and m.to("cpu") is m and m.dtype == torch.float16
    with pytest.raises(ValueError):
        m.to(torch.float32)                                        # MFMA inputs are 16-bit
    with pytest.raises(RuntimeError):                              # no CPU path, loudly
        m(torch.tensor([[-200, 5]]), images=[torch.zeros(2, 3, 56, 56)])
    with pytest.raises(ValueError):
        m(None, images=None)


def test_builder_surface_without_gpu():
    tok, model, proc, ctx = load_pretrained_model("synthetic:tiny:0", None, "VideoITG-tiny", device_map="cuda:0")
    assert isinstance(tok, ByteTokenizer) and ctx == 2048 and proc is model.get_vision_tower().image_processor
    assert model.config.mm_projector_type == "seq_mlp" and str(model.device) == "cuda:0"
    assert model.get_vision_tower().num_patches == 16 and model.tie_weights() is None
    with pytest.raises(NotImplementedError):
        load_pretrained_model("synthetic:tiny", "some/base", "x")
    with pytest.raises(NotImplementedError):
        load_pretrained_model("synthetic:tiny", None, "x", load_8bit=True)


def test_checkpoint_directory_layouts(tmp_path):
    """HF-layout directory with safetensors + a separate tower directory; tower keys in all three layouts seen in the wild."""
    from safetensors.torch import save_file
    cfg = tiny_config()
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    ck = tmp_path / "ckpt"
    tw = ck / "vision_tower"
    tw.mkdir(parents=True)
    d = cfg.to_dict()
    d["mm_vision_tower"] = "google/siglip-so400m-patch14-384"
    (ck / "config.json").write_text(json.dumps(d))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ck / "model.safetensors"))
    save_file({k: v.contiguous() for k, v in vsd.items()}, str(tw / "model.safetensors"))
    from videoitg_amd.model.builder import _load_directory
    cfg2, sd2, vsd2 = _load_directory(str(ck))
    assert cfg2.hidden_size == cfg.hidden_size and set(sd2) == set(sd) and set(vsd2) == set(vsd)
    assert torch.equal(sd2["out_proj.weight"], sd["out_proj.weight"])
    # embedded (4.47 naming) and flattened (5.x naming) tower keys normalise to the Hub layout
    emb = {"model.vision_tower.vision_tower." + k: v for k, v in vsd.items()}
    flat = {"model.vision_tower.vision_tower." + k[len("vision_model."):]: v for k, v in vsd.items()}
    for layout in (emb, flat):
        llm, vis = split_checkpoint({**sd, **layout})
        assert set(llm) == set(sd) and set(vis) == set(vsd)
    assert set(normalise_vision_keys(vsd)) == set(vsd)
    # the Hub repo google/siglip-so400m-patch14-384 is the full two-tower model: its text tower has layers of the SAME shapes
    # (text_model.encoder.layers.N...) plus logit_scale / logit_bias; none of it may leak into (or overwrite) the vision tower
    two_tower = dict(vsd)
    for k, v in vsd.items():
        if ".encoder.layers." in k or ".embeddings." in k:
            two_tower["text_model." + k[len("vision_model."):]] = torch.full_like(v, 1e9)
    two_tower["logit_scale"] = torch.zeros(1)
    two_tower["logit_bias"] = torch.zeros(1)
    picked = normalise_vision_keys(two_tower)
    assert set(picked) == set(vsd) and all(torch.equal(picked[k], vsd[k]) for k in vsd)
    (tw / "model.safetensors").unlink()
    os.rmdir(tw)
    with pytest.raises(FileNotFoundError):
        _load_directory(str(ck))


def test_image_processor_matches_siglip_definition():
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (3, 50, 70, 3), dtype=np.uint8)
    p = SiglipImageProcessor(size=56)
    out = p.preprocess(frames, return_tensors="pt")["pixel_values"]
    assert out.shape == (3, 3, 56, 56) and out.dtype == torch.float32 and out.min() >= -1 and out.max() <= 1
    from PIL import Image
    ref = np.asarray(Image.fromarray(frames[1]).resize((56, 56), resample=Image.BICUBIC), dtype=np.float32) / 255.0
    ref = ((ref - 0.5) / 0.5).transpose(2, 0, 1)
    assert np.abs(out[1].numpy() - ref).max() < 1e-6
    same = p.preprocess([Image.fromarray(f) for f in frames], return_tensors="pt")["pixel_values"]
    assert torch.equal(out, same)                                   # ndarray and PIL-list inputs agree
    sq = rng.integers(0, 256, (2, 56, 56, 3), dtype=np.uint8)
    exact = p.preprocess(sq, return_tensors="pt")["pixel_values"]                       # already 56x56: no resize path
    assert np.abs(exact[0].numpy() - ((sq[0].astype(np.float32) / 255 - 0.5) / 0.5).transpose(2, 0, 1)).max() < 1e-6
    with pytest.raises(ValueError):
        p.preprocess(frames, return_tensors="tf")


def test_model_object_survives_accelerate_wrap_on_cpu():
    """lmms_eval/models/videoitg.py:193-205: accelerator.prepare_model(model, evaluation_mode=True).  No GPU here: the wrap must keep
    the operator surface intact and the forward must still fail loudly (there is no CPU path to fall back to)."""
    import pytest
    import torch
    from accelerate import Accelerator
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.model.grounding_qwen2 import EagleQwenG
    cfg = tiny_config()
    m = EagleQwenG(cfg, synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0), torch.float16, "cuda:0")
    w = Accelerator(cpu=True).prepare_model(m, evaluation_mode=True)
    w.eval()
    w.tie_weights()
    assert w.config is cfg and w.get_vision_tower().image_processor is not None
    with pytest.raises(RuntimeError, match="no CPU path"):
        w(torch.tensor([[-200, 5]]), attention_mask=None, images=[torch.zeros(2, 3, 56, 56)])


def test_written_checkpoint_layout_loads_through_the_directory_reader(tmp_path):
    """CPU twin of tests/test_gpu_checkpoint.py on the tiny architecture: synth_device.write_hf_checkpoint lays the model out like a release (sharded
    safetensors + index + config + tokenizer; the SigLIP repo beside it with the 27th-layer / head / text-tower keys the selector never reads) and
    `_load_directory` + AutoTokenizer get exactly the tensors back - the text tower's poisoned layers (same shapes as the vision layers) never leak."""
    from videoitg_amd.model.builder import _load_directory
    from videoitg_amd.synth_device import write_hf_checkpoint
    cfg = tiny_config()
    info = write_hf_checkpoint(cfg, str(tmp_path / "ckpt"), str(tmp_path / "siglip"), seed=3, device="cpu", shard_bytes=300_000)
    assert info["shards"] >= 3 and os.path.exists(tmp_path / "ckpt" / "model.safetensors.index.json")
    cfg2, sd, vsd = _load_directory(str(tmp_path / "ckpt"))
    want_l = synth.counter_state_dict(list(synth.llm_param_specs(cfg)), 3)
    want_v = synth.counter_state_dict(list(synth.vision_param_specs(cfg)), 3)
    assert cfg2.to_dict()["hidden_size"] == cfg.hidden_size and cfg2.vision.intermediate_size == cfg.vision.intermediate_size
    assert set(sd) == set(want_l) and all(torch.equal(sd[k].float(), want_l[k]) for k in sd)            # 16-bit exact values: fp16 files lose nothing
    assert set(want_v) <= set(vsd) and all(torch.equal(vsd[k].float(), want_v[k]) for k in want_v)
    assert not any("text_model" in k or k.startswith("logit_") for k in vsd)
    assert any(k.startswith("vision_model.head.") for k in vsd) and "vision_model.post_layernorm.weight" in vsd    # carried, never packed
    tok, model, proc, ctx = load_pretrained_model(str(tmp_path / "ckpt"), None, "VideoITG-tiny", device_map="cuda:0")
    assert not isinstance(tok, ByteTokenizer) and len(tok("which frame ?").input_ids) == 3


def test_split_weights_layout_and_refusals_on_cpu():
    """host side of the operand-split tier (videoitg_amd.engine.SplitWeights): every GEMM weight that meets a [hi | lo] operand is laid twice side by side;
    the tier refuses the combinations it does not define (bf16 pairs, fp8 weights, re-rounded q rows) before anything is launched"""
    from videoitg_amd.engine import PackedWeights, SplitWeights
    cfg = tiny_config()
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    pw = PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cpu", vit_q_prescaled=False)
    sw = pw.doubled()
    assert sw is pw.doubled()                                         # built once
    assert len(sw.vit) == cfg.num_vit_layers_needed() and len(sw.llm) == cfg.num_hidden_layers
    for lw, l2 in zip(pw.vit, sw.vit):
        for name, w in (("qkv", lw.qkv_w), ("out", lw.out_w), ("fc1", lw.fc1_w), ("fc2", lw.fc2_w)):
            k = w.shape[1]
            assert l2[name].shape == (w.shape[0], 2 * k) and torch.equal(l2[name][:, :k], w) and torch.equal(l2[name][:, k:], w)
    for lw, l2 in zip(pw.llm, sw.llm):
        assert l2["gu"].shape == (lw.gu_w.shape[0], 2 * lw.gu_w.shape[1]) and torch.equal(l2["down"][:, lw.down_w.shape[1]:], lw.down_w)
    assert sw.proj0.shape[1] == 2 * pw.proj0_w.shape[1] and sw.proj2.shape[1] == 2 * pw.proj2_w.shape[1]
    # the softmax scale is NOT folded into the q rows: they are the checkpoint's own 16-bit values
    Dv = cfg.vision.hidden_size
    assert torch.equal(pw.vit[0].qkv_w[:Dv, :Dv], vsd["vision_model.encoder.layers.0.self_attn.q_proj.weight"].half())
    with pytest.raises(ValueError, match="vit_q_prescaled"):
        SplitWeights(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cpu", vit_q_prescaled=True))
    with pytest.raises(ValueError, match="float16"):
        SplitWeights(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.bfloat16, "cpu", vit_q_prescaled=False))
    m = EagleQwenG(cfg, sd, vsd, torch_dtype=torch.float16, device="cpu")
    m.operand_split = True
    with pytest.raises(RuntimeError):                                 # still no CPU path, whatever the tier
        m(torch.tensor([[-200, 5]]), images=[torch.zeros(2, 3, 56, 56)])
