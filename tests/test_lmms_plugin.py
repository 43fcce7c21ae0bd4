"""lmms-eval plugin twin: registration contract and (on the GPU) the results.jsonl line format."""
import json
import types

import numpy as np
import pytest
import torch


def test_plugin_registration_contract():
    from videoitg_amd.lmms_plugin import models
    assert models.AVAILABLE_MODELS == {"videoitg": "VideoITG"}       # lmms_eval/models/__init__.py:84-89
    mod = __import__("videoitg_amd.lmms_plugin.models.videoitg", fromlist=["VideoITG"])
    cls = getattr(mod, models.AVAILABLE_MODELS["videoitg"])
    for name in ("generate_until", "loglikelihood", "generate_until_multi_round", "rank", "world_size", "device", "batch_size",
                 "tokenizer", "model", "config", "eot_token_id", "max_length"):
        assert hasattr(cls, name), name
    import inspect
    params = inspect.signature(cls.__init__).parameters
    for k, d in dict(num_frames=512, target_fps=1, multiple=1, batch_size=1, use_flash_attn=True).items():
        assert params[k].default == d


@pytest.mark.gpu
def test_generate_until_writes_reference_jsonl(tmp_path):
    from PIL import Image
    from videoitg_amd.lmms_plugin.models.videoitg import VideoITG
    from videoitg_amd.mm_utils import get_seq_frames

    def fake_loader(path, num_frames, target_fps):
        idx = get_seq_frames(900, 30.0, target_fps, num_frames)            # 30 sampled ids out of a 900-frame "video"
        rng = np.random.default_rng(len(path))
        return [Image.fromarray(rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)) for _ in idx], idx

    lm = VideoITG(pretrained="synthetic:tiny:0", device="cuda:0", num_frames=8, output_dir=str(tmp_path), video_loader=fake_loader)
    lm.task_dict = {"t": {"test": {0: {"v": "a.mp4"}, 1: {"v": "bb.mp4"}}}}
    reqs = [types.SimpleNamespace(args=("which frame?", {}, lambda d: [d["v"]], i, "t", "test")) for i in (0, 1)]
    res = lm.generate_until(reqs)
    lines = [json.loads(l) for l in open(tmp_path / "results.jsonl")]
    assert lines == res and len(lines) == 2
    for o in lines:
        assert list(o) == ["index", "logits", "num_frames", "contexts", "video_path", "doc_id"]
        assert sorted(o["index"]) == get_seq_frames(900, 30.0, 1, 8) and o["num_frames"] == 1
        assert o["logits"] == sorted(o["logits"], reverse=True) and all(round(v, 2) == v for v in o["logits"])
