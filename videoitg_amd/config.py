"""Hyper-parameters of the VideoITG frame-selector path.

The reference reads these from two HF ``config.json`` files (the ITG checkpoint and the SigLIP
tower) with ``getattr(config, key, default)``; the keys and defaults mirrored here are the ones
the hot path consults:

* ``mm_use_4_vision_tokens``   /root/reference/eagle/model/eagle_archv1.py:223 (default True!)
* ``mm_projector_type``        /root/reference/eagle/model/multimodal_projector/builder.py:49
* ``vision_token_num`` / ``vision_min_num``  .../multimodal_projector/builder.py:67
* ``mm_vision_select_layer`` / ``mm_vision_select_feature``  .../multimodal_encoder/clip_encoder.py:106-107
* ``tokenizer_model_max_length``  eagle_archv1.py:341-345
* ``tokenizer_padding_side``      eagle_archv1.py:361

Numeric defaults are the public model-card values of ``google/siglip-so400m-patch14-384`` and
``Qwen/Qwen2-7B-Instruct``.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import asdict, dataclass, field
from typing import Optional


@dataclass
class VisionConfig:
    image_size: int = 384
    patch_size: int = 14
    hidden_size: int = 1152
    intermediate_size: int = 4304
    num_hidden_layers: int = 27
    num_attention_heads: int = 16
    layer_norm_eps: float = 1e-6
    num_channels: int = 3

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid


@dataclass
class ITGConfig:
    vocab_size: int = 152064
    hidden_size: int = 3584
    intermediate_size: int = 18944
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1_000_000.0
    max_position_embeddings: int = 32768
    # multimodal keys (eagle_archv1.py / multimodal_projector/builder.py)
    mm_vision_tower: str = "google/siglip-so400m-patch14-384"
    mm_projector_type: str = "seq_mlp"
    mm_use_4_vision_tokens: bool = False
    mm_hidden_size: int = 1152
    mm_vision_select_layer: int = -2
    mm_vision_select_feature: str = "patch"
    vision_token_num: int = 16384
    vision_min_num: int = 1
    tokenizer_model_max_length: Optional[int] = None
    tokenizer_padding_side: str = "right"
    max_sequence_length: Optional[int] = None
    vision: VisionConfig = field(default_factory=VisionConfig)

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def num_vit_layers_needed(self) -> int:
        """hidden_states[select_layer] of an L-layer tower = output after L+1+select_layer layers
        (select_layer<0).  -2 -> L-1 layers (clip_encoder.py:123-129)."""
        L = self.vision.num_hidden_layers
        sl = self.mm_vision_select_layer
        n = L + 1 + sl if sl < 0 else sl
        if not 0 <= n <= L:
            raise ValueError(f"mm_vision_select_layer {sl} out of range for {L} layers")
        return n

    def pooled_grid(self, num_frames: int, grid: Optional[int] = None) -> int:
        """HW = min(floor(sqrt(vision_token_num / T)), ori_HW) with ori_HW the ACTUAL token grid of the input
        (mlp_proj.py:46-54, eval branch: ori_HW = int(P ** 0.5)); `grid` defaults to the tower's native one."""
        hw = math.floor((self.vision_token_num / num_frames) ** 0.5)
        return min(hw, self.vision.grid if grid is None else grid)

    def to_dict(self) -> dict:
        return asdict(self)

    @staticmethod
    def from_dict(d: dict) -> "ITGConfig":
        d = dict(d)
        v = d.pop("vision", None)
        known = {k: d[k] for k in ITGConfig.__dataclass_fields__ if k in d and k != "vision"}
        cfg = ITGConfig(**known)
        if v is not None:
            cfg.vision = VisionConfig(**{k: v[k] for k in VisionConfig.__dataclass_fields__ if k in v})
        return cfg

    @staticmethod
    def from_pretrained(path: str) -> "ITGConfig":
        """Read ``config.json`` of an ITG checkpoint directory (+ ``vision_config.json`` or a
        ``vision_config`` sub-dict for the tower).  Honours the reference's defaults for absent keys."""
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        if "mm_use_4_vision_tokens" not in d:
            d["mm_use_4_vision_tokens"] = True  # eagle_archv1.py:223 default
        v = d.get("vision_config") or d.get("vision")
        vp = os.path.join(path, "vision_config.json")
        if v is None and os.path.exists(vp):
            with open(vp) as f:
                v = json.load(f)
        d["vision"] = v
        return ITGConfig.from_dict(d)


def tiny_config(**over) -> ITGConfig:
    """Small config that keeps every structural quirk of the full model: head_dim 128 with GQA,
    q/k/v bias, ViT head_dim 72, an FFN width that is not a multiple of 32, select_layer -2."""
    cfg = ITGConfig(
        vocab_size=320, hidden_size=256, intermediate_size=592, num_hidden_layers=2,
        num_attention_heads=2, num_key_value_heads=1, mm_hidden_size=144, vision_token_num=72,
        vision=VisionConfig(image_size=56, patch_size=14, hidden_size=144, intermediate_size=272,
                            num_hidden_layers=3, num_attention_heads=2),
    )
    for k, v in over.items():
        if k == "vision":
            for kk, vv in v.items():
                setattr(cfg.vision, kk, vv)
        else:
            setattr(cfg, k, v)
    return cfg
