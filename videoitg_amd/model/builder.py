"""``load_pretrained_model`` twin (row A6 of SURVEY.md §8a; /root/reference/eagle/model/builder.py:42-176).

Same signature and 4-tuple result ``(tokenizer, model, image_processor, context_len)``.  ``model_path`` is

* a local HF-layout checkpoint directory: ``config.json`` + ``*.safetensors`` / ``pytorch_model*.bin`` with the
  key names of SURVEY.md §8(b); the SigLIP tower is taken from (first hit) ``$VIDEOITG_VISION_TOWER``,
  ``config.mm_vision_tower`` if it is a local directory, ``<model_path>/vision_tower``, or tower keys embedded in the
  checkpoint.  The reference pulls the tower from the Hub (builder.py:163-165); there is no network here;
* ``"synthetic[:tiny|full][:seed]"``: seeded random-init weights of the named architecture (videoitg_amd.synth) with
  a byte-level stand-in tokenizer — the only option offline, since ``nvidia/VideoITG-8B`` is unreachable.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Tuple

import torch

from ..config import ITGConfig, tiny_config
from ..constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN
from .grounding_qwen2 import EagleQwenG

SD = Dict[str, torch.Tensor]


class ByteTokenizer:
    """Stand-in tokenizer for synthetic checkpoints: one id per UTF-8 byte (+3), ids 0/1/2 = pad/bos/eos."""
    pad_token_id, bos_token_id, eos_token_id = 0, None, 2
    padding_side = "right"

    def __init__(self, vocab_size: int):
        self.vocab_size = vocab_size
        self.added = []

    def __call__(self, text: str):
        from types import SimpleNamespace
        return SimpleNamespace(input_ids=[3 + (b % (self.vocab_size - 3)) for b in text.encode("utf-8")])

    def add_tokens(self, toks, special_tokens=False):
        self.added.extend(toks)
        return len(toks)

    def __len__(self):
        return self.vocab_size

    def decode(self, ids, **_):
        return bytes(max(0, int(i) - 3) % 256 for i in ids).decode("utf-8", "replace")


def _read_tensors(path: str) -> SD:
    out: SD = {}
    st = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if st:
        from safetensors.torch import load_file
        for f in st:
            out.update(load_file(f, device="cpu"))
        return out
    for f in sorted(glob.glob(os.path.join(path, "pytorch_model*.bin"))):
        out.update(torch.load(f, map_location="cpu", weights_only=True))
    if not out:
        raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin under {path}")
    return out


def normalise_vision_keys(sd: SD) -> SD:
    """Map any of the layouts seen in the wild onto ``vision_model.*``: the Hub repo (``vision_model.…``), tower
    keys embedded in an ITG checkpoint (``model.vision_tower.vision_tower.vision_model.…``) and the transformers-5.x
    flattened form without the ``vision_model.`` level."""
    out: SD = {}
    for k, v in sd.items():
        # the Hub SigLIP repo is the full two-tower model: its text tower has the same widths as the so400m vision tower
        # (text_model.encoder.layers.N...), so it must be dropped by NAME or it would silently overwrite the vision layers
        if k.startswith("text_model.") or ".text_model." in k or k in ("logit_scale", "logit_bias"):
            continue
        for anchor in ("embeddings.", "encoder.layers.", "post_layernorm.", "head."):
            i = k.find(anchor)
            if i >= 0 and (i == 0 or k[i - 1] == "."):
                out["vision_model." + k[i:]] = v
                break
    return out


def split_checkpoint(all_sd: SD) -> Tuple[SD, SD]:
    vis = {k: v for k, v in all_sd.items() if "vision_tower" in k or k.startswith("vision_model.")}
    llm = {k: v for k, v in all_sd.items() if k not in vis}
    return llm, normalise_vision_keys(vis)


def _load_directory(model_path: str) -> Tuple[ITGConfig, SD, SD]:
    cfg = ITGConfig.from_pretrained(model_path)
    sd, vsd = split_checkpoint(_read_tensors(model_path))
    tower_dir = None
    for cand in (os.environ.get("VIDEOITG_VISION_TOWER"), cfg.mm_vision_tower, os.path.join(model_path, "vision_tower")):
        if cand and os.path.isdir(cand):
            tower_dir = cand
            break
    if tower_dir is not None:
        vsd = normalise_vision_keys(_read_tensors(tower_dir))
        vc = os.path.join(tower_dir, "config.json")
        if os.path.exists(vc):
            with open(vc) as f:
                d = json.load(f)
            d = d.get("vision_config", d)
            for k in cfg.vision.__dataclass_fields__:
                if k in d:
                    setattr(cfg.vision, k, d[k])
    if not vsd:
        raise FileNotFoundError(
            f"no SigLIP tower weights: set VIDEOITG_VISION_TOWER or place them under {model_path}/vision_tower "
            f"(the reference downloads '{cfg.mm_vision_tower}' from the Hub, builder.py:163-165; no network here)")
    return cfg, sd, vsd


def _load_synthetic(spec: str) -> Tuple[ITGConfig, SD, SD]:
    from .. import synth
    parts = spec.split(":")
    kind = parts[1] if len(parts) > 1 and parts[1] else "tiny"
    seed = int(parts[2]) if len(parts) > 2 else 0
    cfg = {"tiny": tiny_config, "full": ITGConfig}[kind]()
    return cfg, synth.synth_llm_state_dict(cfg, seed), synth.synth_vision_state_dict(cfg, seed)


def load_pretrained_model(model_path, model_base, model_name, load_8bit=False, load_4bit=False, device_map="auto",
                          device="cuda", use_flash_attn=False, **kwargs):
    """-> (tokenizer, model, image_processor, context_len)   (builder.py:42,176)"""
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes 8/4-bit loading is not part of the selector path")
    if model_base is not None:
        raise NotImplementedError("LoRA / projector-only checkpoints (model_base) are not part of the selector path")
    dtype = kwargs.pop("torch_dtype", torch.float16)                       # builder.py:59
    operand_split = kwargs.pop("operand_split", None)                     # (this build's precision tier: EagleQwenG.operand_split)
    if device_map in (None, "auto"):
        dev = device if device != "cuda" else "cuda"
    elif isinstance(device_map, dict):
        dev = next(iter(device_map.values()))
    else:
        dev = device_map
    # use_flash_attn: attention here is always the fused non-causal HIP kernel; the flag is accepted and ignored.
    if str(model_path).startswith("synthetic"):
        cfg, sd, vsd = _load_synthetic(str(model_path))
        tokenizer = ByteTokenizer(cfg.vocab_size)
    else:
        cfg, sd, vsd = _load_directory(model_path)
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(model_path)              # builder.py:118
    model = EagleQwenG(cfg, sd, vsd, torch_dtype=dtype, device=dev)
    if operand_split is not None:
        model.operand_split = bool(operand_split)
    # builder.py:155-161 — special tokens (the embedding table is not grown: the ITG prompt never uses them)
    if getattr(cfg, "mm_use_im_patch_token", True):
        tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
    if getattr(cfg, "mm_use_im_start_end", False):
        tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
    image_processor = model.get_vision_tower().image_processor
    context_len = cfg.max_sequence_length if cfg.max_sequence_length else 2048   # builder.py:171-174
    return tokenizer, model, image_processor, context_len
