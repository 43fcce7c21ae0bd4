"""Random-init weights of the full architecture generated directly in HBM (benchmark / smoke use).

Same shapes, key names and per-tensor distributions as ``videoitg_amd.synth`` but drawn with the device generator,
one layer at a time (fp32 layer -> packed 16-bit layout -> fp32 freed), so a 7.6 B-parameter model is ready in
seconds.  NOT bit-identical to the CPU generator of ``synth`` — parity tests use ``synth``; the benchmark only needs
weights of the right architecture (no checkpoint is reachable offline).
"""
from __future__ import annotations

import zlib

import torch

from . import synth
from .config import ITGConfig
from .engine import PackedWeights


def _draw(specs, device, seed):
    out = {}
    for k, shape, mean, std in specs:
        g = torch.Generator(device=device)
        g.manual_seed((seed * 1_000_003 + zlib.crc32(k.encode())) % (2 ** 63 - 1))
        t = torch.empty(shape, dtype=torch.float32, device=device)
        t.normal_(mean, std, generator=g)
        out[k] = t
    return out


def packed_random_weights(cfg: ITGConfig, dtype: torch.dtype, device, seed: int = 0) -> PackedWeights:
    device = torch.device(device)
    pw = PackedWeights(cfg=cfg, dtype=dtype, device=device)
    vspecs = list(synth.vision_param_specs(cfg))
    lspecs = list(synth.llm_param_specs(cfg))
    pw.pack_vision_embeddings(_draw([s for s in vspecs if s[0].startswith("vision_model.embeddings.")], device, seed))
    for i in range(cfg.num_vit_layers_needed()):
        pre = f"vision_model.encoder.layers.{i}."
        pw.pack_vision_layer(_draw([s for s in vspecs if s[0].startswith(pre)], device, seed), i)
    pw.pack_projector_and_head(_draw([s for s in lspecs if not s[0].startswith("model.layers.")], device, seed))
    for i in range(cfg.num_hidden_layers):
        pre = f"model.layers.{i}."
        pw.pack_llm_layer(_draw([s for s in lspecs if s[0].startswith(pre)], device, seed), i)
    return pw
