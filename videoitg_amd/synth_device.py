"""Random-init weights of the full architecture generated directly in HBM (benchmark / smoke use).

Same shapes, key names and per-tensor distributions as ``videoitg_amd.synth``, drawn with ``synth.counter_tensor`` - a
stateless integer hash of (seed, key, element index) that is BIT-IDENTICAL on CPU and GPU - one layer at a time (fp32
layer -> packed 16-bit layout -> fp32 freed), so a 7.6 B-parameter model is ready in seconds and the fp32 oracle can
reproduce exactly the same weights on a CPU (tests/golden/make_fullsize_golden.py; bench.py's `parity` block).  No
checkpoint is reachable offline.
"""
from __future__ import annotations

import torch

from . import synth
from .config import ITGConfig
from .engine import PackedWeights


def _draw(specs, device, seed, plan=None):
    return synth.counter_state_dict(specs, seed, device, plan)


def packed_random_weights(cfg: ITGConfig, dtype: torch.dtype, device, seed: int = 0, fp8: bool = False,
                          vit_dtype=None, plan: "synth.HeavyTailPlan | None" = None, vit_q_prescaled=None) -> PackedWeights:
    """`plan`: the heavy-tailed variant (synth.HeavyTailPlan; fixtures carrying a "heavy_tail" record, synth.fixture_plan);
    vit_q_prescaled=False: the q projection keeps the checkpoint's own 16-bit weights (what the operand-split tier needs)"""
    device = torch.device(device)
    pw = PackedWeights(cfg=cfg, dtype=dtype, device=device, fp8=fp8, vit_dtype=vit_dtype, vit_q_prescaled=vit_q_prescaled)
    vspecs = list(synth.vision_param_specs(cfg))
    lspecs = list(synth.llm_param_specs(cfg))
    pw.pack_vision_embeddings(_draw([s for s in vspecs if s[0].startswith("vision_model.embeddings.")], device, seed, plan))
    for i in range(cfg.num_vit_layers_needed()):
        pre = f"vision_model.encoder.layers.{i}."
        pw.pack_vision_layer(_draw([s for s in vspecs if s[0].startswith(pre)], device, seed, plan), i)
    pw.pack_projector_and_head(_draw([s for s in lspecs if not s[0].startswith("model.layers.")], device, seed, plan))
    for i in range(cfg.num_hidden_layers):
        pre = f"model.layers.{i}."
        pw.pack_llm_layer(_draw([s for s in lspecs if s[0].startswith(pre)], device, seed, plan), i)
    return pw
