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


def write_hf_checkpoint(cfg: ITGConfig, out_dir: str, tower_dir: str, seed: int = 0, device="cpu", shard_bytes: int = 4 << 30,
                        dtype: torch.dtype = torch.float16, plan: "synth.HeavyTailPlan | None" = None) -> dict:
    """Write the counter-generated model as the files a released checkpoint consists of (the factory's first contact with a
    full-size checkpoint LAYOUT, since nvidia/VideoITG-8B itself is unreachable offline):

    * `out_dir`: config.json (cfg.to_dict() + `mm_vision_tower` -> tower_dir), model-0000i-of-0000N.safetensors + model.safetensors.index.json with
      the LLM / projector / head keys of SURVEY.md 8(b), 16-bit tensors (the values are 16-bit exact, so nothing is lost), a stand-in fast tokenizer
      (tokenizer.json + tokenizer_config.json: AutoTokenizer.from_pretrained must work, as at /root/reference/eagle/model/builder.py:118);
    * `tower_dir`: the SigLIP repo as the Hub carries it - config.json with a `vision_config`, ONE model.safetensors holding `vision_model.*` for ALL
      layers (the 27th is never run), `vision_model.post_layernorm.*`, the attention-pooling `vision_model.head.*`, a (truncated) `text_model.*` tower
      of the same layer shapes, `logit_scale` / `logit_bias` - everything load_pretrained_model has to ignore by name.

    Tensors are produced one layer at a time on `device` and moved to the host as `dtype`.  -> {"bytes": ..., "shards": ..., "tensors": ...}"""
    import json
    import os
    from safetensors.torch import save_file
    os.makedirs(out_dir, exist_ok=True)
    os.makedirs(tower_dir, exist_ok=True)
    dev = torch.device(device)

    def draw(specs):
        return {k: t.to(dtype).cpu().contiguous() for k, t in synth.counter_state_dict(specs, seed, dev, plan).items()}
    # ---- tower --------------------------------------------------------------------------------------------------------------------------------
    v = cfg.vision
    tw = draw(list(synth.vision_param_specs(cfg)))
    Dv, Fv = v.hidden_size, v.intermediate_size
    extra = [("vision_model.post_layernorm.weight", (Dv,), 1.0, 0.02), ("vision_model.post_layernorm.bias", (Dv,), 0.0, 0.02),
             ("vision_model.head.probe", (1, 1, Dv), 0.0, 0.02), ("vision_model.head.attention.in_proj_weight", (3 * Dv, Dv), 0.0, Dv ** -0.5),
             ("vision_model.head.attention.in_proj_bias", (3 * Dv,), 0.0, 0.02), ("vision_model.head.attention.out_proj.weight", (Dv, Dv), 0.0, Dv ** -0.5),
             ("vision_model.head.attention.out_proj.bias", (Dv,), 0.0, 0.02), ("vision_model.head.layernorm.weight", (Dv,), 1.0, 0.02),
             ("vision_model.head.layernorm.bias", (Dv,), 0.0, 0.02), ("vision_model.head.mlp.fc1.weight", (Fv, Dv), 0.0, Dv ** -0.5),
             ("vision_model.head.mlp.fc1.bias", (Fv,), 0.0, 0.02), ("vision_model.head.mlp.fc2.weight", (Dv, Fv), 0.0, Fv ** -0.5),
             ("vision_model.head.mlp.fc2.bias", (Dv,), 0.0, 0.02), ("logit_scale", (1,), 0.0, 1.0), ("logit_bias", (1,), 0.0, 1.0),
             ("text_model.embeddings.token_embedding.weight", (256, Dv), 0.0, 0.02), ("text_model.embeddings.position_embedding.weight", (64, Dv), 0.0, 0.02),
             ("text_model.final_layer_norm.weight", (Dv,), 1.0, 0.02), ("text_model.final_layer_norm.bias", (Dv,), 0.0, 0.02),
             ("text_model.head.weight", (Dv, Dv), 0.0, Dv ** -0.5), ("text_model.head.bias", (Dv,), 0.0, 0.02)]
    # two text layers with the VISION layers' shapes and poisoned values: if one leaked into the tower the logits would show it
    for i in range(min(2, v.num_hidden_layers)):
        for k, sh, m, sd_ in synth.vision_param_specs(cfg):
            if k.startswith(f"vision_model.encoder.layers.{i}."):
                extra.append(("text_model." + k[len("vision_model."):], sh, 1000.0, 1.0))
    tw.update({k: t.to(dtype).cpu().contiguous() for k, t in synth.counter_state_dict(extra, seed + 17, dev).items()})
    save_file(tw, os.path.join(tower_dir, "model.safetensors"), metadata={"format": "pt"})
    with open(os.path.join(tower_dir, "config.json"), "w") as f:
        json.dump({"model_type": "siglip", "vision_config": {k: getattr(v, k) for k in v.__dataclass_fields__}, "text_config": {"hidden_size": Dv}}, f)
    total = sum(t.numel() * t.element_size() for t in tw.values())
    n_tensors = len(tw)
    del tw
    # ---- LLM + projector + head, sharded ----------------------------------------------------------------------------------------------------------
    lspecs = list(synth.llm_param_specs(cfg))
    groups = [[s for s in lspecs if s[0] == "model.embed_tokens.weight"]]
    groups += [[s for s in lspecs if s[0].startswith(f"model.layers.{i}.")] for i in range(cfg.num_hidden_layers)]
    groups.append([s for s in lspecs if not s[0].startswith("model.layers.") and s[0] != "model.embed_tokens.weight"])
    shards, cur, cur_bytes = [], {}, 0
    for g in groups:
        part = draw(g)
        nbytes = sum(t.numel() * t.element_size() for t in part.values())
        if cur and cur_bytes + nbytes > shard_bytes:
            shards.append(cur)
            cur, cur_bytes = {}, 0
        cur.update(part)
        cur_bytes += nbytes
    if cur:
        shards.append(cur)
    weight_map = {}
    for i, sh in enumerate(shards):
        name = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, os.path.join(out_dir, name), metadata={"format": "pt"})
        weight_map.update({k: name for k in sh})
        total += sum(t.numel() * t.element_size() for t in sh.values())
        n_tensors += len(sh)
        shards[i] = None
    with open(os.path.join(out_dir, "model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f)
    d = cfg.to_dict()
    d.update(mm_vision_tower=os.path.abspath(tower_dir), architectures=["EagleQwenG"], model_type="eagle_qwen_g", torch_dtype=str(dtype).split(".")[-1])
    with open(os.path.join(out_dir, "config.json"), "w") as f:
        json.dump(d, f, indent=1)
    # ---- stand-in tokenizer files (word level over a tiny vocabulary: the parity tests pass token ids, not text) -------------------------------------
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {"<|endoftext|>": 0, "<unk>": 1, **{w: i + 2 for i, w in enumerate("which frame shows the answer to this question ? a b c d . ,".split())}}
    tk = Tokenizer(models.WordLevel(vocab=vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", pad_token="<|endoftext|>", eos_token="<|endoftext|>").save_pretrained(out_dir)
    return {"bytes": total, "shards": len(shards), "tensors": n_tensors}
