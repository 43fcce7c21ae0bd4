"""Seeded synthetic weights / inputs in the reference's checkpoint key layout.

Real weights (``nvidia/VideoITG-8B``, ``google/siglip-so400m-patch14-384``) are unreachable offline
(SURVEY.md §8c), so parity and the benchmark run on random-init weights of the same architecture.
Every tensor is generated from its own ``torch.Generator`` seeded by (seed, crc32(key)), so tensors can
be produced one at a time, in any order, and are bit-identical wherever torch's CPU generator is.

Values are rounded so that they are exactly representable in BOTH bf16 and fp16 (bf16 rounding, then
anything below fp16's normal range is flushed to zero): the fp32 oracle, a bf16 device run and an fp16
device run all see the same weight values.

Key names follow SURVEY.md §8(b) "Checkpoint layout the loader must accept":
LLM/projector/head keys as in the ITG checkpoint, SigLIP keys as in the separate Hub repo
(``vision_model.*``; /root/reference/eagle/model/builder.py:163-165 loads it separately).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterator, Tuple

import torch

from .config import ITGConfig

_FP16_MIN_NORMAL = 2.0 ** -14


def round_16(t: torch.Tensor) -> torch.Tensor:
    """fp32 values exactly representable in bf16 and fp16."""
    t = t.to(torch.bfloat16).to(torch.float32)
    t = torch.where(t.abs() < _FP16_MIN_NORMAL, torch.zeros_like(t), t)
    return t.clamp_(-60000.0, 60000.0)


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1_000_003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))
    return g


def _normal(seed, key, shape, std, mean=0.0) -> torch.Tensor:
    t = torch.empty(shape, dtype=torch.float32)
    t.normal_(mean, std, generator=_gen(seed, key))
    return round_16(t)


# ---- counter-based generator: bit-identical on every device ------------------------------------------------------------
# torch's CPU and HIP generators draw different streams, so weights drawn on the GPU (fast) could not be reproduced by the
# fp32 oracle on a CPU.  The full-size parity fixtures (tests/golden/make_fullsize_golden.py) and bench.py's `parity` block
# need exactly that, so full-size weights come from a stateless integer hash of (seed, key, flat element index):
# 32-bit-valued arithmetic carried in int64 (no overflow anywhere), the four bytes of the hash summed (Irwin-Hall(4):
# mean 510, variance 4*(256^2-1)/12 = 21845) -> centred -> one fp32 multiply -> one fp32 add -> round_16.  Every step is an
# exactly defined integer or single IEEE operation, so CPU and GPU agree in every bit.
_M32 = 0xFFFFFFFF
_IH4_STD = 21845.0 ** 0.5
_CHUNK_CPU, _CHUNK_DEV = 1 << 20, 1 << 26      # cache-resident on a CPU (8x faster than one big pass), few launches on a GPU


def _mix32(x: torch.Tensor) -> torch.Tensor:
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & _M32
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & _M32
    return x ^ (x >> 16)


def key_seed(seed: int, key: str) -> int:
    return (seed * 1_000_003 + zlib.crc32(key.encode())) & _M32


def counter_values(seed: int, key: str, flat_index: torch.Tensor, mean: float, std: float) -> torch.Tensor:
    """fp32 values of tensor `key` at the given flat element indices (int64 tensor on any device)."""
    x = _mix32((_mix32(flat_index & _M32) + key_seed(seed, key)) & _M32)
    b = (x & 255) + ((x >> 8) & 255) + ((x >> 16) & 255) + (x >> 24)
    scale = torch.tensor(std / _IH4_STD, dtype=torch.float32).item()          # the fp32 value, as a Python float
    v = (b - 510).to(torch.float32) * scale
    if mean != 0.0:
        v = v + torch.tensor(mean, dtype=torch.float32).item()
    return round_16(v)


def counter_tensor(seed: int, key: str, shape, mean: float, std: float, device="cpu") -> torch.Tensor:
    """Whole tensor `key` (fp32, 16-bit exact), generated on `device` in bounded chunks."""
    n = 1
    for s in shape:
        n *= int(s)
    out = torch.empty(n, dtype=torch.float32, device=device)
    chunk = _CHUNK_CPU if torch.device(device).type == "cpu" else _CHUNK_DEV
    for s0 in range(0, n, chunk):
        idx = torch.arange(s0, min(n, s0 + chunk), dtype=torch.int64, device=device)
        out[s0:s0 + idx.numel()] = counter_values(seed, key, idx, mean, std)
    return out.reshape(tuple(shape))


def counter_rows(seed: int, key: str, shape, rows: torch.Tensor, mean: float, std: float) -> torch.Tensor:
    """Selected rows of the 2-D tensor `key` without generating the rest (embedding rows of a prompt)."""
    cols = int(shape[1])
    rows = rows.to(torch.int64)
    idx = rows[:, None] * cols + torch.arange(cols, dtype=torch.int64, device=rows.device)[None, :]
    return counter_values(seed, key, idx.reshape(-1), mean, std).reshape(rows.numel(), cols)


def counter_state_dict(specs, seed: int = 0, device="cpu", plan: "HeavyTailPlan | None" = None) -> Dict[str, torch.Tensor]:
    sd = {k: counter_tensor(seed, k, s, m, sd, device) for k, s, m, sd in specs}
    return plan.apply(sd) if plan is not None else sd


# ---- heavy-tailed variant: outlier channels planted the way trained towers / decoders carry them -----------------------
# The plain draw above is benign: N(0, 1/fan_in) weights keep every residual stream within +-10, and a 16-bit residual stream is
# never stressed.  Trained SigLIP / Qwen2 checkpoints are not like that: a handful of hidden channels carry "massive
# activations" two orders above the rest - a near-constant component written by a bias early in the stack plus a token-dependent
# one written by a few large rows of the MLP's output projection - and the norm gains hand the following projections inputs of
# ordinary size again (gain of a planted channel small, the others above one).  `HeavyTailPlan` turns a plain state dict into one
# with that anatomy, by EXACT operations only (power-of-two row / gain factors, constants that are 16-bit exact), so CPU oracle
# and GPU engine still share every bit:
#   tower   : fc2.weight[c, :] x row_gain in every layer, fc2.bias[c] = +-vit_bias in layers `vit_bias_layers`,
#             layer_norm{1,2}.weight of layer l: planted channels x vit_norm[l][1], every other channel x vit_norm[l][0]
#   decoder : down_proj.weight[c, :] x row_gain in every layer; the projector's last bias [c] and column c of the embedding table =
#             +-llm_bias (every token enters the decoder with the constant); input / post-attention RMSNorm gains of layer l x
#             llm_norm[l][0 | 1], the final norm x final_norm[0 | 1]
# The per-layer gain pairs are DATA (tests/golden/heavy_tail_plan.json, written by tests/golden/calibrate_heavy_tail.py from a small
# complete pass): powers of two chosen so that after each norm the ordinary channels have the unit scale of the plain draw and the
# planted ones ~4x that.  Residual streams reach 10^2 - 10^3 in the planted channels (checks.vit_absmax / llm_absmax of a fixture).
class HeavyTailPlan:
    def __init__(self, vit_channels, llm_channels, vit_norm, llm_norm, final_norm, row_gain=64.0, vit_bias=128.0,
                 vit_bias_layers=(1, 2), llm_bias=1024.0):
        self.vit_channels, self.llm_channels = tuple(int(c) for c in vit_channels), tuple(int(c) for c in llm_channels)
        self.vit_norm = [tuple(float(v) for v in p) for p in vit_norm]
        self.llm_norm = [tuple(float(v) for v in p) for p in llm_norm]
        self.final_norm = tuple(float(v) for v in final_norm)
        self.row_gain = float(row_gain)
        self.vit_bias, self.vit_bias_layers, self.llm_bias = float(vit_bias), tuple(vit_bias_layers), float(llm_bias)

    @staticmethod
    def channels(cfg: ITGConfig, seed: int = 7, n: int = 4):
        g = _gen(seed, "heavy_tail_channels")
        vc = sorted(torch.randperm(cfg.mm_hidden_size, generator=g)[:n].tolist())
        lc = sorted(torch.randperm(cfg.hidden_size, generator=g)[:n].tolist())
        return vc, lc

    def to_json(self) -> dict:
        return dict(vit_channels=list(self.vit_channels), llm_channels=list(self.llm_channels), vit_norm=[list(p) for p in self.vit_norm],
                    llm_norm=[list(p) for p in self.llm_norm], final_norm=list(self.final_norm), row_gain=self.row_gain,
                    vit_bias=self.vit_bias, vit_bias_layers=list(self.vit_bias_layers), llm_bias=self.llm_bias)

    @classmethod
    def from_json(cls, d: dict) -> "HeavyTailPlan":
        return cls(**d)

    @staticmethod
    def _signs(n, device):
        return torch.tensor([1.0 if j % 2 == 0 else -1.0 for j in range(n)], dtype=torch.float32, device=device)

    @staticmethod
    def _norm_gain(w, chans, pair):
        out = w * pair[0]
        idx = torch.tensor(chans, dtype=torch.long, device=w.device)
        out[idx] = w[idx] * pair[1]
        return out

    def apply(self, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """in place on the fp32 tensors of a (partial) state dict; keys it does not know pass through"""
        for k in list(sd):
            t = sd[k]
            if k.startswith("vision_model.encoder.layers."):
                layer = int(k.split(".")[3])
                idx = torch.tensor(self.vit_channels, dtype=torch.long, device=t.device)
                if k.endswith("mlp.fc2.weight"):
                    t[idx] = t[idx] * self.row_gain
                elif k.endswith("mlp.fc2.bias") and layer in self.vit_bias_layers:
                    t[idx] = self._signs(len(self.vit_channels), t.device) * self.vit_bias
                elif (k.endswith("layer_norm1.weight") or k.endswith("layer_norm2.weight")) and layer < len(self.vit_norm):
                    sd[k] = self._norm_gain(t, self.vit_channels, self.vit_norm[layer])
            elif k.startswith("model.layers."):
                layer = int(k.split(".")[2])
                idx = torch.tensor(self.llm_channels, dtype=torch.long, device=t.device)
                if k.endswith("mlp.down_proj.weight"):
                    t[idx] = t[idx] * self.row_gain
                elif k.endswith("layernorm.weight") and layer < len(self.llm_norm):
                    sd[k] = self._norm_gain(t, self.llm_channels, self.llm_norm[layer])
            elif k == "model.norm.weight":
                sd[k] = self._norm_gain(t, self.llm_channels, self.final_norm)
            elif k == "model.mm_projector.out_projection.2.bias":
                idx = torch.tensor(self.llm_channels, dtype=torch.long, device=t.device)
                t[idx] = self._signs(len(self.llm_channels), t.device) * self.llm_bias
            elif k == "model.embed_tokens.weight":            # full table, or the gathered rows of a prompt (oracle/fullsize.py)
                idx = torch.tensor(self.llm_channels, dtype=torch.long, device=t.device)
                t[:, idx] = self._signs(len(self.llm_channels), t.device) * self.llm_bias
        return sd


def fixture_plan(rec: dict):
    """the HeavyTailPlan of a complete-pass fixture record, or None for the plain draw"""
    return HeavyTailPlan.from_json(rec["heavy_tail"]) if rec.get("heavy_tail") else None


def llm_param_specs(cfg: ITGConfig) -> Iterator[Tuple[str, tuple, float, float]]:
    """(key, shape, mean, std) for every LLM / projector / head tensor."""
    D, F = cfg.hidden_size, cfg.intermediate_size
    hd = cfg.head_dim
    nq, nkv = cfg.num_attention_heads * hd, cfg.num_key_value_heads * hd
    yield "model.embed_tokens.weight", (cfg.vocab_size, D), 0.0, 1.0
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        yield p + "input_layernorm.weight", (D,), 1.0, 0.02
        yield p + "self_attn.q_proj.weight", (nq, D), 0.0, D ** -0.5
        yield p + "self_attn.q_proj.bias", (nq,), 0.0, 0.02
        yield p + "self_attn.k_proj.weight", (nkv, D), 0.0, D ** -0.5
        yield p + "self_attn.k_proj.bias", (nkv,), 0.0, 0.02
        yield p + "self_attn.v_proj.weight", (nkv, D), 0.0, D ** -0.5
        yield p + "self_attn.v_proj.bias", (nkv,), 0.0, 0.02
        yield p + "self_attn.o_proj.weight", (D, nq), 0.0, nq ** -0.5
        yield p + "post_attention_layernorm.weight", (D,), 1.0, 0.02
        yield p + "mlp.gate_proj.weight", (F, D), 0.0, D ** -0.5
        yield p + "mlp.up_proj.weight", (F, D), 0.0, D ** -0.5
        yield p + "mlp.down_proj.weight", (D, F), 0.0, F ** -0.5
    yield "model.norm.weight", (D,), 1.0, 0.02
    Dv = cfg.mm_hidden_size
    yield "model.mm_projector.out_projection.0.weight", (D, Dv), 0.0, Dv ** -0.5
    yield "model.mm_projector.out_projection.0.bias", (D,), 0.0, 0.02
    yield "model.mm_projector.out_projection.2.weight", (D, D), 0.0, D ** -0.5
    yield "model.mm_projector.out_projection.2.bias", (D,), 0.0, 0.02
    # score head: scaled up so that the logits are spread out (SURVEY.md §7 hard-part 2)
    yield "out_proj.weight", (1, D), 0.0, 4.0 * D ** -0.5
    yield "out_proj.bias", (1,), 0.0, 0.02


def vision_param_specs(cfg: ITGConfig) -> Iterator[Tuple[str, tuple, float, float]]:
    v = cfg.vision
    Dv, Fv, ps, C = v.hidden_size, v.intermediate_size, v.patch_size, v.num_channels
    e = "vision_model.embeddings."
    yield e + "patch_embedding.weight", (Dv, C, ps, ps), 0.0, (C * ps * ps) ** -0.5
    yield e + "patch_embedding.bias", (Dv,), 0.0, 0.02
    yield e + "position_embedding.weight", (v.num_patches, Dv), 0.0, 0.02
    for i in range(v.num_hidden_layers):
        p = f"vision_model.encoder.layers.{i}."
        yield p + "layer_norm1.weight", (Dv,), 1.0, 0.02
        yield p + "layer_norm1.bias", (Dv,), 0.0, 0.02
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            yield p + f"self_attn.{n}.weight", (Dv, Dv), 0.0, Dv ** -0.5
            yield p + f"self_attn.{n}.bias", (Dv,), 0.0, 0.02
        yield p + "layer_norm2.weight", (Dv,), 1.0, 0.02
        yield p + "layer_norm2.bias", (Dv,), 0.0, 0.02
        yield p + "mlp.fc1.weight", (Fv, Dv), 0.0, Dv ** -0.5
        yield p + "mlp.fc1.bias", (Fv,), 0.0, 0.02
        yield p + "mlp.fc2.weight", (Dv, Fv), 0.0, Fv ** -0.5
        yield p + "mlp.fc2.bias", (Dv,), 0.0, 0.02


def synth_tensor(seed: int, key: str, shape, mean: float, std: float) -> torch.Tensor:
    return _normal(seed, key, shape, std, mean)


def synth_llm_state_dict(cfg: ITGConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: _normal(seed, k, s, sd, m) for k, s, m, sd in llm_param_specs(cfg)}


def synth_vision_state_dict(cfg: ITGConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: _normal(seed, k, s, sd, m) for k, s, m, sd in vision_param_specs(cfg)}


def synth_pixel_values(num_frames: int, image_size: int, seed: int = 1234) -> torch.Tensor:
    """``fp32 [T,3,R,R]`` in [-1,1] (what SiglipImageProcessor emits), low-pass
    filtered so patches are not white noise, 16-bit exact."""
    g = _gen(seed, f"pixels{num_frames}x{image_size}")
    coarse = torch.rand((num_frames, 3, (image_size + 7) // 8, (image_size + 7) // 8), generator=g)
    x = torch.nn.functional.interpolate(coarse, size=(image_size, image_size), mode="bilinear",
                                        align_corners=False)
    x = x + 0.1 * torch.rand(x.shape, generator=g)
    x = (x / 1.1) * 2.0 - 1.0
    return round_16(x)


def synth_prompt_ids(length: int, vocab_size: int, seed: int = 4321) -> torch.Tensor:
    g = _gen(seed, f"prompt{length}")
    hi = min(vocab_size, 151_643)
    return torch.randint(0, hi, (length,), generator=g, dtype=torch.long)


def workload_inputs(cfg: ITGConfig, T: int, R: int, prompt_len: int, pix_seed: int = 1234):
    """(input_ids int64 [1, 1+L], pixel_values fp32 [T,3,R,R]) - the synthetic inputs of SURVEY.md §8(d): `<image>` followed by
    `prompt_len` seeded token ids; low-pass-filtered pixels in [-1,1], exactly representable in bf16 and fp16.  bench.py, the GPU
    parity tests and the full-size oracle fixtures (tests/golden/full_*.json) all start from this one function."""
    ids = torch.cat([torch.tensor([-200]), synth_prompt_ids(prompt_len, cfg.vocab_size)]).unsqueeze(0)
    return ids, synth_pixel_values(T, R, seed=pix_seed)


def planted_frames(T: int, k: int = 32, plant_seed: int = 99) -> list:
    """the k frame indices (sorted) that carry the planted signal: a seeded draw without replacement"""
    g = _gen(plant_seed, f"planted{T}x{k}")
    return sorted(torch.randperm(T, generator=g)[:min(k, T)].tolist())


def planted_inputs(cfg: ITGConfig, T: int, R: int, prompt_len: int, pix_seed: int = 1234, k: int = 32, plant_seed: int = 99):
    """workload_inputs with a planted per-frame signal (hard part 2 iii): the k planted frames are bright
    (pixels * 0.5 + 0.5), all others dark (pixels * 0.5 - 0.5).  With random-init weights the scores of `workload_inputs` crowd together
    (rank-32/33 gap 1.5e-3 at 512 frames), so whether a lower-precision tier keeps the Top-32 SET is undecidable there; a global brightness
    step survives the 26-layer tower, the pooling and the 28-layer decoder as a logit offset of several units, which makes the oracle's
    Top-k = the planted set with a wide gap (recorded in the fixture).  -> (input_ids, pixel_values fp32 16-bit-exact, planted indices)"""
    ids, pix = workload_inputs(cfg, T, R, prompt_len, pix_seed)
    planted = planted_frames(T, k, plant_seed)
    mask = torch.zeros(T, dtype=torch.bool)
    mask[planted] = True
    off = torch.where(mask, 0.5, -0.5).view(T, 1, 1, 1)
    return ids, round_16(pix * 0.5 + off), planted


def fixture_inputs(cfg: ITGConfig, rec: dict):
    """(input_ids, pixel_values) of a complete-pass oracle fixture record (tests/golden/full_*.json)"""
    if rec.get("plant"):
        pl = rec["plant"]
        ids, pix, _ = planted_inputs(cfg, rec["frames"], rec["resolution"], rec["prompt_len"], rec["pixel_seed"], pl["k"], pl["seed"])
        return ids, pix
    return workload_inputs(cfg, rec["frames"], rec["resolution"], rec["prompt_len"], rec["pixel_seed"])


def synth_video_frames(seed: int, T: int, H: int, W: int):
    """uint8 [T,H,W,3] decoded-frame stand-in (numpy): noise blended with a smooth gradient, plus a full-contrast band so
    that the bicubic resize both interpolates mid-range values and overshoots into the [0, 255] clip."""
    import numpy as np
    rng = np.random.default_rng(seed)
    noise = rng.integers(0, 256, (T, H, W, 3), dtype=np.uint8).astype(np.int32)
    yy, xx = np.mgrid[0:H, 0:W]
    grad = ((xx * 255) // max(W - 1, 1) + (yy * 255) // max(H - 1, 1)) // 2
    out = (noise + 3 * grad[None, :, :, None]) // 4
    out[:, : H // 4] = noise[:, : H // 4]
    return out.astype(np.uint8)
