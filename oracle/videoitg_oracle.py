"""ORACLE — test infrastructure, NOT product code.

CPU (PyTorch fp32) restatement of the VideoITG frame-selector forward, rows A7-A13 of SURVEY.md §8(a).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product path (``videoitg_amd``) never does and fails loudly without its HIP library.

Parity status: the reference holds no tests/golden vectors for this path and its published artefacts are
missing blobs, so this oracle is pinned against outputs of the REFERENCE ITSELF run
in the build container: ``tests/golden/make_golden.py`` imports ``/root/reference``'s ``EagleQwenG`` (with
the harness-side shims of SURVEY.md Appendix B), runs it on seeded synthetic weights and commits
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` requires this file to match those to <=2e-5.

The arithmetic of the reference lives in un-vendored third-party wheels: transformers==4.47.1
(``models/siglip/modeling_siglip.py``, ``models/qwen2/modeling_qwen2.py``; /root/reference/requirements.txt:18)
and flash-attn==2.4.2 (README.md:144).  Their published algorithms are restated below and each function
cites the reference call site that reaches them.

All tensors are fp32 on CPU.  Weights come in as ``dict[str, Tensor]`` with the checkpoint key names of
SURVEY.md §8(b).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

IGNORE_INDEX = -100        # /root/reference/eagle/constants.py:9
IMAGE_TOKEN_INDEX = -200   # /root/reference/eagle/constants.py:10

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------------------------
# A7  SigLIP vision tower   (reference call site: eagle/model/multimodal_encoder/clip_encoder.py:123-154
#                            -> [3P] transformers SiglipVisionModel(output_hidden_states=True))
# ----------------------------------------------------------------------------------------------
def siglip_pos_embedding(vsd: SD, grid: int) -> torch.Tensor:
    """Learned position table [G0*G0, Dv]; bicubic-resampled to grid x grid when the input side differs
    from the native one ([3P] SiglipVisionEmbeddings.interpolate_pos_encoding: bicubic, align_corners=False).
    The reference always runs at the native size (SURVEY.md §0 discrepancy 1)."""
    pos = vsd["vision_model.embeddings.position_embedding.weight"]
    g0 = int(round(math.sqrt(pos.shape[0])))
    if g0 == grid:
        return pos
    p = pos.reshape(1, g0, g0, -1).permute(0, 3, 1, 2)
    p = F.interpolate(p, size=(grid, grid), mode="bicubic", align_corners=False)
    return p.permute(0, 2, 3, 1).reshape(grid * grid, -1)


def siglip_embed(pixel_values: torch.Tensor, vsd: SD, patch: int) -> torch.Tensor:
    """conv2d(3->Dv, k=patch, s=patch, valid) + bias, flatten row-major over the patch grid, + pos-emb.
    pixel_values [T,3,R,R] -> [T,P,Dv]."""
    w = vsd["vision_model.embeddings.patch_embedding.weight"]
    b = vsd["vision_model.embeddings.patch_embedding.bias"]
    x = F.conv2d(pixel_values, w, b, stride=patch)            # [T,Dv,G,G]
    T, Dv, G, _ = x.shape
    x = x.flatten(2).transpose(1, 2)                          # [T,P,Dv]
    return x + siglip_pos_embedding(vsd, G).unsqueeze(0)


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    """``gelu_pytorch_tanh`` (SigLIP MLP activation)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def siglip_layer(x: torch.Tensor, vsd: SD, i: int, num_heads: int, eps: float) -> torch.Tensor:
    """Pre-LN encoder layer: LN -> MHA (non-causal, scale hd^-1/2, fp32 softmax) -> +res -> LN -> fc1 ->
    GELU-tanh -> fc2 -> +res."""
    p = f"vision_model.encoder.layers.{i}."
    T, P, Dv = x.shape
    hd = Dv // num_heads
    h = F.layer_norm(x, (Dv,), vsd[p + "layer_norm1.weight"], vsd[p + "layer_norm1.bias"], eps)
    q = F.linear(h, vsd[p + "self_attn.q_proj.weight"], vsd[p + "self_attn.q_proj.bias"])
    k = F.linear(h, vsd[p + "self_attn.k_proj.weight"], vsd[p + "self_attn.k_proj.bias"])
    v = F.linear(h, vsd[p + "self_attn.v_proj.weight"], vsd[p + "self_attn.v_proj.bias"])
    q = q.view(T, P, num_heads, hd).transpose(1, 2)
    k = k.view(T, P, num_heads, hd).transpose(1, 2)
    v = v.view(T, P, num_heads, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
    a = torch.softmax(s, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(T, P, Dv)
    o = F.linear(o, vsd[p + "self_attn.out_proj.weight"], vsd[p + "self_attn.out_proj.bias"])
    x = x + o
    h = F.layer_norm(x, (Dv,), vsd[p + "layer_norm2.weight"], vsd[p + "layer_norm2.bias"], eps)
    h = F.linear(h, vsd[p + "mlp.fc1.weight"], vsd[p + "mlp.fc1.bias"])
    h = gelu_tanh(h)
    h = F.linear(h, vsd[p + "mlp.fc2.weight"], vsd[p + "mlp.fc2.bias"])
    return x + h


def siglip_tower(pixel_values: torch.Tensor, vsd: SD, *, patch: int, num_heads: int, num_layers: int,
                 select_layer: int = -2, eps: float = 1e-6, frame_chunk: int = 64) -> torch.Tensor:
    """``hidden_states[select_layer]`` of the tower (clip_encoder.py:123-129): hidden_states[i] is the
    output after i layers, so -2 is the output of layer L-1; layer L, post_layernorm and the pooling head
    never influence it and are not computed.  Frames are independent, so chunking (the reference uses
    chunks of 384, clip_encoder.py:140-149) does not change values."""
    n_run = num_layers + 1 + select_layer if select_layer < 0 else select_layer
    assert 0 <= n_run <= num_layers
    outs = []
    for s in range(0, pixel_values.shape[0], frame_chunk):
        x = siglip_embed(pixel_values[s:s + frame_chunk], vsd, patch)
        for i in range(n_run):
            x = siglip_layer(x, vsd, i, num_heads, eps)
        outs.append(x)
    return torch.cat(outs, 0)


# ----------------------------------------------------------------------------------------------
# A9  seq_mlp projector   (eagle/model/multimodal_projector/mlp_proj.py:39-75)
# ----------------------------------------------------------------------------------------------
def pooled_grid(num_frames: int, grid: int, vision_token_num: int) -> int:
    """mlp_proj.py:48,54 (eval branch)."""
    return min(math.floor((vision_token_num / num_frames) ** 0.5), grid)


def bilinear_pool(feat: torch.Tensor, hw: int) -> torch.Tensor:
    """[T,P,C] -> [T,hw*hw,C]: F.interpolate(bilinear, align_corners=False, no antialias) over the G x G
    token grid (mlp_proj.py:57-68).  The T>512 chunk branch (mlp_proj.py:57-64) is value-identical
    because frames are independent."""
    T, P, C = feat.shape
    g = int(P ** 0.5)
    if hw >= g:
        return feat
    x = feat.view(T, g, g, C).permute(0, 3, 1, 2)
    x = F.interpolate(x, size=(hw, hw), mode="bilinear", align_corners=False)
    return x.permute(0, 2, 3, 1).reshape(T, hw * hw, C)


def mlp_projector(feat: torch.Tensor, sd: SD, vision_token_num: int, hw: Optional[int] = None) -> torch.Tensor:
    """One video: [T,P,Dv] -> [T*HW*HW, D]: pool, Linear -> GELU(erf) -> Linear (mlp_proj.py:31-35,72-73).  hw: the grid to pool to when it
    is not the eval rule's (mlp_proj.py:51-52: training mode draws it at random; the caller passes what was drawn)."""
    T, P, _ = feat.shape
    hw = pooled_grid(T, int(P ** 0.5), vision_token_num) if hw is None else min(hw, int(P ** 0.5))
    x = bilinear_pool(feat, hw)
    pre = "model.mm_projector.out_projection."
    x = F.linear(x, sd[pre + "0.weight"], sd[pre + "0.bias"])
    x = F.gelu(x)  # exact erf
    x = F.linear(x, sd[pre + "2.weight"], sd[pre + "2.bias"])
    return x.flatten(0, 1)


# ----------------------------------------------------------------------------------------------
# A10  multimodal splice   (eagle/model/eagle_archv1.py:211-400)
# ----------------------------------------------------------------------------------------------
def splice(input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], visual: Sequence[torch.Tensor],
           sd: SD, *, max_length: Optional[int] = None, padding_side: str = "right"
           ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """ids [B,L] (+mask) and one projected visual block per <image> token ->
    (inputs_embeds [B,N,D], attention_mask bool [B,N], token_types int64 [B,N]).
    token types: 1/2 text (labels are None at inference -> every text token is IGNORE -> 1),
    3 image, 4 pad (eagle_archv1.py:277,356,398)."""
    emb = sd["model.embed_tokens.weight"]
    B = input_ids.shape[0]
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
    attention_mask = attention_mask.bool()
    rows, types = [], []
    img = 0
    for b in range(B):
        ids = input_ids[b][attention_mask[b]]                      # eagle_archv1.py:272
        pos = [-1] + torch.where(ids == IMAGE_TOKEN_INDEX)[0].tolist() + [ids.shape[0]]
        parts, tt = [], []
        n_img = len(pos) - 2
        if n_img == 0:                                             # eagle_archv1.py:282-294
            parts.append(emb[ids])
            tt.append(torch.full((ids.shape[0],), 1, dtype=torch.long))
            img += 1
        for i in range(n_img + 1 if n_img else 0):
            seg = ids[pos[i] + 1:pos[i + 1]]
            parts.append(emb[seg])
            tt.append(torch.full((seg.shape[0],), 1, dtype=torch.long))
            if i < n_img:
                parts.append(visual[img])
                tt.append(torch.full((visual[img].shape[0],), 3, dtype=torch.long))
                img += 1
        row, t = torch.cat(parts), torch.cat(tt)
        if max_length is not None:                                 # eagle_archv1.py:341-345
            row, t = row[:max_length], t[:max_length]
        rows.append(row)
        types.append(t)
    N = max(r.shape[0] for r in rows)
    D = emb.shape[1]
    out = torch.zeros(B, N, D)
    mask = torch.zeros(B, N, dtype=torch.bool)
    tts = torch.full((B, N), 4, dtype=torch.long)
    for b, (row, t) in enumerate(zip(rows, types)):
        n = row.shape[0]
        sl = slice(N - n, N) if padding_side == "left" else slice(0, n)   # eagle_archv1.py:361-381
        out[b, sl] = row
        mask[b, sl] = True
        tts[b, sl] = t
    return out, mask, tts


# ----------------------------------------------------------------------------------------------
# A11  Qwen2 decoder stack, bidirectional   (eagle/model/language_model/grounding_qwen2.py:45-66,135-145
#       -> [3P] Qwen2Model / Qwen2DecoderLayer / flash-attn with is_causal=False)
# ----------------------------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """[3P] Qwen2RMSNorm: fp32 variance, x * rsqrt(var + eps) * w."""
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def rope_tables(n: int, head_dim: int, theta: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """[3P] Qwen2RotaryEmbedding: inv_freq = theta^(-2i/hd), emb = cat(freqs, freqs); positions 0..n-1
    (position_ids None -> arange, eagle_archv1.py:396-397)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.outer(torch.arange(n, dtype=torch.float32), inv)
    emb = torch.cat([fr, fr], -1)
    return emb.cos(), emb.sin()


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


def attention_noncausal(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, key_mask: Optional[torch.Tensor],
                        q_chunk: int = 1024) -> torch.Tensor:
    """q [B,Hq,N,hd], k/v [B,Hkv,N,hd] (GQA: query head h uses kv head h // (Hq/Hkv)); full bidirectional
    softmax over all unmasked keys, fp32.  key_mask bool [B,N] (True = real token)."""
    B, Hq, N, hd = q.shape
    rep = Hq // k.shape[1]
    k = k.repeat_interleave(rep, 1)
    v = v.repeat_interleave(rep, 1)
    scale = hd ** -0.5
    out = torch.empty_like(q)
    bias = None
    if key_mask is not None and not bool(key_mask.all()):
        bias = torch.zeros(B, 1, 1, N).masked_fill(~key_mask[:, None, None, :], float("-inf"))
    for s in range(0, N, q_chunk):
        sc = torch.matmul(q[:, :, s:s + q_chunk], k.transpose(-1, -2)) * scale
        if bias is not None:
            sc = sc + bias
        out[:, :, s:s + q_chunk] = torch.matmul(torch.softmax(sc, -1), v)
    return out


def qwen2_layer(x: torch.Tensor, sd: SD, i: int, *, num_heads: int, num_kv_heads: int, eps: float,
                cos: torch.Tensor, sin: torch.Tensor, key_mask: Optional[torch.Tensor]) -> torch.Tensor:
    p = f"model.layers.{i}."
    B, N, D = x.shape
    hd = D // num_heads
    h = rms_norm(x, sd[p + "input_layernorm.weight"], eps)
    q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"])
    k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
    v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
    q = q.view(B, N, num_heads, hd).transpose(1, 2)
    k = k.view(B, N, num_kv_heads, hd).transpose(1, 2)
    v = v.view(B, N, num_kv_heads, hd).transpose(1, 2)
    q = q * cos + rotate_half(q) * sin
    k = k * cos + rotate_half(k) * sin
    o = attention_noncausal(q, k, v, key_mask).transpose(1, 2).reshape(B, N, D)
    x = x + F.linear(o, sd[p + "self_attn.o_proj.weight"])
    h = rms_norm(x, sd[p + "post_attention_layernorm.weight"], eps)
    g = F.linear(h, sd[p + "mlp.gate_proj.weight"])
    u = F.linear(h, sd[p + "mlp.up_proj.weight"])
    return x + F.linear(F.silu(g) * u, sd[p + "mlp.down_proj.weight"])


def qwen2_stack(x: torch.Tensor, sd: SD, *, num_layers: int, num_heads: int, num_kv_heads: int,
                eps: float, theta: float, key_mask: Optional[torch.Tensor] = None,
                collect: Optional[list] = None) -> torch.Tensor:
    B, N, D = x.shape
    cos, sin = rope_tables(N, D // num_heads, theta)
    for i in range(num_layers):
        x = qwen2_layer(x, sd, i, num_heads=num_heads, num_kv_heads=num_kv_heads, eps=eps, cos=cos, sin=sin,
                        key_mask=key_mask)
        if collect is not None:
            collect.append(x)
    return rms_norm(x, sd["model.norm.weight"], eps)


# ----------------------------------------------------------------------------------------------
# A12  score head   (grounding_qwen2.py:146-156)
# ----------------------------------------------------------------------------------------------
def score_head(hidden: torch.Tensor, token_types: torch.Tensor, frames_per_video: Sequence[int], sd: SD
               ) -> List[torch.Tensor]:
    """One [T,1] tensor per SAMPLE b, with T = images[b].shape[0] (grounding_qwen2.py:149-156 indexes `images` by the batch row, not by
    the <image> occurrence): a sample with several <image> tokens pools ALL its visual rows in images[b].shape[0] equal groups, and a
    sample without any gives the mean over an empty group = NaN (tests/golden/helpers.json "forward_errors")."""
    out = []
    for b in range(hidden.shape[0]):
        T = frames_per_video[b]
        h = hidden[b][token_types[b] == 3].reshape(T, -1, hidden.shape[-1]).mean(1)
        out.append(F.linear(h, sd["out_proj.weight"], sd["out_proj.bias"]))
    return out


# ----------------------------------------------------------------------------------------------
# A13  post-processing   (infer.py:72-79, lmms_eval/models/videoitg.py:302-308)
# ----------------------------------------------------------------------------------------------
def rank_frames(logits: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """sigmoid -> sort descending.  The reference's torch.sort is unstable on ties;
    oracle and product both define ties as "lower frame position first" (stable)."""
    s = torch.sigmoid(logits.reshape(-1).float())
    order = torch.sort(s, descending=True, stable=True).indices
    return s[order], order


def topk_selection(logits: torch.Tensor, frame_idx: Sequence[int], k: int) -> List[int]:
    """infer.py:74-79: first k of the descending order mapped through frame_idx, then ascending."""
    _, order = rank_frames(logits)
    return sorted(frame_idx[i] for i in order[:k].tolist())


# ----------------------------------------------------------------------------------------------
# whole path
# ----------------------------------------------------------------------------------------------
def forward(cfg, sd: SD, vsd: SD, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor],
            images: Sequence[torch.Tensor], *, return_intermediates: bool = False, pooled_grids: Optional[Sequence[int]] = None):
    """EagleQwenG.forward (grounding_qwen2.py:86-181) for inference: returns ``logits`` = list of [T_b,1].
    ``cfg`` is any object with the attribute names of ``videoitg_amd.config.ITGConfig``."""
    v = cfg.vision
    images = [im if im.dim() == 4 else im.unsqueeze(0) for im in images]
    feats = siglip_tower(torch.cat(list(images), 0).float(), vsd, patch=v.patch_size,
                         num_heads=v.num_attention_heads, num_layers=v.num_hidden_layers,
                         select_layer=cfg.mm_vision_select_layer, eps=v.layer_norm_eps)
    per_video = torch.split(feats, [im.shape[0] for im in images], 0)
    visual = [mlp_projector(f, sd, cfg.vision_token_num, None if pooled_grids is None else pooled_grids[i]) for i, f in enumerate(per_video)]
    x, mask, tts = splice(input_ids, attention_mask, visual, sd,
                          max_length=cfg.tokenizer_model_max_length, padding_side=cfg.tokenizer_padding_side)
    layers: list = []
    h = qwen2_stack(x, sd, num_layers=cfg.num_hidden_layers, num_heads=cfg.num_attention_heads,
                    num_kv_heads=cfg.num_key_value_heads, eps=cfg.rms_norm_eps, theta=cfg.rope_theta,
                    key_mask=mask, collect=layers if return_intermediates else None)
    logits = score_head(h, tts, [im.shape[0] for im in images], sd)
    if return_intermediates:
        return logits, dict(vit=feats, visual=visual, inputs_embeds=x, token_types=tts, mask=mask,
                            layers=layers, final=h)
    return logits


def grounding_loss(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """BCE-with-logits with the capped positive weight of /root/reference/eagle/model/language_model/grounding_qwen2.py:159-169:
    pos_weight = min(5, sqrt(#neg / max(1, #pos))); torch's formula (1-y) x + (1 + (pw-1) y) softplus(-x), mean, fp32."""
    x = logits.float().view(-1)
    y = labels.float().view(-1)
    pos = y.sum()
    neg = y.numel() - pos
    pw = torch.minimum(torch.tensor(5.0), torch.sqrt(neg / torch.clamp(pos, min=1.0)))
    sp = torch.log1p(torch.exp(-x.abs())) + torch.clamp(-x, min=0.0)
    return ((1.0 - y) * x + (1.0 + (pw - 1.0) * y) * sp).mean()
