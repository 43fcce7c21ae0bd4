"""ORACLE-SIDE numerics model — test infrastructure, NOT product code.

Two rounding models layered on the fp32 restatement (oracle/videoitg_oracle.py), used to derive the error budget
that DESIGN.md quotes and that tests/test_numerics_budget.py checks on CPU:

* ``mode="product"``: rounds to the 16-bit type exactly where libvideoitg_hip.so does (MFMA inputs only: norm
  outputs, fused-QKV output, RoPE output, softmax probabilities, attention output, GELU / SwiGLU output, pooled
  features, projector hidden); residual stream, accumulators, norm/softmax statistics stay fp32.
* ``mode="reference16"``: rounds after EVERY op like the reference's own inference does when the whole module is
  ``.half()`` (/root/reference/eagle/model/builder.py:59, infer.py:21): residual adds, norm outputs, GEMM outputs,
  activations all produce 16-bit tensors (matmuls accumulate in fp32 inside the BLAS kernel, flash-attn keeps
  softmax statistics in fp32 and P in 16 bit).

* ``mode="fp8"``: ``"product"`` plus the fp8 weight path of ``videoitg_amd/csrc/gemm_fp8.hip`` (BASELINE.json configs[4]): at the
  tower / decoder layer GEMMs (qkv, out, fc1, fc2; qkv, o, gate/up, down) the weight is quantised per OUTPUT CHANNEL and the
  activation per TOKEN to OCP e4m3fn (scale = amax * fp32(1/448), value * (1/scale) rounded to nearest even, clamp +-448), exactly
  the arithmetic of ``quant_rows_kernel`` / ``norm_quant_kernel``; norm outputs are quantised from their fp32 value (the fused
  kernel never rounds them to 16 bit), attention / GELU / SwiGLU outputs from their 16-bit value.  With ``fp8_mx`` (the product's
  default) the tower's GELU output and the decoder's SwiGLU output reach fc2 / down as e4m3 + OCP-MX block scales written by the producing
  GEMM's epilogue (``vitg_gemm_fp8_mx``): one
  power-of-two scale per 32 consecutive values, e = ceil(log2(amax / 448)), quantised from the fp32 value (``q8_mx_rows``).

Comparing each against the fp32 oracle gives (a) the rounding noise the product SHOULD show if its kernels are
otherwise exact and (b) the noise the reference's own fp16 run carries.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from . import videoitg_oracle as O


class Q:
    def __init__(self, dtype: Optional[torch.dtype], mode: str, fp8_mx: bool = True, exact=()):
        self.dtype, self.mode, self.fp8_mx = dtype, mode, fp8_mx
        # rounding sites carried at (hi + lo) split precision instead of one 16-bit value: modelled as exact (22 of fp32's 24 bits).
        # names: norm1 norm2 qkv p attn_out act pool proj (tools/split_site_model.py; the product's `operand_split` tier)
        self.exact = frozenset(exact)

    def mm(self, t, site=None):       # MFMA-input rounding (both modes)
        return t if (self.dtype is None or site in self.exact) else t.to(self.dtype).float()

    def every(self, t):    # op-output rounding (reference16 only)
        return self.mm(t) if self.mode == "reference16" else t

    # ---- fp8 weight path: operands of the layer GEMMs -------------------------------------------------------------------
    def a_norm(self, t, site=None):   # a norm output as GEMM operand
        return q8_rows(t) if self.mode == "fp8" else self.mm(t, site)

    def a_act(self, t):    # an already 16-bit-rounded activation as GEMM operand
        return q8_rows(t) if self.mode == "fp8" else t

    def a_gelu(self, t):   # the tower's GELU output (fp32 in) as fc2's operand
        if self.mode == "fp8":
            return q8_mx_rows(t) if self.fp8_mx else q8_rows(self.mm(t))
        return self.mm(t, "act")

    def a_swiglu(self, t):  # the decoder's silu(gate) * up (fp32 in) as down's operand
        if self.mode == "fp8":
            return q8_mx_rows(t) if self.fp8_mx else q8_rows(self.mm(t))
        return self.mm(t, "act")

    def w(self, t):        # a layer GEMM weight (rows = output channels)
        return q8_rows(t) if self.mode == "fp8" else t


_INV448 = torch.tensor(1.0 / 448.0, dtype=torch.float32)


def q8_rows(t: torch.Tensor) -> torch.Tensor:
    """per-row OCP e4m3fn quantise-dequantise with the kernel's arithmetic (fp32): scale = amax * fp32(1/448), q = RNE(x * (1/scale))"""
    t = t.float()
    amax = t.abs().amax(-1, keepdim=True)
    sc = torch.where(amax > 0, amax * _INV448, torch.ones_like(amax))
    q = (t * (1.0 / sc)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()
    return q * sc


def q8_mx_rows(t: torch.Tensor) -> torch.Tensor:
    """OCP-MX e4m3 quantise-dequantise along the last dimension with the arithmetic of the MX epilogue of gemm_fp8.hip: blocks of 32,
    E8M0 exponent e = ceil(log2(amax * fp32(1/448))) taken from the float's bits (exponent field + 1 if any mantissa bit is set),
    e >= -126, q = RNE(x * 2^-e) as e4m3fn"""
    t = t.float()
    n = t.shape[-1]
    pad = (-n) % 32
    tb = F.pad(t, (0, pad)).reshape(*t.shape[:-1], -1, 32)
    amax = tb.abs().amax(-1, keepdim=True)
    bits = (amax * _INV448).view(torch.int32)
    e = ((bits >> 23) & 255) - 127 + ((bits & 0x7FFFFF) != 0).to(torch.int32)
    e = torch.clamp(e, min=-126)
    inv = torch.exp2(-e.float())
    q = (tb * inv).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * torch.exp2(e.float())
    return q.reshape(*t.shape[:-1], -1)[..., :n]


def _attn(q, k, v, scale, qz: Q, q_chunk=1024):
    rep = q.shape[1] // k.shape[1]
    k, v = k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1)
    out = torch.empty_like(q)
    for s in range(0, q.shape[2], q_chunk):
        sc = torch.matmul(q[:, :, s:s + q_chunk], k.transpose(-1, -2)) * scale
        m = sc.max(-1, keepdim=True).values
        p = torch.exp(sc - m)
        l = p.sum(-1, keepdim=True)                         # fp32 row sum of the unrounded p (as the kernel does)
        out[:, :, s:s + q_chunk] = torch.matmul(qz.mm(p, "p"), v) / l
    return out


def forward(cfg, sd, vsd, input_ids, attention_mask, images: Sequence[torch.Tensor], dtype=torch.float16,
            mode: str = "product", fp8_mx: bool = True, frame_chunk: Optional[int] = None, exact=()):
    """frame_chunk: run the tower on that many frames at a time (frames are independent; a 512-frame video's attention scores alone
    would be 34 GB in one piece) - same values, bounded memory (tools/heavy_tail_rounding_model.py at the headline size)"""
    qz = Q(dtype, mode, fp8_mx, exact)
    v = cfg.vision
    images = [im if im.dim() == 4 else im.unsqueeze(0) for im in images]
    pix = qz.mm(torch.cat(list(images), 0).float())
    if frame_chunk is not None and pix.shape[0] > frame_chunk:
        x = torch.cat([_tower(cfg, vsd, pix[s:s + frame_chunk], qz) for s in range(0, pix.shape[0], frame_chunk)], 0)
    else:
        x = _tower(cfg, vsd, pix, qz)
    return _after_tower(cfg, sd, x, input_ids, attention_mask, images, qz, mode)


def _tower(cfg, vsd, pix, qz: Q):
    v = cfg.vision
    # ---- tower
    w = vsd["vision_model.embeddings.patch_embedding.weight"]
    x = F.conv2d(pix, w, vsd["vision_model.embeddings.patch_embedding.bias"], stride=v.patch_size)
    T, Dv, G, _ = x.shape
    x = qz.every(x.flatten(2).transpose(1, 2) + O.siglip_pos_embedding(vsd, G).unsqueeze(0))
    H = v.num_attention_heads
    hd = Dv // H
    n_run = v.num_hidden_layers + 1 + cfg.mm_vision_select_layer
    for i in range(n_run):
        p = f"vision_model.encoder.layers.{i}."
        h = qz.a_norm(F.layer_norm(x, (Dv,), vsd[p + "layer_norm1.weight"], vsd[p + "layer_norm1.bias"], v.layer_norm_eps), "norm1")
        q, k, vv = (qz.mm(F.linear(h, qz.w(vsd[p + f"self_attn.{n}_proj.weight"]), vsd[p + f"self_attn.{n}_proj.bias"]), "qkv")
                    .view(T, -1, H, hd).transpose(1, 2) for n in "qkv")
        o = qz.mm(_attn(q, k, vv, hd ** -0.5, qz), "attn_out").transpose(1, 2).reshape(T, -1, Dv)
        x = qz.every(x + qz.every(F.linear(qz.a_act(o), qz.w(vsd[p + "self_attn.out_proj.weight"]), vsd[p + "self_attn.out_proj.bias"])))
        h = qz.a_norm(F.layer_norm(x, (Dv,), vsd[p + "layer_norm2.weight"], vsd[p + "layer_norm2.bias"], v.layer_norm_eps), "norm2")
        f = qz.a_gelu(O.gelu_tanh(qz.every(F.linear(h, qz.w(vsd[p + "mlp.fc1.weight"]), vsd[p + "mlp.fc1.bias"]))))
        x = qz.every(x + qz.every(F.linear(f, qz.w(vsd[p + "mlp.fc2.weight"]), vsd[p + "mlp.fc2.bias"])))
    return x


def _after_tower(cfg, sd, x, input_ids, attention_mask, images, qz: Q, mode: str):
    feats = torch.split(x, [im.shape[0] for im in images], 0)
    # ---- projector
    pre = "model.mm_projector.out_projection."
    visual = []
    for f in feats:
        hw = O.pooled_grid(f.shape[0], int(f.shape[1] ** 0.5), cfg.vision_token_num)
        pl = qz.mm(O.bilinear_pool(f, hw), "pool")
        g = qz.mm(F.gelu(qz.every(F.linear(pl, sd[pre + "0.weight"], sd[pre + "0.bias"]))), "proj")
        visual.append(qz.every(F.linear(g, sd[pre + "2.weight"], sd[pre + "2.bias"])).flatten(0, 1))
    xs, mask, tts = O.splice(input_ids, attention_mask, visual, sd, max_length=cfg.tokenizer_model_max_length,
                             padding_side=cfg.tokenizer_padding_side)
    # ---- LLM, one sample at a time on its unpadded rows (value-identical to the masked batch)
    nq, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    D = cfg.hidden_size
    hd = D // nq
    logits = []
    for b, im in enumerate(images):
        x = xs[b][mask[b]].unsqueeze(0)
        N = x.shape[1]
        cos, sin = O.rope_tables(N, hd, cfg.rope_theta)
        if mode == "reference16":
            cos, sin = qz.mm(cos), qz.mm(sin)            # HF casts the tables to the activation dtype
        for i in range(cfg.num_hidden_layers):
            p = f"model.layers.{i}."
            h = qz.a_norm(O.rms_norm(x, sd[p + "input_layernorm.weight"], cfg.rms_norm_eps), "norm1")
            q = qz.mm(F.linear(h, qz.w(sd[p + "self_attn.q_proj.weight"]), sd[p + "self_attn.q_proj.bias"]), "qkv").view(1, N, nq, hd).transpose(1, 2)
            k = qz.mm(F.linear(h, qz.w(sd[p + "self_attn.k_proj.weight"]), sd[p + "self_attn.k_proj.bias"]), "qkv").view(1, N, nkv, hd).transpose(1, 2)
            vv = qz.mm(F.linear(h, qz.w(sd[p + "self_attn.v_proj.weight"]), sd[p + "self_attn.v_proj.bias"]), "qkv").view(1, N, nkv, hd).transpose(1, 2)
            q = qz.mm(q * cos + O.rotate_half(q) * sin, "qkv")
            k = qz.mm(k * cos + O.rotate_half(k) * sin, "qkv")
            o = qz.mm(_attn(q, k, vv, hd ** -0.5, qz), "attn_out").transpose(1, 2).reshape(1, N, D)
            x = qz.every(x + qz.every(F.linear(qz.a_act(o), qz.w(sd[p + "self_attn.o_proj.weight"]))))
            h = qz.a_norm(O.rms_norm(x, sd[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps), "norm2")
            g = qz.every(F.linear(h, qz.w(sd[p + "mlp.gate_proj.weight"])))
            u = qz.every(F.linear(h, qz.w(sd[p + "mlp.up_proj.weight"])))
            f = qz.a_swiglu(qz.every(F.silu(g)) * u)
            x = qz.every(x + qz.every(F.linear(f, qz.w(sd[p + "mlp.down_proj.weight"]))))
        hfin = qz.every(O.rms_norm(x, sd["model.norm.weight"], cfg.rms_norm_eps))
        tt = tts[b][mask[b]]
        pooled = qz.every(hfin[0][tt == 3].reshape(im.shape[0], -1, D).mean(1))
        logits.append(qz.every(F.linear(pooled, sd["out_proj.weight"], sd["out_proj.bias"])))
    return logits
