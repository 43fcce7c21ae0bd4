"""The selector forward (rows A7-A12 of SURVEY.md §8a) as a sequence of calls into libvideoitg_hip.so.

``PackedWeights`` re-lays a checkpoint (HF key names) once at load time into the shapes the
kernels want (fused QKV, 32/32-interleaved gate/up for the fused SwiGLU epilogue, every GEMM K dimension
zero-padded to a multiple of 64, fp32 biases / norm weights / position table).  ``SelectorEngine.forward``
then runs ViT -> pool -> projector -> splice -> bidirectional Qwen2 -> score head with an fp32 residual
stream and 16-bit MFMA inputs.  torch is used for allocation and stream plumbing only.

Reference call sites mirrored: clip_encoder.py:123-154 (tower, hidden_states[-2]), mlp_proj.py:39-75
(projector), eagle_archv1.py:211-400 (splice), grounding_qwen2.py:45-66,135-156 (decoder + head).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F  # weight re-layout only (pad / interpolate of the position table at load time)

from .torch_ops import K          # kernel launches go through torch.ops.videoitg.* (the dispatcher-registered custom ops)
from .config import ITGConfig
from .constants import IMAGE_TOKEN_INDEX
from .ops import pad64

SD = Dict[str, torch.Tensor]


def _pad2(w: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    r, c = w.shape
    if r == rows and c == cols:
        return w
    return F.pad(w, (0, cols - c, 0, rows - r))


def _pad1(b: torch.Tensor, n: int) -> torch.Tensor:
    return b if b.shape[0] == n else F.pad(b, (0, n - b.shape[0]))


class Fp8Weight:
    """[N, pad128(K)] e4m3fn bytes + one fp32 scale per output channel (vitg_quant_rows_fp8 on the packed 16-bit weight)."""

    def __init__(self, w8: torch.Tensor, sw: torch.Tensor):
        self.w8, self.sw = w8, sw

    @property
    def shape(self):
        return self.w8.shape

    def __getitem__(self, rows):
        return Fp8Weight(self.w8[rows], self.sw[rows])


def _fuse_norm_enabled() -> bool:
    from . import _lib as _l
    return _l.PROFILING and os.environ.get("VITG_FUSE_NORM", "0") not in ("", "0")


FP8_GEMMS = ("vit.qkv", "vit.out", "vit.fc1", "vit.fc2", "llm.qkv", "llm.o", "llm.gu", "llm.down")


def normalise_fp8(v) -> object:
    """fp8 weight-path selector -> False | True | frozenset of GEMM names (FP8_GEMMS).  True = every layer GEMM, "vit" / "llm" = that
    part's four, a comma-separated string or an iterable of names = exactly those (the per-GEMM precision policies of
    profiles/r03_fp8_pareto.md).  Truthy non-bool scalars (1, numpy.bool_, an env-parsed int) mean True, falsy ones False; anything
    else is an error - a caller that believes it is on the fp8 tier must not silently get 16-bit weights."""
    if v is None:
        return False
    if isinstance(v, (bool, int)) or type(v).__name__ in ("bool_", "bool"):
        return bool(v)
    if isinstance(v, str):
        names = [n.strip() for n in v.split(",") if n.strip()]
    elif isinstance(v, (set, frozenset, list, tuple)):
        names = list(v)
    else:
        raise ValueError(f"fp8 must be a bool, 'vit', 'llm' or GEMM names out of {FP8_GEMMS}, not {v!r}")
    out = set()
    for n in names:
        if n in ("vit", "llm"):
            out.update(g for g in FP8_GEMMS if g.startswith(n + "."))
        elif n in FP8_GEMMS:
            out.add(n)
        else:
            raise ValueError(f"fp8 must be a bool, 'vit', 'llm' or GEMM names out of {FP8_GEMMS}, not {v!r}")
    if not out:
        return False
    return True if len(out) == len(FP8_GEMMS) else frozenset(out)


@dataclass
class VitLayerW:
    # fused-norm consumers (16-bit path): s[n] = sum_k gamma[k] W[n,k] and c = W beta + bias of the GEMM behind each LayerNorm
    qkv_cs: Optional[torch.Tensor] = None; qkv_c: Optional[torch.Tensor] = None
    fc1_cs: Optional[torch.Tensor] = None; fc1_c: Optional[torch.Tensor] = None
    ln1_w: torch.Tensor = None; ln1_b: torch.Tensor = None
    qkv_w: torch.Tensor = None; qkv_b: torch.Tensor = None
    out_w: torch.Tensor = None; out_b: torch.Tensor = None
    ln2_w: torch.Tensor = None; ln2_b: torch.Tensor = None
    fc1_w: torch.Tensor = None; fc1_b: torch.Tensor = None
    fc2_w: torch.Tensor = None; fc2_b: torch.Tensor = None


@dataclass
class LlmLayerW:
    in_ln: torch.Tensor
    qkv_w: torch.Tensor; qkv_b: torch.Tensor
    o_w: torch.Tensor
    post_ln: torch.Tensor
    gu_w: torch.Tensor
    down_w: torch.Tensor


@dataclass
class PackedWeights:
    cfg: ITGConfig
    dtype: torch.dtype
    device: torch.device
    vit_dtype: Optional[torch.dtype] = None         # MFMA input type of the SigLIP tower (None = dtype): the tower tolerates bf16 (8-bit
                                                    # mantissa) inside the 1e-3 score bound, the 28-layer bidirectional LLM does not
                                                    # (profiles/r02_parity.md), so "tower bf16 + LLM fp16" is a supported mix
    vit_q_prescaled: bool = None                    # the q rows (and bias) of the tower's qkv projection carry head_dim^-0.5 * log2(e): the attention
                                                    # then exponentiates K Q^T as it leaves the matrix pipe (vitg_attention variant bit 1;
                                                    # csrc/attention_ab.hip, QS).  One rounding of c * W instead of W - no extra rounding of Q
                                                    # (a checkpoint stored in 16 bits sees a second rounding of its q weights: 2^-12 relative).
                                                    # None = on unless VITG_Q_PRESCALE=0
    fp8: object = False                             # BASELINE.json configs[4]: tower / decoder GEMM weights as e4m3fn + per-channel scales;
                                                    # True = both parts, "vit" / "llm" = that part only (the other stays 16-bit)
    patch_w: torch.Tensor = None; patch_b: torch.Tensor = None
    patch_wf: torch.Tensor = None                  # the same weight in the (c, ky)-pair layout of vitg_patch_embed16 (im2col inside the GEMM's loader)
    pos_native: torch.Tensor = None                # fp32 [P0, Dv]
    pos_cache: Dict[int, torch.Tensor] = field(default_factory=dict)
    vit: List[VitLayerW] = field(default_factory=list)
    proj0_w: torch.Tensor = None; proj0_b: torch.Tensor = None
    proj2_w: torch.Tensor = None; proj2_b: torch.Tensor = None
    embed: torch.Tensor = None
    llm: List[LlmLayerW] = field(default_factory=list)
    norm_w: torch.Tensor = None
    head_w: torch.Tensor = None; head_b: torch.Tensor = None

    # ---- packing ---------------------------------------------------------------------------------
    def __post_init__(self):
        self.fp8 = normalise_fp8(self.fp8)
        if self.vit_dtype is None:
            self.vit_dtype = self.dtype
        if self.vit_q_prescaled is None:
            self.vit_q_prescaled = os.environ.get("VITG_Q_PRESCALE", "1") not in ("", "0")
        self._packing = self.dtype                  # type the _w16 / _wl helpers currently emit (switched while the tower is packed)
        self._packing_part = "llm"

    def _w16(self, w: torch.Tensor, rows: Optional[int] = None, cols: Optional[int] = None) -> torch.Tensor:
        w = w.to(self.device)
        rows = w.shape[0] if rows is None else rows
        cols = pad64(w.shape[1]) if cols is None else cols
        return _pad2(w, rows, cols).to(self._packing).contiguous()

    def fp8_on(self, gemm: str) -> bool:
        """is this layer GEMM (one of FP8_GEMMS) on the fp8 weight path?"""
        return self.fp8 is True or (isinstance(self.fp8, frozenset) and gemm in self.fp8)

    def fp8_part(self, part: str) -> bool:
        """is ANY GEMM of this part ("vit" / "llm") on the fp8 weight path?"""
        return any(self.fp8_on(g) for g in FP8_GEMMS if g.startswith(part + "."))

    def _wl(self, w: torch.Tensor, rows: Optional[int] = None, gemm: str = ""):
        """weight of a tower / decoder layer GEMM: 16-bit, or - on the fp8 weight path - e4m3fn bytes + per-output-channel scales
        (quantised from the SAME 16-bit-rounded, zero-padded matrix the 16-bit path would multiply with)"""
        w16 = self._w16(w, rows)
        if not self.fp8_on(gemm):                               # (fp8 normalised in __post_init__: False | True | frozenset of GEMM names)
            return w16
        w8, sw = K.quant_rows_fp8(w16, w16.shape[1], torch.empty((w16.shape[0], K.pad128(w16.shape[1])), dtype=torch.uint8, device=self.device),
                                  torch.empty(w16.shape[0], dtype=torch.float32, device=self.device))
        return Fp8Weight(w8, sw)

    def _f32(self, t: torch.Tensor, n: Optional[int] = None) -> torch.Tensor:
        t = t.to(self.device, torch.float32)
        return (_pad1(t, n) if n is not None else t).contiguous()

    def pack_vision_embeddings(self, vsd: SD) -> None:
        e = "vision_model.embeddings."
        w = vsd[e + "patch_embedding.weight"]
        self._packing = self.vit_dtype
        self.patch_w = self._w16(w.reshape(w.shape[0], -1))
        # (c, ky)-pair layout for the fused loader (csrc/gemm_mi16.hip PATCH): k = 16 (c p + ky) + slot; slots 0..7 = kx 0..7, slots 8..15 =
        # kx p-8..p-1 where the slots of the kx that chunk RE-READS (p-8..7) are zero; rows of K padding zero.  Same 16-bit values as patch_w.
        p_ = w.shape[-1]
        if w.dim() == 4 and w.shape[1] == 3 and w.shape[2] == p_ and 8 <= p_ <= 16 and p_ % 2 == 0:
            rows = w.reshape(w.shape[0], 3 * p_, p_).to(self.device)
            wf = torch.zeros((w.shape[0], (3 * p_ + 3) // 4 * 4, 16), dtype=rows.dtype, device=self.device)
            wf[:, :3 * p_, :8] = rows[:, :, :8]
            wf[:, :3 * p_, 16 - (p_ - 8):] = rows[:, :, 8:]
            self.patch_wf = wf.reshape(w.shape[0], -1).to(self._packing).contiguous()
        self._packing = self.dtype
        self.patch_b = self._f32(vsd[e + "patch_embedding.bias"])
        self.pos_native = self._f32(vsd[e + "position_embedding.weight"])

    def pack_vision_layer(self, vsd: SD, i: int) -> None:
        v = self.cfg.vision
        p = f"vision_model.encoder.layers.{i}."
        fvp = pad64(v.intermediate_size)
        # softmax scale and the base-2 factor folded into the q projection (fp32, before the one rounding to 16 bits)
        qs = v.head_dim ** -0.5 * 1.4426950408889634 if self.vit_q_prescaled else 1.0
        qkv_w = torch.cat([vsd[p + f"self_attn.{n}_proj.weight"].float() * (qs if n == "q" else 1.0) for n in "qkv"], 0)
        qkv_b = torch.cat([vsd[p + f"self_attn.{n}_proj.bias"].float() * (qs if n == "q" else 1.0) for n in "qkv"], 0)
        self._packing, self._packing_part = self.vit_dtype, "vit"
        self.vit.append(VitLayerW(
            ln1_w=self._f32(vsd[p + "layer_norm1.weight"]), ln1_b=self._f32(vsd[p + "layer_norm1.bias"]),
            qkv_w=self._wl(qkv_w, gemm="vit.qkv"), qkv_b=self._f32(qkv_b),
            out_w=self._wl(vsd[p + "self_attn.out_proj.weight"], gemm="vit.out"), out_b=self._f32(vsd[p + "self_attn.out_proj.bias"]),
            ln2_w=self._f32(vsd[p + "layer_norm2.weight"]), ln2_b=self._f32(vsd[p + "layer_norm2.bias"]),
            fc1_w=self._wl(vsd[p + "mlp.fc1.weight"], rows=fvp, gemm="vit.fc1"), fc1_b=self._f32(vsd[p + "mlp.fc1.bias"], fvp),
            fc2_w=self._wl(vsd[p + "mlp.fc2.weight"], gemm="vit.fc2"), fc2_b=self._f32(vsd[p + "mlp.fc2.bias"]),
        ))
        self._packing, self._packing_part = self.dtype, "llm"
        if _fuse_norm_enabled() and not isinstance(self.vit[-1].qkv_w, Fp8Weight):
            # load-time constants of the fused LayerNorm consumers (profiling library + VITG_FUSE_NORM=1 only: the fused kernels measured
            # 4 % slower than the stand-alone norm, profiles/r02_fused_norm.md, and are not in the product library), from the
            # 16-bit-rounded weights the MFMA will see
            lw = self.vit[-1]
            Dv = v.hidden_size
            for wname, bname, gam, bet, cs, cc in (("qkv_w", "qkv_b", lw.ln1_w, lw.ln1_b, "qkv_cs", "qkv_c"),
                                                  ("fc1_w", "fc1_b", lw.ln2_w, lw.ln2_b, "fc1_cs", "fc1_c")):
                wf = getattr(lw, wname)[:, :Dv].float()
                setattr(lw, cs, (wf @ gam).contiguous())
                setattr(lw, cc, (wf @ bet + getattr(lw, bname)).contiguous())

    def pack_projector_and_head(self, sd: SD) -> None:
        pre = "model.mm_projector.out_projection."
        dp = pad64(self.cfg.hidden_size)
        self.proj0_w = self._w16(sd[pre + "0.weight"], rows=dp)
        self.proj0_b = self._f32(sd[pre + "0.bias"], dp)
        self.proj2_w = self._w16(sd[pre + "2.weight"])
        self.proj2_b = self._f32(sd[pre + "2.bias"])
        self.embed = sd["model.embed_tokens.weight"].to(self.device, self.dtype).contiguous()
        self.norm_w = self._f32(sd["model.norm.weight"])
        self.head_w = self._f32(sd["out_proj.weight"].reshape(-1))
        self.head_b = self._f32(sd["out_proj.bias"].reshape(-1))

    def pack_llm_layer(self, sd: SD, i: int) -> None:
        p = f"model.layers.{i}."
        fp = pad64(self.cfg.intermediate_size)
        qkv_w = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        qkv_b = torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        g = sd[p + "mlp.gate_proj.weight"].to(self.device)
        u = sd[p + "mlp.up_proj.weight"].to(self.device)
        g = _pad2(g, fp, g.shape[1]).reshape(fp // 32, 1, 32, -1)
        u = _pad2(u, fp, u.shape[1]).reshape(fp // 32, 1, 32, -1)
        gu = torch.cat([g, u], 1).reshape(2 * fp, -1)           # rows: [32 gate | 32 up] repeated
        self.llm.append(LlmLayerW(
            in_ln=self._f32(sd[p + "input_layernorm.weight"]),
            qkv_w=self._wl(qkv_w, gemm="llm.qkv"), qkv_b=self._f32(qkv_b),
            o_w=self._wl(sd[p + "self_attn.o_proj.weight"], gemm="llm.o"),
            post_ln=self._f32(sd[p + "post_attention_layernorm.weight"]),
            gu_w=self._wl(gu, gemm="llm.gu"),
            down_w=self._wl(sd[p + "mlp.down_proj.weight"], gemm="llm.down"),
        ))

    @staticmethod
    def from_state_dicts(cfg: ITGConfig, sd: SD, vsd: SD, dtype: torch.dtype, device, fp8: "bool | str" = False,
                         vit_dtype: Optional[torch.dtype] = None, vit_q_prescaled: Optional[bool] = None) -> "PackedWeights":
        pw = PackedWeights(cfg=cfg, dtype=dtype, device=torch.device(device), fp8=fp8, vit_dtype=vit_dtype, vit_q_prescaled=vit_q_prescaled)
        pw.pack_vision_embeddings(vsd)
        for i in range(cfg.num_vit_layers_needed()):
            pw.pack_vision_layer(vsd, i)
        pw.pack_projector_and_head(sd)
        for i in range(cfg.num_hidden_layers):
            pw.pack_llm_layer(sd, i)
        return pw

    def doubled(self) -> "SplitWeights":
        """the [W | W] copies the operand-split tier multiplies K-concatenated (hi | lo) operands with (built once, on first use)"""
        if getattr(self, "_doubled", None) is None:
            self._doubled = SplitWeights(self)
        return self._doubled

    def pos_table(self, grid: int) -> torch.Tensor:
        """fp32 [grid*grid, Dv]; bicubic resample of the native table when the input side is not the native one
        ([3P] SiglipVisionEmbeddings.interpolate_pos_encoding).  Load-time/cache work, not on the timed path."""
        if grid not in self.pos_cache:
            pos = self.pos_native
            g0 = int(round(math.sqrt(pos.shape[0])))
            if g0 != grid:
                p = pos.reshape(1, g0, g0, -1).permute(0, 3, 1, 2)
                p = F.interpolate(p, size=(grid, grid), mode="bicubic", align_corners=False)
                pos = p.permute(0, 2, 3, 1).reshape(grid * grid, -1).contiguous()
            self.pos_cache[grid] = pos
        return self.pos_cache[grid]


def _dbl(w: torch.Tensor) -> torch.Tensor:
    return torch.cat([w, w], 1).contiguous()


class SplitWeights:
    """Operand-split tier (csrc/split.hip): a split A operand arrives as [hi | lo] with K = 2 Kp, so the weight of every GEMM that consumes one is laid
    twice side by side, [N, 2 Kp] = [W | W] - sum_k hi_k w_k + sum_k lo_k w_k in one accumulator, through the unchanged GEMM main loop.  The patch
    embedding is not here: its operand is the 16-bit-exact pixel tensor itself."""

    def __init__(self, pw: PackedWeights):
        if pw.fp8 is not False:
            raise ValueError("the operand-split tier multiplies 16-bit-exact weights; it does not combine with the fp8 weight path")
        if pw.dtype != torch.float16 or pw.vit_dtype != torch.float16:
            raise ValueError("the operand-split tier splits into fp16 pairs (11 + 11 bits): load the weights as torch.float16")
        if pw.vit_q_prescaled:
            raise ValueError("PackedWeights(vit_q_prescaled=True) rounds (scale * W_q) to 16 bits - a deviation of 2^-12 per weight from the checkpoint "
                             "that the operand-split tier exists to avoid; build the weights with vit_q_prescaled=False")
        self.vit = [dict(qkv=_dbl(l.qkv_w), out=_dbl(l.out_w), fc1=_dbl(l.fc1_w), fc2=_dbl(l.fc2_w)) for l in pw.vit]
        self.proj0, self.proj2 = _dbl(pw.proj0_w), _dbl(pw.proj2_w)
        self.llm = [dict(qkv=_dbl(l.qkv_w), o=_dbl(l.o_w), gu=_dbl(l.gu_w), down=_dbl(l.down_w)) for l in pw.llm]


class SelectorEngine:
    """Runs the forward.  One instance per process / GPU."""

    def __init__(self, weights: PackedWeights, attn_variant: int = 0, operand_split: bool = False):
        self.w = weights
        # operand-split precision tier (round 6; csrc/split.hip): every MFMA operand as an fp16 (hi, lo) pair - fp32-grade GEMMs and attention at
        # 2 x / 3 x the MFMA work, for weights on which 11-bit operands miss north_star's 1e-3 (massive-activation channels, profiles/r05_heavy_tail.md).
        # fp32 residual streams, eager launches (no graph capture); context-parallel through the blocking K|V exchange of pairs.
        self.operand_split = bool(operand_split)
        self.sw = weights.doubled() if self.operand_split else None
        self.cfg = weights.cfg
        self.dtype = weights.dtype                # LLM / projector MFMA input type
        self.vdtype = weights.vit_dtype           # tower MFMA input type
        self.device = weights.device
        self.attn_variant = attn_variant
        self.skip_dead_rows = True        # last tower layer only on the tokens the pool reads (see vit_forward)
        # 16-bit residual stream in the TOWER (round 3; vitg_gemm_res16 / vitg_layernorm16): the reference's own arithmetic (model.half()),
        # half the bytes of every LayerNorm read and out-proj / fc2 read-modify-write.  The decoder's stream stays fp32.
        # Default since the gate held on every complete-pass fixture (max |dscore| 3.1e-4 against 1.5e-4 on the fp32 stream, Top-K sets
        # identical, step -2.8 %: profiles/r03_residual16.md); VITG_RES16=0 / engine.residual16 = False restores the fp32 stream.
        self.residual16 = os.environ.get("VITG_RES16", "1") not in ("", "0")
        # patch embedding: im2col inside the GEMM's A-tile loader (round 4, vitg_patch_embed16); VITG_FUSE_PATCHIFY=0 = patchify kernel + GEMM (A/B, cross-check)
        self.fuse_patchify = os.environ.get("VITG_FUSE_PATCHIFY", "1") not in ("", "0")
        self.vit_stream_bytes = 4                        # element size of the tower's residual stream in the LAST vit_forward (2 when the 16-bit stream ran)
        # LayerNorm / RMSNorm folded into the GEMMs around them (vitg_gemm_norm_*).  Measured 4 % SLOWER on the step than the stand-alone
        # norm kernels (profiles/r02_fused_norm.md), so the kernels live in the profiling library and this switch only exists there.
        # (never together with fp8 layer weights - the fused consumers only exist for 16-bit operands)
        self.fuse_norm = _fuse_norm_enabled() and getattr(weights, "fp8", False) is False
        # fp8 tower: fc1 (+ GELU) hands its output to fc2 as e4m3 + MX block scales straight from its epilogue (vitg_gemm_fp8_mx) instead
        # of 16 bits + a stand-alone per-token quantiser pass; VITG_FP8_MX=0 keeps the per-token form (A/B, tests)
        self.fp8_mx = os.environ.get("VITG_FP8_MX", "1") not in ("", "0")
        # fp8 tower GEMMs on the tower's 16-bit residual stream (round 4; before: the fp32 stream, its out-proj / fc2 HBM-bound)
        self.fp8_r16 = os.environ.get("VITG_FP8_R16", "1") not in ("", "0")
        self._normplan: Dict[tuple, int] = {}
        self._rope: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._ws: Dict[str, torch.Tensor] = {}
        self._poolsrc: Dict[tuple, object] = {}
        self._splitk: Dict[tuple, tuple] = {}        # (M, N, K) -> (splits, workspace bytes) from vitg_gemm_splitk_plan
        self._splitk16: Dict[tuple, int] = {}        # (M, N, K) -> splits for 16-bit-output GEMMs (vitg_gemm_splitk16)
        self._ws_tag = ""                             # "_side" while launching on the context-parallel side stream
        self.splitk16 = os.environ.get("VITG_SPLITK16", "1") not in ("", "0")     # A/B: 0 = plain kernels for the few-tile 16-bit-output GEMMs
        self._side: Optional[torch.cuda.Stream] = None   # context-parallel layers: K|V projection + exchange run here, under the Q projection
        # VITG_CP_OVERLAP=0 (A/B): fused QKV projection and a blocking exchange, as in round 2.  The split costs one rank ~1 ms of compute per step
        # (two sub-round launches instead of one, tools/cp_rank_time.py) and buys the exchange - est. 0.1-0.15 ms x 28 layers - off the critical path
        self.cp_overlap = os.environ.get("VITG_CP_OVERLAP", "1") not in ("", "0")
        # fp32-residual GEMMs with fewer output tiles than CUs (one context-parallel rank): VITG_STREAMK=1 runs them as a stream-K partition instead
        # of split-K x2.  Opt-in: measured SLOWER on the per-rank o / down shapes (447 / 694 against 682 / 911 TF/s, profiles/r03_context_parallel.md)
        # (profiling library only since round 4: the product library no longer carries the stream-K kernels)
        from . import _lib as _l
        self.stream_k = _l.PROFILING and os.environ.get("VITG_STREAMK", "0") not in ("", "0")

    # workspace: ONE buffer per name, grown to the largest row count seen and handed out as a [:rows] view, so a run over
    # documents with varying prompt lengths / frame counts holds max-size buffers instead of one set per distinct shape.
    # Zero-initialised buffers keep their pad COLUMNS zero (kernels only ever write the payload columns of a row).
    def _buf(self, name: str, shape, dtype, zero: bool = False) -> torch.Tensor:
        rows, rest = int(shape[0]), tuple(int(s) for s in shape[1:])
        t = self._ws.get(name)
        if t is None or t.dtype != dtype or tuple(t.shape[1:]) != rest or t.shape[0] < rows:
            t = (torch.zeros if zero else torch.empty)((rows,) + rest, dtype=dtype, device=self.device)
            self._ws[name] = t                                   # the smaller buffer (if any) is dropped here
        return t[:rows]

    def workspace_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._ws.values())

    def workspace_tensors(self) -> List[torch.Tensor]:
        """the live buffers (a captured graph keeps these referenced so that a later growth cannot free what it replays on)"""
        return list(self._ws.values())

    def release_workspace(self) -> None:
        self._ws.clear()

    # ---- GEMM operands: 16-bit tensors, or (e4m3fn bytes, per-token scales) on the fp8 weight path ---------------------------
    def _norm(self, x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], name: str, d: int, eps: float,
              dt: Optional[torch.dtype] = None, cw=None):
        """LayerNorm (b given) / RMSNorm of the fp32 residual rows as the A operand of the next GEMM, whose weight `cw` decides the
        operand form: e4m3 + per-token scales for an Fp8Weight, 16 bits otherwise"""
        M = x.shape[0]
        if isinstance(cw, Fp8Weight):
            a8 = self._buf(name + "_q8", (M, K.pad128(d)), torch.uint8)
            sa = self._buf(name + "_qs", (M,), torch.float32)
            K.norm_quant_fp8(x, w, b, a8, sa, d, eps)
            return a8, sa
        h = self._buf(name, (M, pad64(d)), dt or self.dtype, zero=True)
        if b is not None:
            K.layernorm(x, w, b, h, d, eps)
        else:
            K.rmsnorm(x, w, h, d, eps)
        return h

    def _operand(self, t: torch.Tensor, name: str, d: int, cw=None):
        """a 16-bit activation (attention output, GELU / SwiGLU output) as the A operand of the next GEMM (weight `cw`)"""
        if not isinstance(cw, Fp8Weight):
            return t
        a8 = self._buf(name + "_q8", (t.shape[0], K.pad128(d)), torch.uint8)
        sa = self._buf(name + "_qs", (t.shape[0],), torch.float32)
        K.quant_rows_fp8(t, d, a8, sa)
        return a8, sa

    def _mlp_fp8_mx(self, h, fc1_w: "Fp8Weight", fc1_b, fc2_w: "Fp8Weight", fc2_b, x: torch.Tensor, name: str) -> None:
        """x += fc2(gelu_tanh(fc1(h))) with the hidden activation kept as e4m3 + MX block scales between the two GEMMs"""
        M, kp = h[0].shape[0], fc2_w.w8.shape[1]
        f8 = self._buf(name + "_f8", (M, kp), torch.uint8)
        fsc = self._buf(name + "_fsc", (kp // 128, M, 4), torch.uint8)
        K.gemm_fp8_mx(h[0], fc1_w.w8, fc1_w.sw, f8, out_dtype=self.vdtype, sa=h[1], o_mx=fsc, bias=fc1_b, act=K.ACT_GELU_TANH)
        K.gemm_fp8_mx(f8, fc2_w.w8, fc2_w.sw, x, out_dtype=self.vdtype, a_mx=fsc, bias=fc2_b, res=x)

    def _producer(self, a, w, x: torch.Tensor, *, bias, res, gamma: Optional[torch.Tensor], name: str, eps: float, rms: bool,
                  dt: torch.dtype, res_mod: int = 0):
        """x = a @ w.T + bias + res (the fp32 residual update).  With fuse_norm and a shape the fused kernels cover, the same launch
        also emits the operand and the row statistics of the norm that FOLLOWS (weight `gamma`): returns (xh, stats) for
        `_consumer`; otherwise runs the plain GEMM and returns None (the caller then normalises x with the stand-alone kernel)."""
        fused = self.fuse_norm and gamma is not None and not isinstance(w, Fp8Weight)
        if fused:
            M, N, Kd = a.shape[0], w.shape[0], w.shape[1]
            key = (M, N, Kd)
            cols = self._normplan.get(key)
            if cols is None:
                cols = K.gemm_norm_plan(M, N, Kd)
                if cols and res_mod == 0 and K.gemm_splitk_plan(M, N, Kd)[0] > 1:
                    cols = 0                                   # few-tile shapes go through split-K instead
                self._normplan[key] = cols
            if cols:
                P = N // cols
                xh = self._buf(name + "_xh", (M, pad64(N)), dt, zero=True)
                part = self._buf(name + "_npart", (M * P * 2,), torch.float32)
                stats = self._buf(name + "_nstat", (M * 2,), torch.float32)
                K.gemm_norm_producer(a, w, x, res=res, next_gamma=gamma, xh=xh, partials=part, bias=bias, res_mod=res_mod)
                K.norm_stats_finalize(part, M, P, cols, eps, rms, stats)
                return xh, stats
        self._gemm(a, w, x, mode=K.GEMM_F32, bias=bias, res=res, res_mod=res_mod)
        return None

    def _gemm(self, a, w, out: torch.Tensor, **kw) -> torch.Tensor:
        if isinstance(w, Fp8Weight):
            return K.gemm_fp8(a[0], a[1], w.w8, w.sw, out, out_dtype=out.dtype if out.dtype != torch.float32 else self.dtype, **kw)
        if kw.get("mode") == K.GEMM_F32 and kw.get("act", 0) == 0 and not kw.get("res_mod", 0) and out.dtype == torch.float32:
            # few output tiles + long K (o / down projection of one context-parallel rank): deterministic split-K
            M, N, Kd = a.shape[0], w.shape[0], w.shape[1]
            plan = self._splitk.get((M, N, Kd))
            if plan is None:
                sk = K.gemm_streamk_plan(M, N, Kd) if self.stream_k else (0, 0)
                plan = self._splitk[(M, N, Kd)] = ("stream", sk[0], sk[1]) if sk[0] > 0 else ("split",) + tuple(K.gemm_splitk_plan(M, N, Kd))
            if plan[0] == "stream":
                # stream-K (round 3): every CU gets the same number of K-tiles; the partials of a tile are added in block order (bit-reproducible)
                ws = self._buf("streamk_ws", (plan[2] // 4,), torch.float32)
                return K.gemm_streamk(a, w, out, ws, plan[1], bias=kw.get("bias"), res=kw.get("res"))
            if plan[1] > 1:
                ws = self._buf("splitk_ws", (plan[1] * M * N,), torch.float32)
                return K.gemm_splitk(a, w, out, ws, plan[1], bias=kw.get("bias"), res=kw.get("res"))
        if (kw.get("mode") == K.GEMM_STORE16 and kw.get("act", 0) == 0 and kw.get("res") is None and kw.get("n") is None and kw.get("m") is None
                and out.dtype == a.dtype and self.splitk16):
            # round 5: the same for 16-bit outputs - the Q and K|V projections of one context-parallel rank (98 / 28 tiles of 256 x 256 on 256 CUs: the plain
            # launch runs a whole K = 3584 tile per block with most of the chip idle); one-GPU shapes plan 1 split and take the plain kernel below
            M, N, Kd = a.shape[0], w.shape[0], w.shape[1]
            sp = self._splitk16.get((M, N, Kd))
            if sp is None:
                sp = self._splitk16[(M, N, Kd)] = K.gemm_splitk_plan(M, N, Kd)[0] if (N % 4 == 0 and out.stride(0) % 4 == 0) else 1
            if sp > 1:
                # (the K|V projection runs on the side stream WHILE the Q projection runs on the main one: each stream its own partial buffer)
                ws = self._buf("splitk_ws" + self._ws_tag, (sp * M * N,), torch.float32)
                return K.gemm_splitk16(a, w, out, ws, sp, bias=kw.get("bias"))
        return K.gemm(a, w, out, **kw)

    # ---- which tokens does the bilinear pool read? (mlp_proj.py:57-68 semantics: align_corners=False) ----------------
    def pool_sources(self, G: int, hw: int):
        """(idx int32 [Pc] sorted token ids y*G+x that carry weight, cmap int32 [G*G] token -> compact row or -1).
        27->5: 81 of 729 tokens, 32->5: 100 of 1024.  None when there is no pooling (hw == G)."""
        key = (G, hw)
        if key not in self._poolsrc:
            if hw >= G:
                self._poolsrc[key] = None
            else:
                used = set()
                sc = torch.tensor(float(G), dtype=torch.float32) / torch.tensor(float(hw), dtype=torch.float32)
                for o in range(hw):                              # same fp32 arithmetic as pool_kernel
                    src = (torch.tensor(float(o), dtype=torch.float32) + 0.5) * sc - 0.5
                    i0 = int(max(src.item(), 0.0))
                    used.update((i0, min(i0 + 1, G - 1)))
                rows = sorted(used)
                idx = torch.tensor([y * G + x for y in rows for x in rows], dtype=torch.int32)
                cmap = torch.full((G * G,), -1, dtype=torch.int32)
                cmap[idx.long()] = torch.arange(idx.numel(), dtype=torch.int32)
                self._poolsrc[key] = (idx.to(self.device), cmap.to(self.device))
        return self._poolsrc[key]

    # ---- A7: SigLIP tower -> fp32 features ------------------------------------------------------------------------
    def vit_forward(self, pix: torch.Tensor, hw: Optional[int] = None):
        """-> (features fp32, G, cmap, Pc).  Without `hw` (or when nothing is pooled) features are dense [T*P, Dv] and
        cmap is None.  With a pooled grid, the LAST needed layer computes K/V for all tokens but queries, out-proj and MLP
        only for the Pc tokens the pool reads (their outputs are the only ones anything downstream consumes), and features
        are compact [T*Pc, Dv] with cmap: token -> compact row."""
        if self.operand_split:
            return self.vit_forward_split(pix, hw)
        cfg, v, w = self.cfg, self.cfg.vision, self.w
        T, _, R, _ = pix.shape
        G = R // v.patch_size
        P = G * G
        M = T * P
        Dv, Dvp, Fvp = v.hidden_size, pad64(v.hidden_size), pad64(v.intermediate_size)
        H, hd = v.num_attention_heads, v.head_dim
        eps, vdt = v.layer_norm_eps, self.vdtype
        # (fp16 only: a bf16 stream would carry 8 mantissa bits through 26 layers; never together with the fused norm)
        # (and only when the projector runs in the same type: vitg_bilinear_pool16 writes the pooled rows in the tower's type - ADVICE r3)
        # (round 4: fp8 tower GEMMs run on it too - vitg_gemm_fp8_res16 / vitg_norm_quant_fp8_16; VITG_FP8_R16=0 keeps that tier on the fp32 stream: A/B)
        r16 = (self.residual16 and vdt == torch.float16 and self.dtype == vdt and (self.fp8_r16 or not w.fp8_part("vit")) and not self.fuse_norm
               and Dv % 8 == 0)
        self.vit_stream_bytes = 2 if r16 else 4          # what this call's residual stream really was (bench.py prices LayerNorm / pool on it)
        x = self._buf("vit_x16" if r16 else "vit_x", (M, Dv), vdt if r16 else torch.float32)
        src = self.pool_sources(G, hw) if (hw is not None and self.skip_dead_rows and len(w.vit) > 0) else None
        full_layers = w.vit if src is None else w.vit[:-1]

        def norm(xs, gw, gb, name, cw=None):
            """LayerNorm of stream rows as the A operand of the next GEMM (weight cw: e4m3 + per-token scales for an Fp8Weight)"""
            if not r16:
                return self._norm(xs, gw, gb, name, Dv, eps, vdt, cw=cw)
            if isinstance(cw, Fp8Weight):
                a8 = self._buf(name + "_q8", (xs.shape[0], K.pad128(Dv)), torch.uint8)
                sa = self._buf(name + "_qs", (xs.shape[0],), torch.float32)
                return K.norm_quant_fp8_16(xs, gw, gb, a8, sa, Dv, eps)
            h_ = self._buf(name, (xs.shape[0], Dvp), vdt, zero=True)
            return K.layernorm16(xs, gw, gb, h_, Dv, eps)

        def update(a, wt, bias, xs, name="vit_u", d=0):
            """xs += a @ wt.T + bias  (16-bit stream: one rounding per update, fp32 accumulation inside the epilogue); `a` = 16-bit rows, quantised
            here (buffer `name`, payload width d) when wt is an Fp8Weight"""
            if r16 and isinstance(wt, Fp8Weight):
                a8, sa = self._operand(a, name, d, wt)
                K.gemm_fp8_res16(a8, wt.w8, wt.sw, xs, res16=xs, sa=sa, bias=bias)
            elif r16:
                K.gemm_res16(a, wt, xs, bias=bias, res16=xs)
            else:
                self._gemm(a, wt, xs, mode=K.GEMM_F32, bias=bias, res=xs)

        def mlp_mx16(hq8, fc1_w, fc1_b, fc2_w, fc2_b, xs, name):
            """_mlp_fp8_mx on the 16-bit stream: the hidden activation leaves fc1 as e4m3 + MX block scales, fc2 adds into the 16-bit rows"""
            Mx, kp = hq8[0].shape[0], fc2_w.w8.shape[1]
            f8 = self._buf(name + "_f8", (Mx, kp), torch.uint8)
            fsc = self._buf(name + "_fsc", (kp // 128, Mx, 4), torch.uint8)
            K.gemm_fp8_mx(hq8[0], fc1_w.w8, fc1_w.sw, f8, out_dtype=self.vdtype, sa=hq8[1], o_mx=fsc, bias=fc1_b, act=K.ACT_GELU_TANH)
            K.gemm_fp8_res16(f8, fc2_w.w8, fc2_w.sw, xs, res16=xs, a_mx=fsc, bias=fc2_b)

        # `pend` = (xh, stats) when the GEMM that produced x also emitted the operand / statistics of the next LayerNorm (fuse_norm)
        # patch embedding (+ bias + position table).  On the 16-bit stream the im2col happens inside the GEMM's operand loader
        # (vitg_patch_embed16: no patchify pass, no [T*P, 640] operand in HBM); other streams / geometries: vitg_patchify + GEMM
        fused_pe = (self.fuse_patchify and r16 and w.patch_wf is not None and pix.dtype == vdt and pix.is_contiguous()
                    and K.patch_embed_fusable(T, R, v.patch_size, Dv))
        if fused_pe:
            K.patch_embed16(pix, w.patch_wf, x, patch=v.patch_size, bias=w.patch_b, pos=w.pos_table(G))
            pend = None
        else:
            a0 = K.patchify(pix, v.patch_size, w.patch_w.shape[1], self.vdtype)
            if r16:
                K.gemm_res16(a0, w.patch_w, x, bias=w.patch_b, res32=w.pos_table(G), res_mod=P)
                pend = None
            else:
                pend = self._producer(a0, w.patch_w, x, bias=w.patch_b, res=w.pos_table(G), res_mod=P, gamma=w.vit[0].ln1_w if w.vit else None,
                                      name="vit", eps=eps, rms=False, dt=vdt)
            del a0
        qkv = self._buf("vit_qkv", (M, 3 * Dv), vdt)
        o = self._buf("vit_o", (M, Dvp), vdt, zero=True)
        mx_mlp = self.fp8_mx and all(isinstance(lw.fc1_w, Fp8Weight) and isinstance(lw.fc2_w, Fp8Weight) for lw in w.vit)
        f = None if mx_mlp else self._buf("vit_f", (M, Fvp), vdt)      # (the MX path keeps the hidden activation in its own e4m3 buffer)
        scale = hd ** -0.5
        vit_variant = self.attn_variant | (2 if w.vit_q_prescaled else 0)     # bit 1: Q arrives as Q * scale * log2(e) (PackedWeights.vit_q_prescaled)
        st = (P * 3 * Dv, 3 * Dv, hd)
        for li, lw in enumerate(full_layers):
            if pend is not None:
                K.gemm_norm_consumer(pend[0], lw.qkv_w, qkv, mode=K.GEMM_STORE16, stats=pend[1], colsum=lw.qkv_cs, bias=lw.qkv_c)
            else:
                h = norm(x, lw.ln1_w, lw.ln1_b, "vit_h", lw.qkv_w)
                self._gemm(h, lw.qkv_w, qkv, mode=K.GEMM_STORE16, bias=lw.qkv_b)
            K.attention(qkv, qkv[:, Dv:], qkv[:, 2 * Dv:], o, batch=T, heads=H, kv_heads=H, seq=P, head_dim=hd,
                        q_strides=st, k_strides=st, v_strides=st, o_strides=(P * Dvp, Dvp, hd), scale=scale,
                        variant=vit_variant)
            if r16:
                update(o, lw.out_w, lw.out_b, x, "vit_o", Dv)
                h = norm(x, lw.ln2_w, lw.ln2_b, "vit_h", lw.fc1_w)
                if self.fp8_mx and isinstance(lw.fc1_w, Fp8Weight) and isinstance(lw.fc2_w, Fp8Weight):
                    mlp_mx16(h, lw.fc1_w, lw.fc1_b, lw.fc2_w, lw.fc2_b, x, "vit")
                    continue
                self._gemm(h, lw.fc1_w, f, mode=K.GEMM_STORE16, bias=lw.fc1_b, act=K.ACT_GELU_TANH)
                update(f, lw.fc2_w, lw.fc2_b, x, "vit_f", Fvp)
                continue
            pend = self._producer(self._operand(o, "vit_o", Dv, lw.out_w), lw.out_w, x, bias=lw.out_b, res=x, gamma=lw.ln2_w, name="vit", eps=eps,
                                  rms=False, dt=vdt)
            if pend is not None:
                K.gemm_norm_consumer(pend[0], lw.fc1_w, f, mode=K.GEMM_STORE16, stats=pend[1], colsum=lw.fc1_cs, bias=lw.fc1_c,
                                     act=K.ACT_GELU_TANH)
            else:
                h = self._norm(x, lw.ln2_w, lw.ln2_b, "vit_h", Dv, eps, vdt, cw=lw.fc1_w)
                if self.fp8_mx and isinstance(lw.fc1_w, Fp8Weight) and isinstance(lw.fc2_w, Fp8Weight):
                    self._mlp_fp8_mx(h, lw.fc1_w, lw.fc1_b, lw.fc2_w, lw.fc2_b, x, "vit")
                    pend = None
                    continue
                self._gemm(h, lw.fc1_w, f, mode=K.GEMM_STORE16, bias=lw.fc1_b, act=K.ACT_GELU_TANH)
            nxt = w.vit[li + 1].ln1_w if li + 1 < len(w.vit) else None          # the norm in front of the NEXT layer's qkv
            pend = self._producer(self._operand(f, "vit_f", Fvp, lw.fc2_w), lw.fc2_w, x, bias=lw.fc2_b, res=x, gamma=nxt, name="vit", eps=eps,
                                  rms=False, dt=vdt)
        if src is None:
            return x, G, None, 0
        # ---- last needed layer on compact query rows -------------------------------------------------------------
        idx, cmap = src
        Pc = idx.numel()
        Mc = T * Pc
        lw = w.vit[-1]
        if pend is not None:                                                       # K|V for every token
            K.gemm_norm_consumer(pend[0], lw.qkv_w[Dv:], qkv[:, Dv:], mode=K.GEMM_STORE16, stats=pend[1], colsum=lw.qkv_cs[Dv:],
                                 bias=lw.qkv_c[Dv:])
        else:
            h = norm(x, lw.ln1_w, lw.ln1_b, "vit_h", lw.qkv_w)
            self._gemm(h, lw.qkv_w[Dv:], qkv[:, Dv:], mode=K.GEMM_STORE16, bias=lw.qkv_b[Dv:])
        xq = self._buf("vit_xq16" if r16 else "vit_xq", (Mc, Dv), vdt if r16 else torch.float32)
        K.gather_rows(x, idx, P, T, xq, Dv)
        # LayerNorm is per row: normalising the gathered residual rows equals gathering the normalised rows
        hq = norm(xq, lw.ln1_w, lw.ln1_b, "vit_hq", lw.qkv_w)
        qc = self._buf("vit_qc", (Mc, Dv), vdt)
        self._gemm(hq, lw.qkv_w[:Dv], qc, mode=K.GEMM_STORE16, bias=lw.qkv_b[:Dv])                  # Q for the pooled tokens
        oc = self._buf("vit_oc", (Mc, Dvp), vdt, zero=True)
        K.attention(qc, qkv[:, Dv:], qkv[:, 2 * Dv:], oc, batch=T, heads=H, kv_heads=H, seq=Pc, kv_seq=P, head_dim=hd,
                    q_strides=(Pc * Dv, Dv, hd), k_strides=st, v_strides=st, o_strides=(Pc * Dvp, Dvp, hd), scale=scale,
                    variant=vit_variant)
        if r16:
            update(oc, lw.out_w, lw.out_b, xq, "vit_oc", Dv)
        else:
            self._gemm(self._operand(oc, "vit_oc", Dv, lw.out_w), lw.out_w, xq, mode=K.GEMM_F32, bias=lw.out_b, res=xq)
        hq = norm(xq, lw.ln2_w, lw.ln2_b, "vit_hq", lw.fc1_w)
        if self.fp8_mx and isinstance(lw.fc1_w, Fp8Weight) and isinstance(lw.fc2_w, Fp8Weight):
            (mlp_mx16 if r16 else self._mlp_fp8_mx)(hq, lw.fc1_w, lw.fc1_b, lw.fc2_w, lw.fc2_b, xq, "vit_c")
            return xq, G, cmap, Pc
        fc = self._buf("vit_fc", (Mc, Fvp), vdt)
        self._gemm(hq, lw.fc1_w, fc, mode=K.GEMM_STORE16, bias=lw.fc1_b, act=K.ACT_GELU_TANH)
        if r16:
            update(fc, lw.fc2_w, lw.fc2_b, xq, "vit_fc", Fvp)
        else:
            self._gemm(self._operand(fc, "vit_fc", Fvp, lw.fc2_w), lw.fc2_w, xq, mode=K.GEMM_F32, bias=lw.fc2_b, res=xq)
        return xq, G, cmap, Pc

    # ---- A9: pool + projector, written straight into rows of the LLM residual stream ------------------
    def project_into(self, feats: torch.Tensor, T: int, G: int, hw: int, dst_rows: torch.Tensor,
                     cmap: Optional[torch.Tensor] = None, pc: int = 0) -> None:
        cfg, w = self.cfg, self.w
        Dv, Dvp, Dp = cfg.vision.hidden_size, pad64(cfg.vision.hidden_size), pad64(cfg.hidden_size)
        nv = T * hw * hw
        if self.operand_split:
            if feats.dtype != torch.float32:
                raise ValueError("the operand-split tier pools fp32 tower features")
            pooled2 = self._buf("pooled2", (nv, 2 * Dvp), torch.float16, zero=True)
            K.bilinear_pool_split(feats, pooled2, T, G, hw, Dv, cmap, pc)
            g2 = self._buf("proj_g2", (nv, 2 * Dp), torch.float16)
            K.gemm_split16(pooled2, self.sw.proj0, g2, mode=K.GEMM_STORE16, lo_off=Dp, bias=w.proj0_b, act=K.ACT_GELU_ERF)
            self._gemm(g2, self.sw.proj2, dst_rows, mode=K.GEMM_F32, bias=w.proj2_b)
            return
        pooled = self._buf("pooled", (nv, Dvp), self.dtype)
        (K.bilinear_pool if feats.dtype == torch.float32 else K.bilinear_pool16)(feats, pooled, T, G, hw, Dv, cmap, pc)   # (16-bit tower stream)
        g = self._buf("proj_g", (nv, Dp), self.dtype)
        K.gemm(pooled, w.proj0_w, g, mode=K.GEMM_STORE16, bias=w.proj0_b, act=K.ACT_GELU_ERF)
        K.gemm(g, w.proj2_w, dst_rows, mode=K.GEMM_F32, bias=w.proj2_b)

    # ---- operand-split tier (csrc/split.hip): the same forward with every MFMA operand as an fp16 (hi, lo) pair --------------------------------
    def vit_forward_split(self, pix: torch.Tensor, hw: Optional[int] = None):
        """A7 on split operands -> (fp32 features, G, cmap, Pc) as vit_forward: dense [T*P, Dv] without a pooled grid, else the LAST needed layer
        computes K|V for all tokens and queries / out-proj / MLP only for the Pc tokens the pool reads (compact rows + cmap).  fp32 residual stream;
        LayerNorm writes [hi | lo] rows, the qkv / fc1 GEMMs write (hi, lo) pairs from their epilogues, attention runs the 3-term products, out-proj /
        fc2 add into the fp32 stream.  The patch embedding takes the 16-bit pixel tensor as it is (pixels and weights are 16-bit exact: nothing to split)."""
        v, w, sw = self.cfg.vision, self.w, self.sw
        T, _, R, _ = pix.shape
        G = R // v.patch_size
        P = G * G
        M = T * P
        Dv, Dvp, Fvp = v.hidden_size, pad64(v.hidden_size), pad64(v.intermediate_size)
        H, hd, eps = v.num_attention_heads, v.head_dim, v.layer_norm_eps
        f16 = torch.float16
        if pix.dtype != f16:
            pix = pix.to(f16)           # the model's input IS the fp16 tensor (infer.py:55: preprocess(...)["pixel_values"].half()); patches enter the MFMA as they are
        self.vit_stream_bytes = 4
        x = self._buf("vit_x", (M, Dv), torch.float32)
        a0 = K.patchify(pix, v.patch_size, w.patch_w.shape[1], f16)
        self._gemm(a0, w.patch_w, x, mode=K.GEMM_F32, bias=w.patch_b, res=w.pos_table(G), res_mod=P)
        del a0
        h = self._buf("vit_h2", (M, 2 * Dvp), f16, zero=True)
        qkv = self._buf("vit_qkv2", (M, 6 * Dv), f16)                     # [q k v hi | q k v lo]
        o = self._buf("vit_o2", (M, 2 * Dvp), f16, zero=True)
        f = self._buf("vit_f2", (M, 2 * Fvp), f16)
        st = (P * 6 * Dv, 6 * Dv, hd)
        scale = hd ** -0.5
        src = self.pool_sources(G, hw) if (hw is not None and self.skip_dead_rows and len(w.vit) > 0) else None
        layers = list(zip(w.vit, sw.vit))
        for lw, l2 in (layers if src is None else layers[:-1]):
            K.norm_split(x, lw.ln1_w, lw.ln1_b, h, Dv, eps)
            K.gemm_split16(h, l2["qkv"], qkv, mode=K.GEMM_STORE16, lo_off=3 * Dv, bias=lw.qkv_b)
            K.attention_split(qkv, qkv[:, Dv:], qkv[:, 2 * Dv:], o, lo=(3 * Dv, 3 * Dv, 3 * Dv, Dvp), batch=T, heads=H, kv_heads=H, seq=P, head_dim=hd,
                              q_strides=st, k_strides=st, v_strides=st, o_strides=(P * 2 * Dvp, 2 * Dvp, hd), scale=scale)
            self._gemm(o, l2["out"], x, mode=K.GEMM_F32, bias=lw.out_b, res=x)
            K.norm_split(x, lw.ln2_w, lw.ln2_b, h, Dv, eps)
            K.gemm_split16(h, l2["fc1"], f, mode=K.GEMM_STORE16, lo_off=Fvp, bias=lw.fc1_b, act=K.ACT_GELU_TANH)
            self._gemm(f, l2["fc2"], x, mode=K.GEMM_F32, bias=lw.fc2_b, res=x)
        if src is None:
            return x, G, None, 0
        # ---- last needed layer on compact query rows (as vit_forward: K|V for every token, everything else for the Pc tokens the pool reads) ----
        idx, cmap = src
        Pc = idx.numel()
        Mc = T * Pc
        lw, l2 = layers[-1]
        K.norm_split(x, lw.ln1_w, lw.ln1_b, h, Dv, eps)
        K.gemm_split16(h, l2["qkv"][Dv:], qkv[:, Dv:], mode=K.GEMM_STORE16, lo_off=3 * Dv, bias=lw.qkv_b[Dv:])
        xq = self._buf("vit_xq", (Mc, Dv), torch.float32)
        K.gather_rows(x, idx, P, T, xq, Dv)
        hq = self._buf("vit_hq2", (Mc, 2 * Dvp), f16, zero=True)
        K.norm_split(xq, lw.ln1_w, lw.ln1_b, hq, Dv, eps)                  # LayerNorm is per row: normalising the gathered rows = gathering the normalised rows
        qc = self._buf("vit_qc2", (Mc, 2 * Dv), f16)
        K.gemm_split16(hq, l2["qkv"][:Dv], qc, mode=K.GEMM_STORE16, lo_off=Dv, bias=lw.qkv_b[:Dv])
        oc = self._buf("vit_oc2", (Mc, 2 * Dvp), f16, zero=True)
        K.attention_split(qc, qkv[:, Dv:], qkv[:, 2 * Dv:], oc, lo=(Dv, 3 * Dv, 3 * Dv, Dvp), batch=T, heads=H, kv_heads=H, seq=Pc, kv_seq=P, head_dim=hd,
                          q_strides=(Pc * 2 * Dv, 2 * Dv, hd), k_strides=st, v_strides=st, o_strides=(Pc * 2 * Dvp, 2 * Dvp, hd), scale=scale)
        self._gemm(oc, l2["out"], xq, mode=K.GEMM_F32, bias=lw.out_b, res=xq)
        K.norm_split(xq, lw.ln2_w, lw.ln2_b, hq, Dv, eps)
        fc = self._buf("vit_fc2", (Mc, 2 * Fvp), f16)
        K.gemm_split16(hq, l2["fc1"], fc, mode=K.GEMM_STORE16, lo_off=Fvp, bias=lw.fc1_b, act=K.ACT_GELU_TANH)
        self._gemm(fc, l2["fc2"], xq, mode=K.GEMM_F32, bias=lw.fc2_b, res=xq)
        return xq, G, cmap, Pc

    def llm_forward_split(self, x: torch.Tensor, pos0: int = 0, total: Optional[int] = None, kv_gather=None) -> torch.Tensor:
        """A11 on split operands: fp32 x [N, D] (in place) -> final-normed fp32.  Context-parallel use (videoitg_amd.parallel): pos0 / total /
        kv_gather as in llm_forward - the per-layer exchange carries the local K|V rows as PAIRS ([k v hi | k v lo], twice the bytes) through the blocking
        all-gather (the asynchronous in-place form is the default tier's)."""
        cfg, w, sw = self.cfg, self.w, self.sw
        N, D = x.shape
        total = N if total is None else total
        Dp, Fp = pad64(D), pad64(cfg.intermediate_size)
        nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        qkv_n, op, kvw = (nq + 2 * nkv) * hd, pad64(nq * hd), 2 * nkv * hd
        f16 = torch.float16
        h = self._buf("llm_h2", (N, 2 * Dp), f16, zero=True)
        qkv = self._buf("llm_qkv2", (N, 2 * qkv_n), f16)
        o = self._buf("llm_o2", (N, 2 * op), f16, zero=True)
        f = self._buf("llm_f2", (N, 2 * Fp), f16)
        kv_loc = self._buf("llm_kv2", (N, 2 * kvw), f16) if kv_gather is not None else None
        cos, sin = self.rope_tables(total)
        cos, sin = cos[pos0:pos0 + N], sin[pos0:pos0 + N]
        st = (0, 2 * qkv_n, hd)
        for lw, l2 in zip(w.llm, sw.llm):
            K.norm_split(x, lw.in_ln, None, h, D, cfg.rms_norm_eps)
            K.gemm_split16(h, l2["qkv"], qkv, mode=K.GEMM_STORE16, lo_off=qkv_n, bias=lw.qkv_b)
            K.rope_split_(qkv, cos, sin, nq + nkv, hd, qkv_n)
            if kv_gather is None:
                K.attention_split(qkv, qkv[:, nq * hd:], qkv[:, (nq + nkv) * hd:], o, lo=(qkv_n, qkv_n, qkv_n, op), batch=1, heads=nq, kv_heads=nkv, seq=N,
                                  head_dim=hd, q_strides=st, k_strides=st, v_strides=st, o_strides=(0, 2 * op, hd), scale=hd ** -0.5)
            else:
                kv_loc[:, :kvw].copy_(qkv[:, nq * hd:qkv_n])                      # [k v hi | k v lo] rows of this rank
                kv_loc[:, kvw:].copy_(qkv[:, qkv_n + nq * hd:])
                kv = kv_gather(kv_loc)                                            # [total, 2 kvw], blocking (RowGatherer.__call__ or any callable)
                K.attention_split(qkv, kv, kv[:, nkv * hd:], o, lo=(qkv_n, kvw, kvw, op), batch=1, heads=nq, kv_heads=nkv, seq=N, kv_seq=total, head_dim=hd,
                                  q_strides=st, k_strides=(0, 2 * kvw, hd), v_strides=(0, 2 * kvw, hd), o_strides=(0, 2 * op, hd), scale=hd ** -0.5)
            self._gemm(o, l2["o"], x, mode=K.GEMM_F32, res=x)
            K.norm_split(x, lw.post_ln, None, h, D, cfg.rms_norm_eps)
            K.gemm_split16(h, l2["gu"], f, mode=K.GEMM_SWIGLU, lo_off=Fp)
            self._gemm(f, l2["down"], x, mode=K.GEMM_F32, res=x)
        out = self._buf("llm_out", (N, D), torch.float32)
        K.rmsnorm(x, w.norm_w, out, D, cfg.rms_norm_eps)
        return out

    def rope_tables(self, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """fp32 cos/sin [n, hd/2], inv_freq = theta^(-2i/hd), positions 0..n-1 ([3P] Qwen2RotaryEmbedding;
        position_ids=None -> arange, eagle_archv1.py:396-397).  Built in fp32 on the host, cached per length."""
        if n not in self._rope:
            hd = self.cfg.head_dim
            inv = 1.0 / (self.cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
            fr = torch.outer(torch.arange(n, dtype=torch.float32), inv)
            self._rope[n] = (fr.cos().to(self.device).contiguous(), fr.sin().to(self.device).contiguous())
        return self._rope[n]

    # ---- A11: bidirectional Qwen2 stack over fp32 x [N, D] (in place) -> final-normed fp32 -------------
    def llm_forward(self, x: torch.Tensor, pos0: int = 0, total: Optional[int] = None, kv_gather=None,
                    kv_len: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: this rank's rows (all rows on one GPU).  Context-parallel use (videoitg_amd.parallel): pos0 = global
        position of row 0, total = global token count, kv_gather(local K|V rows) -> all K|V rows.
        kv_len (int32 device scalar, single-GPU path): only the first kv_len rows are real tokens - the rest is bucket padding
        that no query may attend to (a captured graph serves every prompt length of its bucket)."""
        if self.operand_split:
            if kv_len is not None:
                raise NotImplementedError("the operand-split tier runs eager forwards (no bucketed graph capture: its attention takes no device key count)")
            return self.llm_forward_split(x, pos0=pos0, total=total, kv_gather=kv_gather)
        cfg, w = self.cfg, self.w
        N, D = x.shape
        total = N if total is None else total
        Dp, Fp = pad64(D), pad64(cfg.intermediate_size)
        nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        qkv_n = (nq + 2 * nkv) * hd
        op = pad64(nq * hd)
        qkv = self._buf("llm_qkv", (N, qkv_n), self.dtype)
        o = self._buf("llm_o", (N, op), self.dtype, zero=True)
        mx_mlp = self.fp8_mx and all(isinstance(lw.gu_w, Fp8Weight) and isinstance(lw.down_w, Fp8Weight) for lw in w.llm)
        f = None if mx_mlp else self._buf("llm_f", (N, Fp), self.dtype)
        cos, sin = self.rope_tables(total)
        cos, sin = cos[pos0:pos0 + N], sin[pos0:pos0 + N]
        scale = hd ** -0.5
        kvw = 2 * nkv * hd
        pend = None                                       # layer 0 normalises the assembled embeddings with the stand-alone kernel
        fuse = kv_gather is None                          # (the context-parallel ranks take the split-K path for o / down instead)
        for li, lw in enumerate(w.llm):
            if kv_gather is not None and hasattr(kv_gather, "local_rows") and self.cp_overlap:
                # context-parallel rank: K|V FIRST, written straight into this rank's slot of the gather buffer, RoPE on the
                # K heads, exchange launched - all on a side stream - while the main stream runs the Q projection and its RoPE; the
                # attention then waits for the exchange only.  (K|V: 28 - 56 output tiles, Q: 98 at M = 1660 - the two GEMMs share the CUs.)
                h = self._norm(x, lw.in_ln, None, "llm_h", D, cfg.rms_norm_eps, cw=lw.qkv_w)
                kcol = nq * hd
                kv_local = kv_gather.local_rows(kvw, self.dtype, self.device)
                main = torch.cuda.current_stream(self.device)
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                self._side.wait_stream(main)                              # h is ready; the previous layer's attention is done with the buffer
                with torch.cuda.stream(self._side):
                    self._ws_tag = "_side"
                    try:
                        self._gemm(h, lw.qkv_w[kcol:], kv_local, mode=K.GEMM_STORE16, bias=lw.qkv_b[kcol:])
                    finally:
                        self._ws_tag = ""
                    K.rope_(kv_local, cos, sin, nkv, hd)
                    work = kv_gather.start()
                self._gemm(h, lw.qkv_w[:kcol], qkv[:, :kcol], mode=K.GEMM_STORE16, bias=lw.qkv_b[:kcol])
                K.rope_(qkv, cos, sin, nq, hd)
                kv = kv_gather.finish(work, self._side)                   # [total, 2*nkv*hd]: K heads then V heads per row
            else:
                if pend is not None:
                    K.gemm_norm_consumer(pend[0], lw.qkv_w, qkv, mode=K.GEMM_STORE16, stats=pend[1], bias=lw.qkv_b)
                else:
                    h = self._norm(x, lw.in_ln, None, "llm_h", D, cfg.rms_norm_eps, cw=lw.qkv_w)
                    self._gemm(h, lw.qkv_w, qkv, mode=K.GEMM_STORE16, bias=lw.qkv_b)
                K.rope_(qkv, cos, sin, nq + nkv, hd)
                if kv_gather is not None:                                 # (a plain callable: gathers the local K|V rows, blocking)
                    kv = kv_gather(qkv[:, nq * hd:])
            if kv_gather is None:
                K.attention(qkv, qkv[:, nq * hd:], qkv[:, (nq + nkv) * hd:], o, batch=1, heads=nq, kv_heads=nkv, seq=N,
                            head_dim=hd, q_strides=(0, qkv_n, hd), k_strides=(0, qkv_n, hd), v_strides=(0, qkv_n, hd),
                            o_strides=(0, op, hd), scale=scale, variant=self.attn_variant, kv_len=kv_len)
            else:
                K.attention(qkv, kv, kv[:, nkv * hd:], o, batch=1, heads=nq, kv_heads=nkv, seq=N, kv_seq=total,
                            head_dim=hd, q_strides=(0, qkv_n, hd), k_strides=(0, kvw, hd), v_strides=(0, kvw, hd),
                            o_strides=(0, op, hd), scale=scale, variant=self.attn_variant)
            pend = self._producer(self._operand(o, "llm_o", nq * hd, lw.o_w), lw.o_w, x, bias=None, res=x, gamma=lw.post_ln if fuse else None,
                                  name="llm", eps=cfg.rms_norm_eps, rms=True, dt=self.dtype)
            if pend is not None:
                K.gemm_norm_consumer(pend[0], lw.gu_w, f, mode=K.GEMM_SWIGLU, stats=pend[1])
            else:
                h = self._norm(x, lw.post_ln, None, "llm_h", D, cfg.rms_norm_eps, cw=lw.gu_w)
                if self.fp8_mx and isinstance(lw.gu_w, Fp8Weight) and isinstance(lw.down_w, Fp8Weight):
                    # silu(gate) * up leaves gate/up's epilogue as e4m3 + MX block scales and is down's operand as it stands
                    kp = lw.down_w.w8.shape[1]
                    f8 = self._buf("llm_f8", (N, kp), torch.uint8)
                    fsc = self._buf("llm_fsc", (kp // 128, N, 4), torch.uint8)
                    K.gemm_fp8_mx(h[0], lw.gu_w.w8, lw.gu_w.sw, f8, out_dtype=self.dtype, sa=h[1], o_mx=fsc, swiglu=True)
                    K.gemm_fp8_mx(f8, lw.down_w.w8, lw.down_w.sw, x, out_dtype=self.dtype, a_mx=fsc, res=x)
                    pend = None
                    continue
                self._gemm(h, lw.gu_w, f, mode=K.GEMM_SWIGLU)
            nxt = w.llm[li + 1].in_ln if (fuse and li + 1 < len(w.llm)) else None
            pend = self._producer(self._operand(f, "llm_f", Fp, lw.down_w), lw.down_w, x, bias=None, res=x, gamma=nxt, name="llm",
                                  eps=cfg.rms_norm_eps, rms=True, dt=self.dtype)
        out = self._buf("llm_out", (N, D), torch.float32)
        K.rmsnorm(x, w.norm_w, out, D, cfg.rms_norm_eps)
        return out

    # ---- whole path for one sample whose frames are already encoded ---------------------------------
    def score_video(self, feats: torch.Tensor, T: int, G: int, ids: torch.Tensor, cmap: Optional[torch.Tensor] = None,
                    pc: int = 0, ids_dev: Optional[torch.Tensor] = None, kv_len: Optional[torch.Tensor] = None) -> torch.Tensor:
        """feats fp32 [T*P, Dv] (this video's rows), ids HOST int64 [L] (unpadded, exactly one IMAGE_TOKEN_INDEX) ->
        logits fp32 [T].  ids_dev: the same ids already on the device (graph capture: no host-to-device copy may be
        recorded; only the text VALUES are read from it, the layout comes from the host copy).  kv_len: see llm_forward - with it,
        `ids` is the BUCKET-PADDED prompt (pad ids after the real text) and the device scalar holds the real token count.
        The one-<image> form of score_sample (what infer.py, the harness plugin, graph capture and the context-parallel path use)."""
        if int((ids.cpu() == IMAGE_TOKEN_INDEX).sum()) != 1:
            raise ValueError("score_video takes exactly one <image> token; samples with none or several go through score_sample")
        return self.score_sample([(feats, T, cmap, pc)], G, ids, T, ids_dev=ids_dev, kv_len=kv_len)

    def score_sample(self, blocks, G: int, ids: torch.Tensor, t_head: int, ids_dev: Optional[torch.Tensor] = None,
                     kv_len: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One row of the batch (eagle_archv1.py:280-337 + grounding_qwen2.py:146-156).  `blocks` = one (feats fp32, T_i, cmap, pc)
        per <image> token of `ids`, in order; every block is pooled with ITS OWN grid (mlp_proj.py:48 works per list entry) and
        spliced between the text segments.  t_head = images[b].shape[0], the frame count the reference's head reshapes this row's
        visual tokens by - for a multi-<image> row that is NOT the row's own clip list (the reference indexes `images` by the batch
        row), so all nvis visual rows are pooled in t_head equal groups; nvis % t_head != 0 fails like the reference's reshape
        (RuntimeError) and a row without <image> gives NaN logits like the reference's mean over an empty group
        (tests/golden/helpers.json "forward_errors")."""
        cfg = self.cfg
        ids = ids.cpu()
        pos = (ids == IMAGE_TOKEN_INDEX).nonzero().flatten().tolist()
        if len(pos) != len(blocks):
            raise ValueError(f"{len(pos)} <image> tokens but {len(blocks)} visual blocks for this sample")
        logits = torch.empty(t_head, dtype=torch.float32, device=self.device)
        if not pos:
            return logits.fill_(float("nan"))                       # no visual rows: the decoder output is never read by the head
        hws = [(b[4] if len(b) > 4 and b[4] is not None else cfg.pooled_grid(b[1], G)) for b in blocks]      # (explicit grid: training-mode pooling)
        nvs = [b[1] * hw * hw for b, hw in zip(blocks, hws)]
        bounds = [-1] + pos + [ids.numel()]
        segs = [(bounds[i] + 1, bounds[i + 1]) for i in range(len(bounds) - 1)]      # text between / around the <image> tokens (ids index ranges)
        n = sum(e - s for s, e in segs) + sum(nvs)
        limit = cfg.tokenizer_model_max_length
        if limit is not None and n > limit:                             # eagle_archv1.py:341-345 truncates the tail
            if limit < n - (segs[-1][1] - segs[-1][0]):
                raise ValueError("tokenizer_model_max_length truncates visual tokens: the reference's per-frame reshape "
                                 "(grounding_qwen2.py:153) fails in this case too")
            segs[-1] = (segs[-1][0], segs[-1][1] - (n - limit))
            n = limit
        nvis = sum(nvs)
        if nvis % t_head:
            raise RuntimeError(f"shape '[{t_head}, -1, {cfg.hidden_size}]' is invalid for input of size {nvis * cfg.hidden_size} "
                               "(grounding_qwen2.py:153: the row's visual tokens do not divide by images[b].shape[0])")
        x = self._buf("llm_x", (n, cfg.hidden_size), torch.float32)
        row, vis_rows = 0, []
        for i, (s0, e0) in enumerate(segs):
            if e0 > s0:
                src = ids_dev[s0:e0] if ids_dev is not None else ids[s0:e0].contiguous().to(self.device)
                K.embed_gather(src, self.w.embed, x[row:row + e0 - s0])
                row += e0 - s0
            if i < len(blocks):
                feats, T, cmap, pc = blocks[i][:4]
                self.project_into(feats, T, G, hws[i], x[row:row + nvs[i]], cmap, pc)
                vis_rows.append((row, nvs[i]))
                row += nvs[i]
        hfin = self.llm_forward(x, kv_len=kv_len)
        if len(vis_rows) == 1:
            hv = hfin[vis_rows[0][0]:]                                  # contiguous: the head reads the rows in place
        else:
            # hidden_states[b][token_types == 3]: the visual rows compacted in sequence order (text sits between the blocks)
            idx = torch.cat([torch.arange(r0, r0 + nv, dtype=torch.int32) for r0, nv in vis_rows]).to(self.device)
            hv = self._buf("llm_hv", (nvis, cfg.hidden_size), torch.float32)
            K.gather_rows(hfin, idx, n, 1, hv, cfg.hidden_size)
        K.pool_score(hv, t_head, nvis // t_head, cfg.hidden_size, self.w.head_w, self.w.head_b, logits)
        return logits

    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], images: Sequence[torch.Tensor],
                pooled_grids: Optional[Sequence[int]] = None) -> List[torch.Tensor]:
        """EagleQwenG.forward (grounding_qwen2.py:86-181), inference branch: list of B tensors [images[b].shape[0], 1] (fp32 logits).
        Padded batches are run sample by sample on the unpadded ids (eagle_archv1.py:272 strips padding; the
        padded rows never influence real ones because keys are masked), which is value-identical.  A sample consumes one entry
        of `images` per <image> token - and one entry when it has none (eagle_archv1.py:282-294) - in batch order.
        pooled_grids: one pooled grid HW per consumed entry of `images` instead of the eval rule min(floor(sqrt(vision_token_num / T)), G) -
        the projector's TRAINING-mode behaviour (mlp_proj.py:52: a random HW per video; the model wrapper draws it)."""
        images = [im if im.dim() == 4 else im.unsqueeze(0) for im in images]
        if input_ids.dim() == 1:
            input_ids = input_ids.unsqueeze(0)
        B = input_ids.shape[0]
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
        ids_host = input_ids.cpu()
        mask_host = attention_mask.bool().cpu()
        samples = [ids_host[b][mask_host[b]] for b in range(B)]          # eagle_archv1.py:272
        counts = [int((s_ == IMAGE_TOKEN_INDEX).sum()) for s_ in samples]
        need = sum(max(c, 1) for c in counts)
        if len(images) < max(need, B):
            raise ValueError(f"{len(images)} videos for a batch of {B} prompts holding {sum(counts)} <image> tokens")
        used = images[:need]                                            # (entries past the consumed ones are never read by the reference either)
        sizes = {tuple(im.shape[1:]) for im in used}
        if len(sizes) != 1:
            raise ValueError("all videos in a batch must share one frame size")
        pix = torch.cat([im.to(self.device) for im in used], 0) if len(used) > 1 else used[0].to(self.device)
        G0 = pix.shape[-1] // self.cfg.vision.patch_size
        if pooled_grids is not None:
            if len(pooled_grids) < len(used):
                raise ValueError(f"{len(pooled_grids)} pooled grids for {len(used)} videos")
            hw_list = [min(int(h), G0) for h in pooled_grids[:len(used)]]
            if min(hw_list) < 1:
                raise ValueError("pooled grid must be >= 1")
        else:
            hw_list = [self.cfg.pooled_grid(im.shape[0], G0) for im in used]
        hws = set(hw_list)
        hw_all = hws.pop() if len(hws) == 1 else None                 # videos of different length pool differently -> dense
        feats, G, cmap, pc = self.vit_forward(pix, hw_all)
        P = pc if cmap is not None else G * G
        rows0, r = [], 0
        for im in used:
            rows0.append(r)
            r += im.shape[0]
        out, img = [], 0
        for b in range(B):
            blocks = [(feats[rows0[i] * P:(rows0[i] + used[i].shape[0]) * P], used[i].shape[0], cmap, pc, hw_list[i]) for i in range(img, img + counts[b])]
            img += max(counts[b], 1)
            t_head = images[b].shape[0]                                  # grounding_qwen2.py:153 indexes `images` by the batch row
            out.append(self.score_sample(blocks, G, samples[b], t_head).reshape(t_head, 1))
        return out


def prompt_bucket(n_post: int, bucket: int) -> int:
    """text tokens after <image>, rounded up to the capture bucket (at least one bucket)"""
    return max(bucket, (n_post + bucket - 1) // bucket * bucket)


class GraphedForward:
    """One video's selector forward captured in a HIP graph (torch.cuda.CUDAGraph) and replayed: ~850 kernel launches become
    one graph launch (BASELINE.json configs[3]: "hipGraph-captured per-video forward").

    The capture is keyed by what fixes the launch geometry: frame count, resolution, the number of text tokens before <image> and
    the BUCKET of the text length after it.  Pixel values, prompt token VALUES and the real token count are read from static
    device buffers refreshed before every replay; rows between the real prompt end and the bucket end are padding that the
    attention kernels exclude by length (vitg_attention_dyn reads the key count from device memory), so one capture serves every
    prompt of its bucket with results identical to the unpadded eager forward (SURVEY.md section 7 hard part 6)."""

    def __init__(self, eng: "SelectorEngine", input_ids: torch.Tensor, pix: torch.Tensor, bucket: int = 32):
        self.eng, self.bucket = eng, bucket
        ids = input_ids.reshape(-1).cpu()
        self.key = self.layout_key(eng, ids, pix, bucket)
        n_pre, n_post_b, _ = self.key
        T = pix.shape[0]
        G = pix.shape[-1] // eng.cfg.vision.patch_size
        self.nv = T * eng.cfg.pooled_grid(T, G) ** 2
        lim = eng.cfg.tokenizer_model_max_length
        if lim is not None and n_pre + self.nv + n_post_b > lim:
            raise ValueError("prompt bucket exceeds tokenizer_model_max_length; use the eager forward (it truncates like the reference)")
        # host-side layout of the padded prompt: [pre][<image>][post padded to the bucket]; the values live on the device
        self.ids_layout = torch.cat([torch.zeros(n_pre, dtype=torch.long), torch.tensor([IMAGE_TOKEN_INDEX]),
                                     torch.zeros(n_post_b, dtype=torch.long)])
        self.ids_dev = torch.zeros(n_pre + 1 + n_post_b, dtype=torch.long, device=eng.device)
        self.kv_len = torch.zeros(1, dtype=torch.int32, device=eng.device)
        self.pix = torch.empty_like(pix, device=eng.device)
        self._stage(ids, pix)
        self._run()                                                       # eager warm-up: allocates every workspace buffer
        torch.cuda.synchronize(eng.device)
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=eng.device)
        side.wait_stream(torch.cuda.current_stream(eng.device))
        with torch.cuda.stream(side):
            self._run()
        torch.cuda.current_stream(eng.device).wait_stream(side)
        with torch.cuda.graph(self.graph):
            self.out = self._run()
        self._held = eng.workspace_tensors()                              # the graph replays on exactly these allocations
        assert self.out.shape == (T,)

    @staticmethod
    def layout_key(eng, ids: torch.Tensor, pix: torch.Tensor, bucket: int):
        ids = ids.reshape(-1)
        pos = (ids == IMAGE_TOKEN_INDEX).nonzero().flatten().tolist()
        if len(pos) != 1:
            raise ValueError(f"exactly one <image> token per sample is supported on the selector path, got {len(pos)}")
        return (pos[0], prompt_bucket(ids.numel() - pos[0] - 1, bucket), tuple(pix.shape))

    def _stage(self, ids: torch.Tensor, pix: torch.Tensor) -> None:
        """refresh the static inputs (stream-ordered copies; nothing here is captured)"""
        n_pre, n_post_b, _ = self.key
        ids = ids.reshape(-1)
        padded = torch.zeros(n_pre + 1 + n_post_b, dtype=torch.long)
        padded[:ids.numel()] = ids.clamp(min=0)                            # the <image> slot is never read
        self.ids_dev.copy_(padded, non_blocking=True)
        self.kv_len.fill_(n_pre + self.nv + ids.numel() - n_pre - 1)
        self.pix.copy_(pix, non_blocking=True)

    def _run(self) -> torch.Tensor:
        eng = self.eng
        T = self.pix.shape[0]
        G0 = self.pix.shape[-1] // eng.cfg.vision.patch_size
        hw = eng.cfg.pooled_grid(T, G0)
        feats, G, cmap, pc = eng.vit_forward(self.pix, hw)
        return eng.score_video(feats, T, G, self.ids_layout, cmap, pc, ids_dev=self.ids_dev, kv_len=self.kv_len)

    def matches(self, input_ids: torch.Tensor, pix: torch.Tensor) -> bool:
        try:
            return self.key == self.layout_key(self.eng, input_ids.cpu(), pix, self.bucket)
        except ValueError:
            return False

    def __call__(self, input_ids: torch.Tensor, pix: torch.Tensor) -> torch.Tensor:
        """-> logits fp32 [T] (a fresh tensor); the layout must be inside the captured bucket"""
        input_ids = input_ids.cpu()
        if not self.matches(input_ids, pix):
            raise ValueError("GraphedForward: frame count, resolution, <image> position and prompt-length bucket must equal the captured ones")
        self._stage(input_ids, pix)
        self.graph.replay()
        return self.out.clone()


class GraphCache:
    """Captured forwards keyed by layout (frames, resolution, <image> position, prompt bucket), least-recently-used eviction.
    A harness run over one benchmark (fixed num_frames, prompts of 20-120 tokens) needs a handful of captures."""

    def __init__(self, eng: "SelectorEngine", bucket: int = 32, max_entries: int = 8):
        self.eng, self.bucket, self.max_entries = eng, bucket, max_entries
        self._graphs: Dict[tuple, GraphedForward] = {}
        self.captures = 0

    def serves(self, input_ids: torch.Tensor, pix: torch.Tensor) -> bool:
        """can a captured forward score this document?  Exactly one <image> token and a bucket-padded prompt inside
        tokenizer_model_max_length; everything else belongs to the eager forward (which truncates like the reference)"""
        try:
            n_pre, n_post_b, shape = GraphedForward.layout_key(self.eng, input_ids.cpu(), pix, self.bucket)
        except ValueError:
            return False
        cfg = self.eng.cfg
        T, G = shape[0], shape[-1] // cfg.vision.patch_size
        lim = cfg.tokenizer_model_max_length
        return lim is None or n_pre + T * cfg.pooled_grid(T, G) ** 2 + n_post_b <= lim

    def __call__(self, input_ids: torch.Tensor, pix: torch.Tensor) -> torch.Tensor:
        input_ids = input_ids.cpu()                                        # one host copy, shared by the key, the match and the staging
        key = GraphedForward.layout_key(self.eng, input_ids, pix, self.bucket)
        g = self._graphs.pop(key, None)
        if g is None:
            if len(self._graphs) >= self.max_entries:
                self._graphs.pop(next(iter(self._graphs)))
            g = GraphedForward(self.eng, input_ids, pix, self.bucket)
            self.captures += 1
        self._graphs[key] = g                                              # (re-)inserted last = most recently used
        return g(input_ids, pix)
