"""``torch.ops.videoitg.*`` — the kernels of libvideoitg_hip.so registered with the PyTorch dispatcher (SURVEY.md §8b last
row: "PyTorch-ROCm custom ops ... schema-typed, stream-ordered on the current HIP stream, no host sync, errors ->
RuntimeError").

A thin layer: every op's CUDA(=HIP) implementation is the ctypes wrapper of ``videoitg_amd.ops`` over the same C ABI
(include/videoitg_hip.h); nothing is computed here.  Schemas carry the mutation annotations (``Tensor(a!)``) so that the
dispatcher, functionalisation and ``torch.compile`` see which buffers an op writes; Meta ("fake") implementations give the
output shapes so the ops trace without running.  There is deliberately NO CPU registration: calling an op on host tensors
raises ``NotImplementedError`` from the dispatcher (the product has no CPU path).

Aliasing: the engine passes ``res`` aliasing ``out`` (x = a @ w.T + x) to gemm / gemm_fp8 / gemm_fp8_mx / gemm_res16; only ``out`` carries
the mutation annotation, so under functionalisation (``torch.compile``, ``torch.func.functionalize``) an aliased ``res`` could be read
from a stale clone.  These ops are therefore **eager-only contracts with Meta kernels for shape tracing**: the product never compiles
or functionalises them (it captures HIP graphs of the eager launches instead), and ``tests/test_torch_ops.py`` pins the eager
aliasing behaviour.

``SelectorEngine`` launches through this module (``videoitg_amd.engine.K``); ``videoitg_amd.ops`` stays the raw binding.
Names are the build's choice: patchify, gemm, layernorm, rmsnorm, rope_, attention, bilinear_pool,
gather_rows, embed_gather, pool_score, rank, preprocess_u8, bce_loss, quant_rows_fp8, norm_quant_fp8, gemm_fp8, gemm_fp8_mx,
gemm_res16 / layernorm16 / bilinear_pool16 (the tower's 16-bit residual stream).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops as _raw

_lib = torch.library.Library("videoitg", "DEF")
_T = torch.Tensor


def _def(schema: str, impl, fake=None):
    name = schema.split("(", 1)[0]
    _lib.define(schema)
    _lib.impl(name, impl, "CUDA")
    if fake is not None:
        torch.library.register_fake(f"videoitg::{name}", fake)


_DT = {0: torch.bfloat16, 1: torch.float16, 2: torch.float32}

# ---- patch embedding ---------------------------------------------------------------------------------------------------
_def("patchify(Tensor pix, int patch, int kpad, int dtype) -> Tensor",
     lambda pix, patch, kpad, dtype: _raw.patchify(pix, patch, kpad, _DT[dtype]),
     lambda pix, patch, kpad, dtype: pix.new_empty((pix.shape[0] * (pix.shape[2] // patch) ** 2, kpad), dtype=_DT[dtype]))

# ---- GEMM with fused epilogue --------------------------------------------------------------------------------------------
_def("gemm(Tensor a, Tensor w, Tensor(a!) out, int mode, int? n=None, Tensor? bias=None, int act=0, Tensor? res=None, "
     "int res_mod=0, int? m=None) -> ()",
     lambda a, w, out, mode, n=None, bias=None, act=0, res=None, res_mod=0, m=None:
         (_raw.gemm(a, w, out, mode=mode, n=n, bias=bias, act=act, res=res, res_mod=res_mod, m=m), None)[1],
     lambda a, w, out, mode, n=None, bias=None, act=0, res=None, res_mod=0, m=None: None)

_def("gemm_res16(Tensor a, Tensor w, Tensor(a!) out, Tensor? bias=None, Tensor? res16=None, Tensor? res32=None, int res_mod=0) -> ()",
     lambda a, w, out, bias=None, res16=None, res32=None, res_mod=0:
         (_raw.gemm_res16(a, w, out, bias=bias, res16=res16, res32=res32, res_mod=res_mod), None)[1],
     lambda *a, **k: None)

_def("patch_embed16(Tensor pix, Tensor wf, Tensor(a!) out, int patch, Tensor? bias=None, Tensor? pos=None) -> ()",
     lambda pix, wf, out, patch, bias=None, pos=None: (_raw.patch_embed16(pix, wf, out, patch=patch, bias=bias, pos=pos), None)[1],
     lambda *a, **k: None)

_def("gemm_splitk16(Tensor a, Tensor w, Tensor(a!) out, Tensor(b!) workspace, int splits, Tensor? bias=None) -> ()",
     lambda a, w, out, workspace, splits, bias=None: (_raw.gemm_splitk16(a, w, out, workspace, splits, bias=bias), None)[1],
     lambda a, w, out, workspace, splits, bias=None: None)
_def("gemm_splitk(Tensor a, Tensor w, Tensor(a!) out, Tensor(b!) workspace, int splits, Tensor? bias=None, Tensor? res=None) -> ()",
     lambda a, w, out, workspace, splits, bias=None, res=None: (_raw.gemm_splitk(a, w, out, workspace, splits, bias=bias, res=res), None)[1],
     lambda *a, **k: None)

# ---- norm folded into the GEMMs around it (profiling library only: measured slower than the stand-alone norm kernels)
if _raw._lib.PROFILING:
  # stream-K (round 3) lost its A/B against split-K x2 on the per-rank shapes: profiling library only since round 4
  _def("gemm_streamk(Tensor a, Tensor w, Tensor(a!) out, Tensor(b!) workspace, int blocks, Tensor? bias=None, Tensor? res=None) -> ()",
       lambda a, w, out, workspace, blocks, bias=None, res=None: (_raw.gemm_streamk(a, w, out, workspace, blocks, bias=bias, res=res), None)[1],
       lambda *a, **k: None)
  _def("gemm_norm_producer(Tensor a, Tensor w, Tensor(a!) out, Tensor res, Tensor next_gamma, Tensor(b!) xh, Tensor(c!) partials, "
       "Tensor? bias=None, int res_mod=0) -> ()",
       lambda a, w, out, res, next_gamma, xh, partials, bias=None, res_mod=0:
           (_raw.gemm_norm_producer(a, w, out, res=res, next_gamma=next_gamma, xh=xh, partials=partials, bias=bias, res_mod=res_mod), None)[1],
       lambda *a, **k: None)
  _def("norm_stats_finalize(Tensor partials, int M, int P, int cols, float eps, bool rms, Tensor(a!) stats) -> ()",
       lambda partials, M, P, cols, eps, rms, stats: (_raw.norm_stats_finalize(partials, M, P, cols, eps, rms, stats), None)[1],
       lambda *a, **k: None)
  _def("gemm_norm_consumer(Tensor xh, Tensor w, Tensor(a!) out, int mode, Tensor stats, Tensor? colsum=None, Tensor? bias=None, "
       "int act=0, int? n=None) -> ()",
       lambda xh, w, out, mode, stats, colsum=None, bias=None, act=0, n=None:
           (_raw.gemm_norm_consumer(xh, w, out, mode=mode, stats=stats, colsum=colsum, bias=bias, act=act, n=n), None)[1],
       lambda *a, **k: None)

# ---- norms / rope ----------------------------------------------------------------------------------------------------------
_def("layernorm(Tensor x, Tensor w, Tensor b, Tensor(a!) out, int d, float eps) -> ()",
     lambda x, w, b, out, d, eps: (_raw.layernorm(x, w, b, out, d, eps), None)[1], lambda x, w, b, out, d, eps: None)
_def("layernorm16(Tensor x, Tensor w, Tensor b, Tensor(a!) out, int d, float eps) -> ()",
     lambda x, w, b, out, d, eps: (_raw.layernorm16(x, w, b, out, d, eps), None)[1], lambda x, w, b, out, d, eps: None)
_def("rmsnorm(Tensor x, Tensor w, Tensor(a!) out, int d, float eps) -> ()",
     lambda x, w, out, d, eps: (_raw.rmsnorm(x, w, out, d, eps), None)[1], lambda x, w, out, d, eps: None)
_def("rope_(Tensor(a!) qkv, Tensor cos, Tensor sin, int heads, int head_dim) -> ()",
     lambda qkv, cos, sin, heads, head_dim: (_raw.rope_(qkv, cos, sin, heads, head_dim), None)[1],
     lambda qkv, cos, sin, heads, head_dim: None)

# ---- operand-split precision tier (csrc/split.hip)
_def("norm_split(Tensor x, Tensor w, Tensor? b, Tensor(a!) out, int d, float eps) -> ()",
     lambda x, w, b, out, d, eps: (_raw.norm_split(x, w, b, out, d, eps), None)[1], lambda *a, **k: None)
_def("rope_split_(Tensor(a!) qkv, Tensor cos, Tensor sin, int heads, int head_dim, int lo_off) -> ()",
     lambda qkv, cos, sin, heads, head_dim, lo_off: (_raw.rope_split_(qkv, cos, sin, heads, head_dim, lo_off), None)[1], lambda *a, **k: None)
_def("bilinear_pool_split(Tensor x, Tensor(a!) out, int T, int G, int HW, int d, Tensor? cmap=None, int pc=0) -> ()",
     lambda x, out, T, G, HW, d, cmap=None, pc=0: (_raw.bilinear_pool_split(x, out, T, G, HW, d, cmap, pc), None)[1], lambda *a, **k: None)
_def("gemm_split16(Tensor a, Tensor w, Tensor(a!) out, int mode, int lo_off, Tensor? bias=None, int act=0) -> ()",
     lambda a, w, out, mode, lo_off, bias=None, act=0: (_raw.gemm_split16(a, w, out, mode=mode, lo_off=lo_off, bias=bias, act=act), None)[1],
     lambda *a, **k: None)
_def("attention_split(Tensor q, Tensor k, Tensor v, Tensor(a!) o, int[] lo, int batch, int heads, int kv_heads, int seq, int head_dim, "
     "int[] q_strides, int[] k_strides, int[] v_strides, int[] o_strides, float scale, int? kv_seq=None) -> ()",
     lambda q, k, v, o, lo, batch, heads, kv_heads, seq, head_dim, q_strides, k_strides, v_strides, o_strides, scale, kv_seq=None:
         (_raw.attention_split(q, k, v, o, lo=lo, batch=batch, heads=heads, kv_heads=kv_heads, seq=seq, head_dim=head_dim, q_strides=q_strides,
                               k_strides=k_strides, v_strides=v_strides, o_strides=o_strides, scale=scale, kv_seq=kv_seq), None)[1],
     lambda *a, **k: None)

# ---- non-causal flash attention ------------------------------------------------------------------------------------------
_def("attention(Tensor q, Tensor k, Tensor v, Tensor(a!) o, int batch, int heads, int kv_heads, int seq, int head_dim, "
     "int[] q_strides, int[] k_strides, int[] v_strides, int[] o_strides, float scale, int variant=0, int? kv_seq=None, "
     "Tensor? kv_len=None) -> ()",
     lambda q, k, v, o, batch, heads, kv_heads, seq, head_dim, q_strides, k_strides, v_strides, o_strides, scale, variant=0,
            kv_seq=None, kv_len=None:
         (_raw.attention(q, k, v, o, batch=batch, heads=heads, kv_heads=kv_heads, seq=seq, head_dim=head_dim, q_strides=q_strides,
                         k_strides=k_strides, v_strides=v_strides, o_strides=o_strides, scale=scale, variant=variant,
                         kv_seq=kv_seq, kv_len=kv_len), None)[1],
     lambda *a, **k: None)

# ---- pooling / gathers / head ------------------------------------------------------------------------------------------------
_def("bilinear_pool(Tensor x, Tensor(a!) out, int T, int G, int HW, int d, Tensor? cmap=None, int pc=0) -> ()",
     lambda x, out, T, G, HW, d, cmap=None, pc=0: (_raw.bilinear_pool(x, out, T, G, HW, d, cmap, pc), None)[1],
     lambda *a, **k: None)
_def("bilinear_pool16(Tensor x, Tensor(a!) out, int T, int G, int HW, int d, Tensor? cmap=None, int pc=0) -> ()",
     lambda x, out, T, G, HW, d, cmap=None, pc=0: (_raw.bilinear_pool16(x, out, T, G, HW, d, cmap, pc), None)[1],
     lambda *a, **k: None)
_def("gather_rows(Tensor src, Tensor idx, int rows_per_batch, int batch, Tensor(a!) out, int width) -> ()",
     lambda src, idx, rows_per_batch, batch, out, width: (_raw.gather_rows(src, idx, rows_per_batch, batch, out, width), None)[1],
     lambda *a, **k: None)
_def("embed_gather(Tensor ids, Tensor table, Tensor(a!) out) -> ()",
     lambda ids, table, out: (_raw.embed_gather(ids, table, out), None)[1], lambda *a, **k: None)
_def("pool_score(Tensor h, int frames, int tokens_per_frame, int d, Tensor w, Tensor b, Tensor(a!) logits) -> ()",
     lambda h, frames, tokens_per_frame, d, w, b, logits: (_raw.pool_score(h, frames, tokens_per_frame, d, w, b, logits), None)[1],
     lambda *a, **k: None)
_def("rank(Tensor logits) -> (Tensor, Tensor)", lambda logits: _raw.rank(logits),
     lambda logits: (logits.new_empty(logits.numel()), logits.new_empty(logits.numel(), dtype=torch.int32)))
_def("bce_loss(Tensor logits, Tensor labels) -> (Tensor, Tensor)", lambda logits, labels: _raw.bce_loss(logits, labels),
     lambda logits, labels: (logits.new_empty(()), logits.new_empty(())))

# ---- preprocessing front-end ------------------------------------------------------------------------------------------------
_def("preprocess_u8(Tensor frames, int size, int out_dtype=2) -> Tensor",
     lambda frames, size, out_dtype=2: _raw.preprocess_u8(frames, size, _DT[out_dtype]),
     lambda frames, size, out_dtype=2: frames.new_empty((frames.shape[0], 3, size, size), dtype=_DT[out_dtype]))


# ---- FP8 weight path ---------------------------------------------------------------------------------------------------------
_def("quant_rows_fp8(Tensor x, int d, Tensor(a!) out, Tensor(b!) scale) -> ()",
     lambda x, d, out, scale: (_raw.quant_rows_fp8(x, d, out, scale), None)[1], lambda *a, **k: None)
_def("norm_quant_fp8(Tensor x, Tensor w, Tensor? b, Tensor(a!) out, Tensor(b!) scale, int d, float eps) -> ()",
     lambda x, w, b, out, scale, d, eps: (_raw.norm_quant_fp8(x, w, b, out, scale, d, eps), None)[1], lambda *a, **k: None)
_def("gemm_fp8(Tensor a8, Tensor sa, Tensor w8, Tensor sw, Tensor(a!) out, int mode, int out_dtype, int? n=None, Tensor? bias=None, "
     "int act=0, Tensor? res=None, int res_mod=0, int? m=None) -> ()",
     lambda a8, sa, w8, sw, out, mode, out_dtype, n=None, bias=None, act=0, res=None, res_mod=0, m=None:
         (_raw.gemm_fp8(a8, sa, w8, sw, out, mode=mode, out_dtype=_DT[out_dtype], n=n, bias=bias, act=act, res=res, res_mod=res_mod, m=m),
          None)[1],
     lambda *a, **k: None)

_def("norm_quant_fp8_16(Tensor x, Tensor w, Tensor? b, Tensor(a!) out, Tensor(b!) scale, int d, float eps) -> ()",
     lambda x, w, b, out, scale, d, eps: (_raw.norm_quant_fp8_16(x, w, b, out, scale, d, eps), None)[1], lambda *a, **k: None)
_def("gemm_fp8_res16(Tensor a8, Tensor w8, Tensor sw, Tensor(a!) out, Tensor res16, Tensor? sa=None, Tensor? a_mx=None, Tensor? bias=None, "
     "int? n=None, int? m=None) -> ()",
     lambda a8, w8, sw, out, res16, sa=None, a_mx=None, bias=None, n=None, m=None:
         (_raw.gemm_fp8_res16(a8, w8, sw, out, res16=res16, sa=sa, a_mx=a_mx, bias=bias, n=n, m=m), None)[1],
     lambda *a, **k: None)
_def("gemm_fp8_mx(Tensor a8, Tensor w8, Tensor sw, Tensor(a!) out, int out_dtype, Tensor? sa=None, Tensor? a_mx=None, Tensor(b!)? o_mx=None, "
     "int? n=None, Tensor? bias=None, int act=0, Tensor? res=None, int? m=None, bool swiglu=False) -> ()",
     lambda a8, w8, sw, out, out_dtype, sa=None, a_mx=None, o_mx=None, n=None, bias=None, act=0, res=None, m=None, swiglu=False:
         (_raw.gemm_fp8_mx(a8, w8, sw, out, out_dtype=_DT[out_dtype], sa=sa, a_mx=a_mx, o_mx=o_mx, n=n, bias=bias, act=act, res=res, m=m,
                           swiglu=swiglu), None)[1],
     lambda *a, **k: None)

# ---- call-compatible namespace for the engine (same signatures as videoitg_amd.ops) ------------------------------------
_ops = torch.ops.videoitg
_CODE = {v: k for k, v in _DT.items()}


class _Namespace:
    """`K.gemm(...)` etc. with the keyword signatures of videoitg_amd.ops, dispatched through torch.ops.videoitg."""
    GEMM_STORE16, GEMM_F32, GEMM_SWIGLU = _raw.GEMM_STORE16, _raw.GEMM_F32, _raw.GEMM_SWIGLU
    ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF = _raw.ACT_NONE, _raw.ACT_GELU_TANH, _raw.ACT_GELU_ERF
    pad64 = staticmethod(_raw.pad64)

    @staticmethod
    def patchify(pix, patch, kpad, dtype):
        return _ops.patchify(pix, patch, kpad, _CODE[dtype])

    @staticmethod
    def gemm(a, w, out, *, mode, n=None, bias=None, act=0, res=None, res_mod=0, m=None):
        _ops.gemm(a, w, out, mode, n, bias, act, res, res_mod, m)
        return out

    gemm_splitk_plan = staticmethod(_raw.gemm_splitk_plan)
    gemm_norm_plan = staticmethod(_raw.gemm_norm_plan)

    @staticmethod
    def gemm_norm_producer(a, w, out, *, res, next_gamma, xh, partials, bias=None, res_mod=0):
        _ops.gemm_norm_producer(a, w, out, res, next_gamma, xh, partials, bias, res_mod)
        return out

    @staticmethod
    def norm_stats_finalize(partials, M, P, cols, eps, rms, stats):
        _ops.norm_stats_finalize(partials, M, P, cols, eps, rms, stats)
        return stats

    @staticmethod
    def gemm_norm_consumer(xh, w, out, *, mode, stats, colsum=None, bias=None, act=0, n=None):
        _ops.gemm_norm_consumer(xh, w, out, mode, stats, colsum, bias, act, n)
        return out

    @staticmethod
    def gemm_splitk16(a, w, out, workspace, splits, *, bias=None):
        _ops.gemm_splitk16(a, w, out, workspace, splits, bias)
        return out

    @staticmethod
    def gemm_splitk(a, w, out, workspace, splits, *, bias=None, res=None):
        _ops.gemm_splitk(a, w, out, workspace, splits, bias, res)
        return out

    gemm_streamk_plan = staticmethod(_raw.gemm_streamk_plan)

    @staticmethod
    def gemm_streamk(a, w, out, workspace, blocks, *, bias=None, res=None):
        _ops.gemm_streamk(a, w, out, workspace, blocks, bias, res)
        return out

    @staticmethod
    def layernorm(x, w, b, out, d, eps):
        _ops.layernorm(x, w, b, out, d, eps)
        return out

    @staticmethod
    def rmsnorm(x, w, out, d, eps):
        _ops.rmsnorm(x, w, out, d, eps)
        return out

    @staticmethod
    def rope_(qkv, cos, sin, heads, head_dim):
        _ops.rope_(qkv, cos, sin, heads, head_dim)
        return qkv

    @staticmethod
    def attention(q, k, v, o, *, batch, heads, kv_heads, seq, head_dim, q_strides, k_strides, v_strides, o_strides, scale,
                  variant=0, kv_seq=None, kv_len=None):
        _ops.attention(q, k, v, o, batch, heads, kv_heads, seq, head_dim, list(q_strides), list(k_strides), list(v_strides),
                       list(o_strides), scale, variant, kv_seq, kv_len)
        return o

    @staticmethod
    def bilinear_pool(x, out, T, G, HW, d, cmap=None, pc=0):
        _ops.bilinear_pool(x, out, T, G, HW, d, cmap, pc)
        return out

    @staticmethod
    def bilinear_pool16(x, out, T, G, HW, d, cmap=None, pc=0):
        _ops.bilinear_pool16(x, out, T, G, HW, d, cmap, pc)
        return out

    @staticmethod
    def norm_split(x, w, b, out, d, eps):
        _ops.norm_split(x, w, b, out, d, eps)
        return out

    @staticmethod
    def rope_split_(qkv, cos, sin, heads, head_dim, lo_off):
        _ops.rope_split_(qkv, cos, sin, heads, head_dim, lo_off)
        return qkv

    @staticmethod
    def bilinear_pool_split(x, out, T, G, HW, d, cmap=None, pc=0):
        _ops.bilinear_pool_split(x, out, T, G, HW, d, cmap, pc)
        return out

    @staticmethod
    def gemm_split16(a, w, out, *, mode, lo_off, bias=None, act=0):
        _ops.gemm_split16(a, w, out, mode, lo_off, bias, act)
        return out

    @staticmethod
    def attention_split(q, k, v, o, *, lo, batch, heads, kv_heads, seq, head_dim, q_strides, k_strides, v_strides, o_strides, scale, kv_seq=None):
        _ops.attention_split(q, k, v, o, list(lo), batch, heads, kv_heads, seq, head_dim, list(q_strides), list(k_strides), list(v_strides),
                             list(o_strides), scale, kv_seq)
        return o

    @staticmethod
    def gemm_res16(a, w, out, *, bias=None, res16=None, res32=None, res_mod=0):
        _ops.gemm_res16(a, w, out, bias, res16, res32, res_mod)
        return out

    @staticmethod
    def patch_embed16(pix, wf, out, *, patch, bias=None, pos=None):
        _ops.patch_embed16(pix, wf, out, patch, bias, pos)
        return out

    patch_embed_fusable = staticmethod(_raw.patch_embed_fusable)

    @staticmethod
    def layernorm16(x, w, b, out, d, eps):
        _ops.layernorm16(x, w, b, out, d, eps)
        return out

    @staticmethod
    def gather_rows(src, idx, rows_per_batch, batch, out, width):
        _ops.gather_rows(src, idx, rows_per_batch, batch, out, width)
        return out

    @staticmethod
    def embed_gather(ids, table, out):
        _ops.embed_gather(ids, table, out)
        return out

    @staticmethod
    def pool_score(h, frames, tokens_per_frame, d, w, b, logits):
        _ops.pool_score(h, frames, tokens_per_frame, d, w, b, logits)
        return logits

    @staticmethod
    def rank(logits):
        return _ops.rank(logits)

    @staticmethod
    def bce_loss(logits, labels):
        return _ops.bce_loss(logits, labels)

    pad128 = staticmethod(_raw.pad128)

    @staticmethod
    def quant_rows_fp8(x, d, out, scale):
        _ops.quant_rows_fp8(x, d, out, scale)
        return out, scale

    @staticmethod
    def norm_quant_fp8(x, w, b, out, scale, d, eps):
        _ops.norm_quant_fp8(x, w, b, out, scale, d, eps)
        return out, scale

    @staticmethod
    def gemm_fp8(a8, sa, w8, sw, out, *, mode, out_dtype, n=None, bias=None, act=0, res=None, res_mod=0, m=None):
        _ops.gemm_fp8(a8, sa, w8, sw, out, mode, _CODE[out_dtype], n, bias, act, res, res_mod, m)
        return out

    @staticmethod
    def norm_quant_fp8_16(x, w, b, out, scale, d, eps):
        _ops.norm_quant_fp8_16(x, w, b, out, scale, d, eps)
        return out, scale

    @staticmethod
    def gemm_fp8_res16(a8, w8, sw, out, *, res16, sa=None, a_mx=None, bias=None, n=None, m=None):
        _ops.gemm_fp8_res16(a8, w8, sw, out, res16, sa, a_mx, bias, n, m)
        return out

    @staticmethod
    def gemm_fp8_mx(a8, w8, sw, out, *, out_dtype, sa=None, a_mx=None, o_mx=None, n=None, bias=None, act=0, res=None, m=None, swiglu=False):
        _ops.gemm_fp8_mx(a8, w8, sw, out, _CODE[out_dtype], sa, a_mx, o_mx, n, bias, act, res, m, swiglu)
        return out

    @staticmethod
    def preprocess_u8(frames, size, out_dtype=torch.float32):
        return _ops.preprocess_u8(frames, size, _CODE[out_dtype])


K = _Namespace()
OP_NAMES: List[str] = ["patchify", "gemm", "gemm_splitk", "gemm_splitk16", "layernorm", "rmsnorm", "rope_", "attention", "bilinear_pool", "gather_rows",
                       "embed_gather", "pool_score", "rank", "bce_loss", "preprocess_u8", "quant_rows_fp8", "norm_quant_fp8", "gemm_fp8", "gemm_fp8_mx",
                       "gemm_res16", "layernorm16", "bilinear_pool16", "patch_embed16", "norm_quant_fp8_16", "gemm_fp8_res16",
                       "norm_split", "rope_split_", "bilinear_pool_split", "gemm_split16", "attention_split"] + (["gemm_streamk"] if _raw._lib.PROFILING else [])
