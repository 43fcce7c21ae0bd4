"""Tensor-facing wrappers over the C ABI (include/videoitg_hip.h).  Tensors are torch CUDA(=HIP) tensors used as
typed views of HBM; every call goes to libvideoitg_hip.so on torch's current stream.  No torch math here."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_GELU_ERF, ACT_GELU_TANH, ACT_NONE, BF16, F16, F32, GEMM_F32, GEMM_STORE16, GEMM_SWIGLU  # noqa: F401

_DT = {torch.bfloat16: BF16, torch.float16: F16, torch.float32: F32}


def dt_code(dtype: torch.dtype) -> int:
    return _DT[dtype]


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t: torch.Tensor, dtype=None):
    if not t.is_cuda:
        raise _lib.VitgError("device tensor required (the selector forward has no CPU path)")
    if dtype is not None and t.dtype != dtype:
        raise _lib.VitgError(f"expected {dtype}, got {t.dtype}")
    return t


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def device_info():
    buf = C.create_string_buffer(64)
    cus = _lib.load().vitg_device_info(buf, 64)
    if cus < 0:
        raise _lib.VitgError("no usable HIP device")
    return buf.value.decode(), cus


def set_option(key: str, value: int) -> None:
    """kernel-selection knob (see include/videoitg_hip.h: vitg_set_option)"""
    _lib.call("vitg_set_option", key.encode(), int(value))


def patchify(pix: torch.Tensor, patch: int, kpad: int, dtype: torch.dtype) -> torch.Tensor:
    """pix [T,3,R,R] (fp32/fp16/bf16, contiguous) -> [T*G*G, kpad]"""
    _req(pix)
    pix = pix.contiguous()
    T, Cc, R, _ = pix.shape
    assert Cc == 3
    G = R // patch
    out = torch.empty((T * G * G, kpad), dtype=dtype, device=pix.device)
    _lib.call("vitg_patchify", _ptr(pix), dt_code(pix.dtype), _ptr(out), T, R, patch, kpad, dt_code(dtype), _stream())
    return out


def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, mode: int, n: Optional[int] = None,
         bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, res: Optional[torch.Tensor] = None,
         res_mod: int = 0, m: Optional[int] = None, wf: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = epilogue(a[:m] @ w[:n].T).  a [M,K] and w [N,K] 16-bit 2-D (row stride may exceed K),
    out 2-D (16-bit for STORE16/SWIGLU, fp32 for F32).  wf: pack_weight_fragments(w[:n]) or None."""
    _req(a)
    _req(w, a.dtype)
    M = a.shape[0] if m is None else m
    N = w.shape[0] if n is None else n
    K = w.shape[1]
    assert a.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1
    if mode == GEMM_F32:
        _req(out, torch.float32)
    else:
        _req(out, a.dtype)
    if bias is not None:
        _req(bias, torch.float32)
    if res is not None:
        _req(res, torch.float32)
    if wf is not None:                                    # profiling library only (include/videoitg_hip_profiling.h)
        _lib.call("vitg_gemm_wf", _ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(wf), M, N, K, dt_code(a.dtype), mode, _ptr(bias),
                  act, _ptr(out), out.stride(0), _ptr(res), 0 if res is None else res.stride(0), res_mod, _stream())
        return out
    _lib.call("vitg_gemm", _ptr(a), a.stride(0), _ptr(w), w.stride(0), M, N, K, dt_code(a.dtype), mode, _ptr(bias), act,
              _ptr(out), out.stride(0), _ptr(res), 0 if res is None else res.stride(0), res_mod, _stream())
    return out


def gemm_res16(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
               res16: Optional[torch.Tensor] = None, res32: Optional[torch.Tensor] = None, res_mod: int = 0) -> torch.Tensor:
    """out16 = round16(a @ w.T + bias + res16 + res32[m % res_mod]) - the tower's 16-bit residual stream (vitg_gemm_res16)"""
    _req(a); _req(w); _req(out, a.dtype)
    M, Kd = a.shape
    N = w.shape[0]
    if res16 is not None:
        _req(res16, a.dtype)
    if res32 is not None:
        _req(res32, torch.float32)
    _lib.call("vitg_gemm_res16", _ptr(a), a.stride(0), _ptr(w), w.stride(0), M, N, Kd, dt_code(a.dtype), _ptr(bias), _ptr(out), out.stride(0),
              _ptr(res16), res16.stride(0) if res16 is not None else 0, _ptr(res32), res32.stride(0) if res32 is not None else 0, res_mod, _stream())
    return out


def patch_embed_fusable(T: int, R: int, patch: int, n: int) -> bool:
    """the geometries vitg_patch_embed16 covers (same tests as the C entry): otherwise vitg_patchify + vitg_gemm_res16"""
    return (8 <= patch <= 16 and patch % 2 == 0 and R >= patch and R % 2 == 0 and n >= 256 and n % 4 == 0
            and T * 3 * R * R * 2 < (1 << 32) and T * (R // patch) ** 2 < (1 << 31))


def patch_embed16(pix: torch.Tensor, wf: torch.Tensor, out: torch.Tensor, *, patch: int, bias: Optional[torch.Tensor] = None,
                  pos: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out16[T*P, N] = im2col(pix) @ W^T + bias + pos[m % P]: SigLIP's patch embedding with the im2col inside the GEMM's operand loader
    (vitg_patch_embed16); wf = the convolution weight in the (c, ky)-pair layout (PackedWeights.patch_wf)"""
    _req(pix); _req(wf, pix.dtype); _req(out, pix.dtype)
    T, C3, R, R2 = pix.shape
    if C3 != 3 or R != R2 or not pix.is_contiguous():
        raise ValueError("pixel_values must be a contiguous [T, 3, R, R] tensor")
    if pos is not None:
        _req(pos, torch.float32)
    if bias is not None:
        _req(bias, torch.float32)
    G = R // patch
    if out.dim() != 2 or out.shape[0] != T * G * G or out.shape[1] < wf.shape[0] or out.stride(1) != 1:
        raise ValueError(f"out must be [T * (R // patch)^2 = {T * G * G}, >= {wf.shape[0]}] with unit column stride, got {tuple(out.shape)}")
    if bias is not None and bias.numel() < wf.shape[0]:
        raise ValueError("bias shorter than the embedding width")
    _lib.call("vitg_patch_embed16", _ptr(pix), T, R, patch, _ptr(wf), wf.stride(0), wf.shape[0], dt_code(pix.dtype), _ptr(bias), _ptr(out),
              out.stride(0), _ptr(pos), pos.stride(0) if pos is not None else 0, _stream())
    return out


def gemm_norm_plan(M: int, N: int, K: int) -> int:
    """output columns per partial of the fused-norm producer for this shape; 0 = not covered (use the stand-alone norm)"""
    c = C.c_int(0)
    _lib.call("vitg_gemm_norm_plan", M, N, K, C.byref(c))
    return c.value


def gemm_norm_producer(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, res: torch.Tensor, next_gamma: torch.Tensor,
                       xh: torch.Tensor, partials: torch.Tensor, bias: Optional[torch.Tensor] = None, res_mod: int = 0) -> torch.Tensor:
    """out(fp32) = a @ w.T + bias + res;  xh = round16(out * next_gamma);  partials[m, p] = (mean, M2) of out[m, p*cols:(p+1)*cols]"""
    _req(a); _req(w, a.dtype); _req(out, torch.float32); _req(res, torch.float32); _req(next_gamma, torch.float32)
    _req(xh, a.dtype); _req(partials, torch.float32)
    M, K, N = a.shape[0], w.shape[1], w.shape[0]
    assert a.shape[1] == K and xh.shape[0] >= M and xh.stride(1) == 1
    _lib.call("vitg_gemm_norm_producer", _ptr(a), a.stride(0), _ptr(w), w.stride(0), M, N, K, dt_code(a.dtype), _ptr(bias), _ptr(out),
              out.stride(0), _ptr(res), res.stride(0), res_mod, _ptr(next_gamma), _ptr(xh), xh.stride(0), _ptr(partials), _stream())
    return out


def norm_stats_finalize(partials: torch.Tensor, M: int, P: int, cols: int, eps: float, rms: bool, stats: torch.Tensor) -> torch.Tensor:
    _req(partials, torch.float32); _req(stats, torch.float32)
    _lib.call("vitg_norm_stats_finalize", _ptr(partials), M, P, cols, eps, 1 if rms else 0, _ptr(stats), _stream())
    return stats


def gemm_norm_consumer(xh: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, mode: int, stats: torch.Tensor,
                       colsum: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
                       n: Optional[int] = None) -> torch.Tensor:
    """out = epilogue(rstd[m] * (xh @ w.T - mean[m] * colsum) + bias)  (16-bit output modes)"""
    _req(xh); _req(w, xh.dtype); _req(out, xh.dtype); _req(stats, torch.float32)
    M, K = xh.shape[0], w.shape[1]
    N = w.shape[0] if n is None else n
    assert xh.shape[1] == K
    _lib.call("vitg_gemm_norm_consumer", _ptr(xh), xh.stride(0), _ptr(w), w.stride(0), M, N, K, dt_code(xh.dtype), mode, _ptr(bias), act,
              _ptr(out), out.stride(0), _ptr(stats), _ptr(colsum), _stream())
    return out


def gemm_splitk_plan(M: int, N: int, K: int):
    """-> (splits, workspace_bytes): splits == 1 means the plain GEMM is the right call (vitg_gemm_splitk_plan)"""
    sp, nb = C.c_int(0), C.c_long(0)
    _lib.call("vitg_gemm_splitk_plan", M, N, K, C.byref(sp), C.byref(nb))
    return sp.value, nb.value


def gemm_splitk(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, workspace: torch.Tensor, splits: int, *,
                bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out(fp32) = a @ w.T + bias + res through `splits` K-partials in `workspace` (fp32, >= splits*M*N), fixed summation order"""
    _req(a)
    _req(w, a.dtype)
    _req(out, torch.float32)
    _req(workspace, torch.float32)
    M, K, N = a.shape[0], w.shape[1], w.shape[0]
    assert a.shape[1] == K and workspace.numel() >= splits * M * N
    _lib.call("vitg_gemm_splitk", _ptr(a), a.stride(0), _ptr(w), w.stride(0), M, N, K, dt_code(a.dtype), _ptr(bias), _ptr(out),
              out.stride(0), _ptr(res), 0 if res is None else res.stride(0), _ptr(workspace), splits, _stream())
    return out


def gemm_splitk16(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, workspace: torch.Tensor, splits: int, *,
                  bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out16 = round16(a @ w.T + bias) through `splits` fp32 K-partials in `workspace` (>= splits*M*N floats), fixed summation order (vitg_gemm_splitk16)"""
    _req(a)
    _req(w, a.dtype)
    _req(out, a.dtype)
    _req(workspace, torch.float32)
    M, K, N = a.shape[0], w.shape[1], w.shape[0]
    assert a.shape[1] == K and out.shape[0] >= M and out.shape[1] >= N and out.stride(1) == 1 and workspace.numel() >= splits * M * N
    if bias is not None:
        _req(bias, torch.float32)
    _lib.call("vitg_gemm_splitk16", _ptr(a), a.stride(0), _ptr(w), w.stride(0), M, N, K, dt_code(a.dtype), _ptr(bias), _ptr(out),
              out.stride(0), _ptr(workspace), splits, _stream())
    return out


def gemm_streamk_plan(M: int, N: int, K: int):
    """-> (blocks, workspace_bytes): blocks == 0 means stream-K does not apply to the shape (vitg_gemm_streamk_plan)"""
    nb_, ws_ = C.c_int(0), C.c_long(0)
    _lib.call("vitg_gemm_streamk_plan", M, N, K, C.byref(nb_), C.byref(ws_))
    return nb_.value, ws_.value


def gemm_streamk(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, workspace: torch.Tensor, blocks: int, *,
                 bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out(fp32) = a @ w.T + bias + res through a stream-K partition into `blocks` equal K-runs (vitg_gemm_streamk), fixed summation order"""
    _req(a)
    _req(w, a.dtype)
    _req(out, torch.float32)
    _req(workspace, torch.float32)
    M, K, N = a.shape[0], w.shape[1], w.shape[0]
    assert a.shape[1] == K and workspace.numel() >= blocks * 2 * 65536
    _lib.call("vitg_gemm_streamk", _ptr(a), a.stride(0), _ptr(w), w.stride(0), M, N, K, dt_code(a.dtype), _ptr(bias), _ptr(out),
              out.stride(0), _ptr(res), 0 if res is None else res.stride(0), _ptr(workspace), blocks, _stream())
    return out


def pad128(n: int) -> int:
    return (n + 127) // 128 * 128


def quant_rows_fp8(x: torch.Tensor, d: Optional[int] = None, out: Optional[torch.Tensor] = None,
                   scale: Optional[torch.Tensor] = None):
    """x [M, ld] (fp32 / fp16 / bf16; first d columns are payload) -> (uint8 [M, pad128(d)] e4m3fn bytes, fp32 [M] scales)"""
    _req(x)
    M = x.shape[0]
    d = x.shape[1] if d is None else d
    if out is None:
        out = torch.empty((M, pad128(d)), dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty(M, dtype=torch.float32, device=x.device)
    _req(out, torch.uint8)
    _req(scale, torch.float32)
    assert x.stride(1) == 1 and out.stride(1) == 1 and out.shape[0] >= M and scale.numel() >= M
    _lib.call("vitg_quant_rows_fp8", _ptr(x), x.stride(0), dt_code(x.dtype), _ptr(out), out.stride(0), _ptr(scale), M, d, _stream())
    return out, scale


def norm_quant_fp8(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], out: torch.Tensor, scale: torch.Tensor, d: int,
                   eps: float):
    """LayerNorm (b given) / RMSNorm (b None) of fp32 rows fused with the per-token e4m3fn quantiser"""
    _req(x, torch.float32)
    _req(out, torch.uint8)
    _req(scale, torch.float32)
    _lib.call("vitg_norm_quant_fp8", _ptr(x), x.stride(0), _ptr(_req(w, torch.float32)), _ptr(None if b is None else _req(b, torch.float32)),
              _ptr(out), out.stride(0), _ptr(scale), x.shape[0], d, eps, _stream())
    return out, scale


def norm_quant_fp8_16(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], out: torch.Tensor, scale: torch.Tensor, d: int,
                      eps: float):
    """norm_quant_fp8 on 16-bit rows (the tower's 16-bit residual stream; vitg_norm_quant_fp8_16)"""
    _req(x)
    _req(out, torch.uint8)
    _req(scale, torch.float32)
    _lib.call("vitg_norm_quant_fp8_16", _ptr(x), x.stride(0), dt_code(x.dtype), _ptr(_req(w, torch.float32)),
              _ptr(None if b is None else _req(b, torch.float32)), _ptr(out), out.stride(0), _ptr(scale), x.shape[0], d, eps, _stream())
    return out, scale


def gemm_fp8_res16(a8: torch.Tensor, w8: torch.Tensor, sw: torch.Tensor, out: torch.Tensor, *, res16: torch.Tensor,
                   sa: Optional[torch.Tensor] = None, a_mx: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
                   n: Optional[int] = None, m: Optional[int] = None) -> torch.Tensor:
    """out16 = round16(scales * (a8 @ w8.T) + bias + res16): the 16-bit residual update on the fp8 weight path (vitg_gemm_fp8_res16);
    a8 with per-token scales `sa` or MX block scales `a_mx` (uint8 [K/128, rows >= M, 4])"""
    _req(a8, torch.uint8); _req(w8, torch.uint8); _req(sw, torch.float32); _req(out); _req(res16, out.dtype)
    M = a8.shape[0] if m is None else m
    N = w8.shape[0] if n is None else n
    K = w8.shape[1]
    assert a8.shape[1] == K and a8.stride(1) == 1 and w8.stride(1) == 1 and out.stride(1) == 1 and res16.stride(1) == 1
    assert (sa is None) != (a_mx is None)
    if sa is not None:
        _req(sa, torch.float32)
    else:
        _req(a_mx, torch.uint8)
        assert a_mx.dim() == 3 and a_mx.shape[0] == K // 128 and a_mx.shape[1] >= M and a_mx.shape[2] == 4 and a_mx.is_contiguous()
    if bias is not None:
        _req(bias, torch.float32)
    _lib.call("vitg_gemm_fp8_res16", _ptr(a8), a8.stride(0), _ptr(sa), _ptr(a_mx), 0 if a_mx is None else a_mx.shape[1], _ptr(w8), w8.stride(0),
              _ptr(sw), M, N, K, dt_code(out.dtype), _ptr(bias), _ptr(out), out.stride(0), _ptr(res16), res16.stride(0), _stream())
    return out


def gemm_fp8(a8: torch.Tensor, sa: torch.Tensor, w8: torch.Tensor, sw: torch.Tensor, out: torch.Tensor, *, mode: int,
             out_dtype: torch.dtype, n: Optional[int] = None, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
             res: Optional[torch.Tensor] = None, res_mod: int = 0, m: Optional[int] = None) -> torch.Tensor:
    """out = epilogue(diag(sa) (a8[:m] @ w8[:n].T) diag(sw)); a8 [M,K], w8 [N,K] uint8 e4m3fn bytes, K % 128 == 0"""
    _req(a8, torch.uint8)
    _req(w8, torch.uint8)
    M = a8.shape[0] if m is None else m
    N = w8.shape[0] if n is None else n
    K = w8.shape[1]
    assert a8.shape[1] == K and a8.stride(1) == 1 and w8.stride(1) == 1 and out.stride(1) == 1
    _req(out, torch.float32 if mode == GEMM_F32 else out_dtype)
    _req(sa, torch.float32)
    _req(sw, torch.float32)
    if bias is not None:
        _req(bias, torch.float32)
    if res is not None:
        _req(res, torch.float32)
    _lib.call("vitg_gemm_fp8", _ptr(a8), a8.stride(0), _ptr(sa), _ptr(w8), w8.stride(0), _ptr(sw), M, N, K, dt_code(out_dtype), mode,
              _ptr(bias), act, _ptr(out), out.stride(0), _ptr(res), 0 if res is None else res.stride(0), res_mod, _stream())
    return out


def gemm_fp8_mx(a8: torch.Tensor, w8: torch.Tensor, sw: torch.Tensor, out: torch.Tensor, *, out_dtype: torch.dtype,
                sa: Optional[torch.Tensor] = None, a_mx: Optional[torch.Tensor] = None, o_mx: Optional[torch.Tensor] = None,
                n: Optional[int] = None, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, res: Optional[torch.Tensor] = None,
                m: Optional[int] = None, swiglu: bool = False) -> torch.Tensor:
    """MX forms of gemm_fp8 (vitg_gemm_fp8_mx).  swiglu (producer only): w8 rows are gate / up interleaved as for GEMM_SWIGLU, N/2 outputs.  Block scales live in a uint8 [K/128, rows >= M, 4] tensor (one dword per row and
    128-deep K-tile).  Producer: o_mx given, out uint8 [M, ldo] (ldo % 128 == 0) receives e4m3 bytes, sa = per-token scales of a8.
    Consumer: a_mx given (block scales of a8), out fp32 [M, N] (+ bias + res)."""
    _req(a8, torch.uint8)
    _req(w8, torch.uint8)
    _req(sw, torch.float32)
    M = a8.shape[0] if m is None else m
    N = w8.shape[0] if n is None else n
    K = w8.shape[1]
    assert a8.shape[1] == K and a8.stride(1) == 1 and w8.stride(1) == 1 and out.stride(1) == 1
    assert (a_mx is None) != (o_mx is None)
    if o_mx is not None:
        _req(out, torch.uint8)
        _req(o_mx, torch.uint8)
        _req(sa, torch.float32)
        mode = GEMM_SWIGLU if swiglu else GEMM_STORE16
    else:
        assert not swiglu
        _req(out, torch.float32)
        _req(a_mx, torch.uint8)
        mode = GEMM_F32
    if bias is not None:
        _req(bias, torch.float32)
    if res is not None:
        _req(res, torch.float32)
    for t in (a_mx, o_mx):
        assert t is None or (t.dim() == 3 and t.shape[2] == 4 and t.stride(2) == 1 and t.stride(1) == 4 and t.stride(0) == 4 * t.shape[1])
    _lib.call("vitg_gemm_fp8_mx", _ptr(a8), a8.stride(0), _ptr(sa), _ptr(a_mx), 0 if a_mx is None else a_mx.shape[1], _ptr(w8), w8.stride(0),
              _ptr(sw), M, N, K, dt_code(out_dtype), mode, _ptr(bias), act, _ptr(out), out.stride(0), _ptr(o_mx),
              0 if o_mx is None else o_mx.shape[1], _ptr(res), 0 if res is None else res.stride(0), _stream())
    return out


def pack_weight_fragments(w: torch.Tensor, n: Optional[int] = None) -> torch.Tensor:
    """MFMA-fragment-ordered copy of w[:n] [N,K] for gemm(..., wf=...) (vitg_pack_weight_fragments; profiling library only)"""
    import ctypes
    _req(w)
    N = w.shape[0] if n is None else n
    K = w.shape[1]
    nb = ctypes.c_long(0)
    _lib.call("vitg_weight_fragments_bytes", N, K, ctypes.byref(nb))
    out = torch.empty(nb.value // 2, dtype=w.dtype, device=w.device)
    _lib.call("vitg_pack_weight_fragments", _ptr(w), w.stride(0), N, K, _ptr(out), _stream())
    return out


def mfma_probe_fp8_tflops(random_operands: bool, ms_target: float = 60.0) -> float:
    """the same probe on the fp8 weight path's MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3 bytes)"""
    import ctypes
    dev = torch.device("cuda", torch.cuda.current_device())
    pool = torch.zeros(8 * 64 * 16, dtype=torch.uint8, device=dev)
    if random_operands:                                   # finite e4m3 values of ordinary magnitude (no NaN encodings: exponent field < 15)
        pool = (torch.randint(0, 120, pool.shape, device=dev) | (torch.randint(0, 2, pool.shape, device=dev) << 7)).to(torch.uint8)
    blocks = torch.cuda.get_device_properties(dev).multi_processor_count
    out = torch.empty(blocks * 512, dtype=torch.float32, device=dev)
    fl = ctypes.c_double(0.0)
    iters, tf = 1000, 0.0
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("vitg_mfma_probe", _ptr(pool), 3, blocks, iters, _ptr(out), ctypes.byref(fl), _stream())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tf = fl.value / ms / 1e9
        iters = max(500, int(iters * ms_target / max(ms, 1e-3)))
    return tf


def mfma_probe_tflops(dtype: torch.dtype, random_operands: bool, ms_target: float = 60.0) -> float:
    """MFMA-only rate the device sustains (profiling aid, vitg_mfma_probe): zeros vs random operands differ because the shader
    clock under MFMA load is power managed.  Runs for ~3 x ms_target; returns the last measurement in TFLOP/s."""
    import ctypes
    dev = torch.device("cuda", torch.cuda.current_device())
    pool = torch.zeros(8 * 64 * 8, dtype=dtype, device=dev)
    if random_operands:
        pool = (torch.rand(pool.shape, device=dev) * 2 - 1).to(dtype)
    blocks = torch.cuda.get_device_properties(dev).multi_processor_count
    out = torch.empty(blocks * 512, dtype=torch.float32, device=dev)
    fl = ctypes.c_double(0.0)
    iters, tf = 2000, 0.0
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("vitg_mfma_probe", _ptr(pool), dt_code(dtype), blocks, iters, _ptr(out), ctypes.byref(fl), _stream())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tf = fl.value / ms / 1e9
        iters = max(1000, int(iters * ms_target / max(ms, 1e-3)))
    return tf


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out: torch.Tensor, d: int, eps: float) -> torch.Tensor:
    _req(x, torch.float32)
    _lib.call("vitg_layernorm", _ptr(x), x.stride(0), _ptr(_req(w, torch.float32)), _ptr(_req(b, torch.float32)),
              _ptr(out), out.stride(0), dt_code(out.dtype), x.shape[0], d, eps, _stream())
    return out


def layernorm16(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out: torch.Tensor, d: int, eps: float) -> torch.Tensor:
    """LayerNorm of 16-bit rows into 16-bit rows of the same type (vitg_layernorm16)"""
    _req(x); _req(out, x.dtype)
    _lib.call("vitg_layernorm16", _ptr(x), x.stride(0), _ptr(_req(w, torch.float32)), _ptr(_req(b, torch.float32)), _ptr(out), out.stride(0),
              dt_code(x.dtype), x.shape[0], d, float(eps), _stream())
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, d: int, eps: float) -> torch.Tensor:
    _req(x, torch.float32)
    _lib.call("vitg_rmsnorm", _ptr(x), x.stride(0), _ptr(_req(w, torch.float32)), _ptr(out), out.stride(0),
              dt_code(out.dtype), x.shape[0], d, eps, _stream())
    return out


def rope_(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, heads: int, head_dim: int) -> torch.Tensor:
    """rotate the first `heads` heads of every row of qkv [N, ld] in place"""
    _req(qkv)
    _lib.call("vitg_rope", _ptr(qkv), qkv.stride(0), _ptr(_req(cos, torch.float32)), _ptr(_req(sin, torch.float32)),
              qkv.shape[0], heads, head_dim, dt_code(qkv.dtype), _stream())
    return qkv


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, *, batch: int, heads: int, kv_heads: int,
              seq: int, head_dim: int, q_strides, k_strides, v_strides, o_strides, scale: float, variant: int = 0,
              kv_seq: Optional[int] = None, kv_len: Optional[torch.Tensor] = None):
    """q/k/v/o are base tensors (views at the first element); *_strides = (batch, row, head) in elements.
    kv_len: optional int32 device scalar - the number of keys is read from it at kernel start (<= kv_seq; vitg_attention_dyn)."""
    _req(q)
    if kv_len is not None:
        _req(kv_len, torch.int32)
        _lib.call("vitg_attention_dyn", _ptr(q), _ptr(k), _ptr(v), _ptr(o), batch, heads, kv_heads, seq,
                  seq if kv_seq is None else kv_seq, _ptr(kv_len), head_dim,
                  q_strides[0], k_strides[0], v_strides[0], o_strides[0], q_strides[1], k_strides[1], v_strides[1], o_strides[1],
                  q_strides[2], k_strides[2], v_strides[2], o_strides[2], scale, dt_code(q.dtype), variant, _stream())
        return o
    _lib.call("vitg_attention", _ptr(q), _ptr(k), _ptr(v), _ptr(o), batch, heads, kv_heads, seq,
              seq if kv_seq is None else kv_seq, head_dim,
              q_strides[0], k_strides[0], v_strides[0], o_strides[0], q_strides[1], k_strides[1], v_strides[1], o_strides[1],
              q_strides[2], k_strides[2], v_strides[2], o_strides[2], scale, dt_code(q.dtype), variant, _stream())
    return o


def bilinear_pool(x: torch.Tensor, out: torch.Tensor, T: int, G: int, HW: int, d: int, cmap: Optional[torch.Tensor] = None,
                  pc: int = 0) -> torch.Tensor:
    _req(x, torch.float32)
    if cmap is not None:
        _req(cmap, torch.int32)
    _lib.call("vitg_bilinear_pool", _ptr(x), x.stride(0), _ptr(out), out.stride(0), T, G, HW, d, dt_code(out.dtype), _ptr(cmap), pc,
              _stream())
    return out


def bilinear_pool16(x: torch.Tensor, out: torch.Tensor, T: int, G: int, HW: int, d: int, cmap: Optional[torch.Tensor] = None,
                    pc: int = 0) -> torch.Tensor:
    """bilinear_pool on 16-bit features (vitg_bilinear_pool16)"""
    _req(x); _req(out, x.dtype)
    _lib.call("vitg_bilinear_pool16", _ptr(x), x.stride(0), _ptr(out), out.stride(0), T, G, HW, d, dt_code(x.dtype), _ptr(cmap), pc, _stream())
    return out


def gather_rows(src: torch.Tensor, idx: torch.Tensor, rows_per_batch: int, batch: int, out: torch.Tensor, width: int) -> torch.Tensor:
    """out[b*len(idx)+i, :width] = src[b*rows_per_batch + idx[i], :width]"""
    _req(src)
    _req(idx, torch.int32)
    es = src.element_size()
    _lib.call("vitg_gather_rows", _ptr(src), src.stride(0) * es, _ptr(idx), idx.numel(), rows_per_batch, batch, _ptr(out),
              out.stride(0) * es, width * es, _stream())
    return out


def embed_gather(ids: torch.Tensor, table: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    _req(ids, torch.int64)
    _req(out, torch.float32)
    _lib.call("vitg_embed_gather", _ptr(ids), ids.numel(), _ptr(table), table.shape[1], _ptr(out), out.stride(0),
              dt_code(table.dtype), _stream())
    return out


def pool_score(h: torch.Tensor, frames: int, tokens_per_frame: int, d: int, w: torch.Tensor, b: torch.Tensor,
               logits: torch.Tensor) -> torch.Tensor:
    _req(h, torch.float32)
    _lib.call("vitg_pool_score", _ptr(h), h.stride(0), frames, tokens_per_frame, d, _ptr(_req(w, torch.float32)),
              _ptr(_req(b, torch.float32)), _ptr(_req(logits, torch.float32)), _stream())
    return logits


def rank(logits: torch.Tensor):
    _req(logits, torch.float32)
    n = logits.numel()
    scores = torch.empty(n, dtype=torch.float32, device=logits.device)
    order = torch.empty(n, dtype=torch.int32, device=logits.device)
    _lib.call("vitg_rank", _ptr(logits), n, _ptr(scores), _ptr(order), _stream())
    return scores, order


def preprocess_u8(frames: torch.Tensor, size: int, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """uint8 [T,H,W,3] device frames -> normalised [T,3,size,size] (Pillow-exact bicubic resize + SigLIP rescale/normalise)"""
    _req(frames, torch.uint8)
    if frames.dim() != 4 or frames.shape[-1] != 3:
        raise _lib.VitgError(f"expected uint8 [T,H,W,3], got {tuple(frames.shape)}")
    T, H, W, _ = frames.shape
    if T > 0 and H > 100 * W and size < H and 1 < W != size:
        # Pillow runs the vertical pass first when an image more than 100 times taller than wide is reduced in height; the kernel is horizontal-first: the
        # same arithmetic on the transposed frames, transposed back (include/videoitg_hip.h: the C entry refuses this geometry)
        return preprocess_u8(frames.transpose(1, 2).contiguous(), size, out_dtype).transpose(2, 3).contiguous()
    if T > 0 and (frames.stride(3) != 1 or frames.stride(2) != 3 or (H == 1 and frames.stride(1) < 3 * W) or (T == 1 and frames.stride(0) < frames.stride(1) * H)):
        frames = frames.contiguous()
    out = torch.empty((T, 3, size, size), dtype=out_dtype, device=frames.device)
    if T == 0:
        return out
    need = C.c_long(0)
    _lib.call("vitg_preprocess_tmp_bytes", T, H, W, size, C.byref(need))
    tmp = torch.empty(need.value, dtype=torch.uint8, device=frames.device) if need.value else None
    row_stride = frames.stride(1) if H > 1 else max(frames.stride(1), 3 * W)          # (the stride of a size-1 dimension is arbitrary in torch)
    frame_stride = frames.stride(0) if T > 1 else max(frames.stride(0), row_stride * H)
    _lib.call("vitg_preprocess_u8", _ptr(frames), T, H, W, frame_stride, row_stride, _ptr(out), size, dt_code(out_dtype),
              _ptr(tmp), need.value, _stream())
    return out


def bce_loss(logits: torch.Tensor, labels: torch.Tensor):
    """-> (loss scalar tensor fp32, pos_weight float tensor): the reference's grounding loss (grounding_qwen2.py:159-169)"""
    _req(logits, torch.float32)
    _req(labels, torch.float32)
    if logits.numel() != labels.numel() or logits.numel() == 0:
        raise _lib.VitgError("logits and labels must have the same, non-zero, number of elements")
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    _lib.call("vitg_bce_loss", _ptr(logits.contiguous()), _ptr(labels.contiguous()), logits.numel(), _ptr(out), _stream())
    return out[0], out[1]


# ---- operand-split precision tier (csrc/split.hip): split operands are fp16 [M, 2 * Dp] = [hi | lo] ------------------------------------------
def norm_split(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], out: torch.Tensor, d: int, eps: float) -> torch.Tensor:
    """LayerNorm (b given) / RMSNorm of fp32 rows -> out fp16 [M, 2 * Dp]: hi in [:, :Dp], lo in [:, Dp:] (vitg_norm_split)"""
    _req(x, torch.float32); _req(out, torch.float16)
    lo = out.shape[1] // 2
    _lib.call("vitg_norm_split", _ptr(x), x.stride(0), _ptr(_req(w, torch.float32)), _ptr(b), _ptr(out), out.stride(0), lo, x.shape[0], d, float(eps),
              _stream())
    return out


def rope_split_(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, heads: int, head_dim: int, lo_off: int) -> torch.Tensor:
    _req(qkv, torch.float16)
    _lib.call("vitg_rope_split", _ptr(qkv), qkv.stride(0), lo_off, _ptr(_req(cos, torch.float32)), _ptr(_req(sin, torch.float32)), qkv.shape[0], heads,
              head_dim, _stream())
    return qkv


def bilinear_pool_split(x: torch.Tensor, out: torch.Tensor, T: int, G: int, HW: int, d: int, cmap: Optional[torch.Tensor] = None, pc: int = 0) -> torch.Tensor:
    _req(x, torch.float32); _req(out, torch.float16)
    if cmap is not None:
        _req(cmap, torch.int32)
    _lib.call("vitg_bilinear_pool_split", _ptr(x), x.stride(0), _ptr(out), out.stride(0), out.shape[1] // 2, T, G, HW, d, _ptr(cmap), pc, _stream())
    return out


def gemm_split16(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, mode: int, lo_off: int, bias: Optional[torch.Tensor] = None,
                 act: int = ACT_NONE) -> torch.Tensor:
    """out[:, n] / out[:, lo_off + n] = (hi, lo) of epilogue(a @ w.T) - vitg_gemm's 16-bit-output modes writing split pairs (vitg_gemm_split16)"""
    _req(a, torch.float16); _req(w, torch.float16); _req(out, torch.float16)
    M, Kd = a.shape
    assert w.shape[1] == Kd and a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1
    if bias is not None:
        _req(bias, torch.float32)
    _lib.call("vitg_gemm_split16", _ptr(a), a.stride(0), _ptr(w), w.stride(0), M, w.shape[0], Kd, mode, _ptr(bias), act, _ptr(out), out.stride(0),
              lo_off, _stream())
    return out


def attention_split(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, *, lo, batch: int, heads: int, kv_heads: int, seq: int,
                    head_dim: int, q_strides, k_strides, v_strides, o_strides, scale: float, kv_seq: Optional[int] = None) -> torch.Tensor:
    """q/k/v/o: base tensors of the hi planes; lo = (q_lo, k_lo, v_lo, o_lo) element offsets of the lo planes; *_strides = (batch, row, head)"""
    _req(q, torch.float16); _req(o, torch.float16)
    _lib.call("vitg_attention_split", _ptr(q), _ptr(k), _ptr(v), _ptr(o), lo[0], lo[1], lo[2], lo[3], batch, heads, kv_heads, seq,
              seq if kv_seq is None else kv_seq, head_dim, q_strides[0], k_strides[0], v_strides[0], o_strides[0], q_strides[1], k_strides[1],
              v_strides[1], o_strides[1], q_strides[2], k_strides[2], v_strides[2], o_strides[2], float(scale), _stream())
    return o
