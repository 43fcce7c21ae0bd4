#!/usr/bin/env python3
"""GPU: time-bounded differential fuzzing of the C-ABI kernels against torch fp32 on the values the kernel saw.  The test suite pins chosen shapes; this
draws shapes nobody chose - ragged M, N that is only a multiple of 4, strided operands, one-key attention, key counts that straddle tile edges, ties in
the ranking - and reports every case outside the same error bounds the tests use.  Not part of the suite (its run time is the argument).

    python tools/fuzz_kernels.py [seconds per family = 40] [seed = 0] [families]      -> one line per family, failures listed with their shapes; exit 1 on any
"""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
EPS = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only = set(sys.argv[3].split(",")) if len(sys.argv) > 3 else None           # family names, comma-separated
gen = torch.Generator().manual_seed(seed)


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=gen).item())


def pick(*xs):
    return xs[ri(0, len(xs) - 1)]


def rnd(shape, scale=1.0, dtype=torch.float32):
    return (torch.randn(shape, generator=gen) * scale).to(dtype)


def strided(t, extra):
    """the same values behind a row stride that exceeds the width (a view into a wider buffer)"""
    if not extra:
        return t.to(DEV).contiguous()
    buf = torch.full((t.shape[0], t.shape[1] + extra), float("nan"), dtype=t.dtype, device=DEV)
    buf[:, :t.shape[1]] = t.to(DEV)
    return buf[:, :t.shape[1]]


def rows_m():
    return pick(ri(1, 40), ri(1, 700), ri(200, 5000), ri(200, 5000), ri(5000, 70000))


def fuzz_gemm():
    dt = pick(torch.float16, torch.bfloat16)
    M, N, K = rows_m(), 4 * ri(1, 1100), 64 * ri(1, pick(8, 8, 40, 80))
    if M * N > 6e7 or M * K > 6e7:
        M = max(1, int(6e7 // max(N, K)))
    form = pick("store16", "store16", "f32", "f32_mod", "swiglu", "res16")
    a, w = rnd((M, K), 1.0, dt), rnd((N, K), K ** -0.5, dt)
    b = rnd((N,), 0.1) if ri(0, 3) else None
    af, wf = a.float().to(DEV), w.float().to(DEV)                          # reference on the device through torch's own fp32 GEMM (analysis only): more cases per second
    mm = lambda x, y: (x @ y.t()).cpu()                                    # noqa: E731
    ad, wd = strided(a, pick(0, 0, 8, 64)), strided(w, pick(0, 0, 8))
    lo = pick(0, 0, 4, 36)
    what = f"gemm {form} {dt} M={M} N={N} K={K} lda={ad.stride(0)} ldw={wd.stride(0)} ldo+={lo} bias={b is not None}"
    ref = mm(af, wf) + (b if b is not None else 0)
    if form == "store16":
        act = pick(ops.ACT_NONE, ops.ACT_GELU_TANH, ops.ACT_GELU_ERF)
        ref = {ops.ACT_NONE: lambda x: x, ops.ACT_GELU_TANH: lambda x: F.gelu(x, approximate="tanh"), ops.ACT_GELU_ERF: F.gelu}[act](ref)
        out = strided(torch.full((M, N), float("nan"), dtype=dt), lo)
        ops.gemm(ad, wd, out, mode=ops.GEMM_STORE16, bias=None if b is None else b.to(DEV), act=act)
        tol = EPS[dt] * ref.abs().max().item() + 1e-3
        return what + f" act={act}", (out.float().cpu() - ref).abs().max().item(), tol
    if form in ("f32", "f32_mod"):
        if form == "f32_mod":
            P = ri(1, max(1, min(M, 900)))
            res = rnd((P, N))
            ref = ref + res.repeat(M // P + 1, 1)[:M]
            out = strided(torch.full((M, N), float("nan")), lo)
            ops.gemm(ad, wd, out, mode=ops.GEMM_F32, bias=None if b is None else b.to(DEV), res=res.to(DEV), res_mod=P)
        else:
            x = rnd((M, N))
            ref = ref + x
            out = strided(x, lo)
            ops.gemm(ad, wd, out, mode=ops.GEMM_F32, bias=None if b is None else b.to(DEV), res=out)
        return what, (out.cpu() - ref).abs().max().item(), 2e-4 * math.sqrt(K)
    if form == "res16":
        x = rnd((M, N), 1.0, dt)
        ref = ref + x.float()
        out = strided(x, lo)
        ops.gemm_res16(ad, wd, out, bias=None if b is None else b.to(DEV), res16=out)
        return what, (out.float().cpu() - ref).abs().max().item(), EPS[dt] * ref.abs().max().item() + 1e-3
    # swiglu: N = 2 * fp rows interleaved in 32-row groups [gate | up]
    Fq = 4 * ri(1, 600)
    fp = ops.pad64(Fq)
    g, u = rnd((Fq, K), K ** -0.5, dt), rnd((Fq, K), K ** -0.5, dt)
    ref = F.silu(mm(af, g.float().to(DEV))) * mm(af, u.float().to(DEV))
    gp = F.pad(g, (0, 0, 0, fp - Fq)).reshape(fp // 32, 1, 32, K)
    up = F.pad(u, (0, 0, 0, fp - Fq)).reshape(fp // 32, 1, 32, K)
    gu = torch.cat([gp, up], 1).reshape(2 * fp, K)
    out = strided(torch.full((M, fp), float("nan"), dtype=dt), lo)
    ops.gemm(ad, gu.to(DEV), out, mode=ops.GEMM_SWIGLU)
    got = out.float().cpu()
    pad_ok = bool((got[:, Fq:] == 0).all())
    err = (got[:, :Fq] - ref).abs().max().item()
    return f"gemm swiglu {dt} M={M} F={Fq} K={K} lda={ad.stride(0)}", err if pad_ok else float("inf"), EPS[dt] * ref.abs().max().item() + 1e-3


def ref_attention(q, k, v, scale, nk=None):
    rep = q.shape[1] // k.shape[1]
    k, v = k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1)
    if nk is not None:
        k, v = k[:, :, :nk], v[:, :, :nk]
    return torch.softmax((q @ k.transpose(-1, -2)) * scale, -1) @ v


def fuzz_attention():
    dt = pick(torch.float16, torch.bfloat16)
    D = pick(72, 128)
    HKV = ri(1, 4)
    H = HKV * (1 if D == 72 else pick(1, 2, 7))
    B = ri(1, 3)
    S = pick(ri(1, 70), ri(1, 600), ri(1, 1500), pick(63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025))
    Sk = S if ri(0, 1) else pick(1, ri(1, 70), ri(1, 2000), pick(63, 64, 65, 1024, 1025))
    variant = pick(0, 0, 1)
    dyn = ri(0, 3) == 0
    real = ri(1, Sk) if dyn else Sk
    q, k, v = rnd((B, H, S, D), 1.0, dt), rnd((B, HKV, Sk, D), 1.0, dt), rnd((B, HKV, Sk, D), 1.0, dt)
    if ri(0, 4) == 0:
        k[:, :, ri(0, real - 1)] *= 6.0                                     # a spiked key: forces the running-max rescale
    want = ref_attention(q.float(), k.float(), v.float(), D ** -0.5, real)
    qd = q.permute(0, 2, 1, 3).reshape(B * S, H * D).to(DEV).contiguous()
    kv = torch.cat([k.permute(0, 2, 1, 3).reshape(B * Sk, HKV * D), v.permute(0, 2, 1, 3).reshape(B * Sk, HKV * D)], -1).to(DEV).contiguous()
    if dyn and real < Sk:
        kv.view(B, Sk, -1)[:, real:] = float("nan")                         # rows past the device-side key count must never reach the result
    o = torch.zeros((B * S, H * D), dtype=dt, device=DEV)
    ldk = 2 * HKV * D
    kw = dict(batch=B, heads=H, kv_heads=HKV, seq=S, kv_seq=Sk, head_dim=D, q_strides=(S * H * D, H * D, D), k_strides=(Sk * ldk, ldk, D),
              v_strides=(Sk * ldk, ldk, D), o_strides=(S * H * D, H * D, D), scale=D ** -0.5, variant=variant)
    if dyn:
        kw["kv_len"] = torch.tensor([real], dtype=torch.int32, device=DEV)
    ops.attention(qd, kv, kv[:, HKV * D:], o, **kw)
    got = o.float().cpu().view(B, S, H, D).permute(0, 2, 1, 3)
    err = (got - want).abs().max().item() if torch.isfinite(got).all() else float("inf")
    return f"attention {dt} D={D} B={B} H={H} HKV={HKV} S={S} Sk={Sk} real={real} variant={variant} dyn={dyn}", err, 6 * EPS[dt] * v.float().abs().max().item()


def fuzz_norm():
    kind = pick("ln", "rms", "ln16")
    M, D = rows_m(), (8 if kind == "ln16" else 4) * ri(2, 512 if kind == "ln16" else 1024)
    if M * D > 8e7:
        M = int(8e7 // D)
    w, b = 1 + rnd((D,), 0.2), rnd((D,), 0.2)
    lo = pick(0, 0, 8)
    if kind == "ln16":
        dt = pick(torch.float16, torch.bfloat16)
        x = (rnd((M, D)) * pick(1.0, 8.0) + pick(0.0, 3.0)).to(dt)
        out = strided(torch.zeros((M, D), dtype=dt), lo)
        ops.layernorm16(strided(x, pick(0, 8)), w.to(DEV), b.to(DEV), out, D, 1e-6)
        ref = F.layer_norm(x.float(), (D,), w, b, 1e-6)
        return f"layernorm16 {dt} M={M} D={D}", (out.float().cpu() - ref).abs().max().item(), EPS[dt] * ref.abs().max().item() + 1e-3
    odt = pick(torch.float16, torch.bfloat16, torch.float32)
    x = rnd((M, D)) * pick(1.0, 30.0) + pick(0.0, 5.0)
    out = strided(torch.zeros((M, D), dtype=odt), lo)
    if kind == "ln":
        ops.layernorm(strided(x, pick(0, 4)), w.to(DEV), b.to(DEV), out, D, 1e-6)
        ref = F.layer_norm(x, (D,), w, b, 1e-6)
    else:
        ops.rmsnorm(strided(x, pick(0, 4)), w.to(DEV), out, D, 1e-6)
        ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w
    tol = (EPS[odt] if odt in EPS else 2e-6) * ref.abs().max().item() + (1e-3 if odt in EPS else 2e-5)
    return f"{kind} -> {odt} M={M} D={D}", (out.float().cpu() - ref).abs().max().item(), tol


def fuzz_rank():
    n = pick(ri(1, 40), ri(1, 600), ri(1, 3000))
    lg = rnd((n,), pick(0.5, 3.0, 30.0))
    if ri(0, 1):
        lg = (lg * 4).round() / 4                                           # many exact ties: ties go to the lower index
    if ri(0, 5) == 0:
        lg[ri(0, n - 1)] = float("-inf") if ri(0, 1) else float("inf")
    s, o = ops.rank(lg.to(DEV))
    want_s = torch.sigmoid(lg)
    want_o = torch.sort(want_s, descending=True, stable=True).indices
    ok = torch.equal(o.cpu().long(), want_o) and (s.cpu() - want_s).abs().max().item() <= 1e-6
    if not ok:                                                              # sigmoid on the device may differ by an ulp and reorder near-ties: compare on ITS scores
        ok = torch.equal(o.cpu().long(), torch.sort(s.cpu(), descending=True, stable=True).indices) and (s.cpu() - want_s).abs().max().item() <= 1e-6
    return f"rank n={n}", 0.0 if ok else float("inf"), 0.0


def fuzz_pool_rope():
    if ri(0, 1):
        T, G = ri(1, 6), ri(1, 33)
        HW, D = ri(1, G), 4 * ri(1, 300)
        x = rnd((T * G * G, D))
        pdt = pick(torch.float16, torch.bfloat16)
        out = torch.full((T * HW * HW, ops.pad64(D)), float("nan"), dtype=pdt, device=DEV)
        ops.bilinear_pool(x.to(DEV), out, T, G, HW, D)
        ref = F.interpolate(x.view(T, G, G, D).permute(0, 3, 1, 2), size=(HW, HW), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).reshape(T * HW * HW, D) \
            if HW < G else x
        pad_ok = bool((out[:, D:].float() == 0).all())
        err = (out[:, :D].float().cpu() - ref).abs().max().item()
        return f"bilinear_pool {pdt} T={T} G={G} HW={HW} D={D}", err if pad_ok else float("inf"), EPS[pdt] * ref.abs().max().item() + 1e-6
    dt = pick(torch.float16, torch.bfloat16)
    N, heads, hd = ri(1, 3000), ri(1, 12), pick(128, 64, 32)
    extra = hd * ri(0, 4)
    qkv = rnd((N, heads * hd + extra), 1.0, dt)
    ang = torch.rand((N, hd // 2), generator=gen) * 6.28
    cos, sin = ang.cos(), ang.sin()
    d = qkv.clone().to(DEV)
    ops.rope_(d, cos.to(DEV), sin.to(DEV), heads, hd)
    xf = qkv[:, :heads * hd].float().view(N, heads, hd)
    x1, x2 = xf[..., :hd // 2], xf[..., hd // 2:]
    ref = torch.cat([x1 * cos[:, None] - x2 * sin[:, None], x2 * cos[:, None] + x1 * sin[:, None]], -1).reshape(N, heads * hd)
    got = d.float().cpu()
    untouched = torch.equal(got[:, heads * hd:], qkv[:, heads * hd:].float())
    err = (got[:, :heads * hd] - ref).abs().max().item()
    return f"rope {dt} N={N} heads={heads} hd={hd} extra={extra}", err if untouched else float("inf"), EPS[dt] * ref.abs().max().item() + 1e-3


def fuzz_split():
    kind = pick("gemm", "attention", "norm")
    if kind == "norm":
        M, D = ri(1, 3000), 4 * ri(2, 900)
        Dp = ops.pad64(D)
        x = rnd((M, D)) * pick(1.0, 30.0)
        x[:, ri(0, D - 1)] *= pick(1.0, 300.0)
        w, b = 1 + rnd((D,), 0.2), (rnd((D,), 0.2) if ri(0, 1) else None)
        out = torch.zeros((M, 2 * Dp), dtype=torch.float16, device=DEV)
        ops.norm_split(x.to(DEV), w.to(DEV), None if b is None else b.to(DEV), out, D, 1e-6)
        x64 = x.double()
        ref = (F.layer_norm(x64, (D,), w.double(), b.double(), 1e-6) if b is not None else x64 * torch.rsqrt(x64.pow(2).mean(-1, keepdim=True) + 1e-6) * w.double())
        got = out[:, :D].double().cpu() + out[:, Dp:Dp + D].double().cpu()
        return f"norm_split M={M} D={D} rms={b is None}", ((got - ref).abs().max() / ref.abs().max()).item(), 2e-6
    if kind == "gemm":
        M, N, Kh = ri(1, 3000), 64 * ri(1, 24), 64 * ri(1, 20)
        a = rnd((M, Kh)) * pick(1.0, 10.0)
        hi = a.half()
        lo = (a - hi.float()).half()
        w = rnd((N, Kh), Kh ** -0.5, torch.float16)
        b = rnd((N,), 0.1)
        A2 = torch.cat([hi, lo], 1).to(DEV)
        W2 = torch.cat([w, w], 1).to(DEV).contiguous()
        out = torch.zeros((M, 2 * N), dtype=torch.float16, device=DEV)
        act = pick(ops.ACT_NONE, ops.ACT_GELU_TANH)
        ops.gemm_split16(A2, W2, out, mode=ops.GEMM_STORE16, lo_off=N, bias=b.to(DEV), act=act)
        ref = (hi.double() + lo.double()) @ w.double().t() + b.double()
        if act == ops.ACT_GELU_TANH:
            ref = F.gelu(ref, approximate="tanh")
        got = out[:, :N].double().cpu() + out[:, N:].double().cpu()
        return f"gemm_split16 M={M} N={N} K=2x{Kh} act={act}", ((got - ref).abs().max() / ref.abs().max()).item(), 3e-6
    D = pick(72, 128)
    HKV = ri(1, 3)
    H = HKV * (1 if D == 72 else pick(1, 2, 7))
    B, S, Sk = ri(1, 2), ri(1, 700), pick(1, ri(1, 70), ri(1, 1500), pick(63, 64, 65))
    mk = lambda shape: rnd(shape)                                           # noqa: E731
    q, k, v = mk((B, H, S, D)), mk((B, HKV, Sk, D)), mk((B, HKV, Sk, D))

    def pair(t, rows, heads):                                               # [B, heads, rows, D] -> [B * rows, 2 * heads * D] = [hi | lo]
        f = t.permute(0, 2, 1, 3).reshape(B * rows, heads * D)
        h = f.half()
        return torch.cat([h, (f - h.float()).half()], 1).to(DEV).contiguous(), h.double() + (f - h.float()).half().double()
    q2, qv = pair(q, S, H)
    k2, kvv = pair(k, Sk, HKV)
    v2, vv = pair(v, Sk, HKV)
    Dq = ops.pad64(H * D)
    o = torch.zeros((B * S, 2 * Dq), dtype=torch.float16, device=DEV)
    ops.attention_split(q2, k2, v2, o, lo=(H * D, HKV * D, HKV * D, Dq), batch=B, heads=H, kv_heads=HKV, seq=S, kv_seq=Sk, head_dim=D,
                        q_strides=(S * 2 * H * D, 2 * H * D, D), k_strides=(Sk * 2 * HKV * D, 2 * HKV * D, D), v_strides=(Sk * 2 * HKV * D, 2 * HKV * D, D),
                        o_strides=(S * 2 * Dq, 2 * Dq, D), scale=D ** -0.5)
    ref = ref_attention(qv.view(B, S, H, D).permute(0, 2, 1, 3), kvv.view(B, Sk, HKV, D).permute(0, 2, 1, 3), vv.view(B, Sk, HKV, D).permute(0, 2, 1, 3), D ** -0.5)
    got = (o[:, :H * D].double().cpu() + o[:, Dq:Dq + H * D].double().cpu()).view(B, S, H, D).permute(0, 2, 1, 3)
    return f"attention_split D={D} B={B} H={H} HKV={HKV} S={S} Sk={Sk}", ((got - ref).abs().max() / ref.abs().max()).item(), 3e-6


def fuzz_patch_embed():
    """vitg_patch_embed16 (im2col inside the GEMM's operand loader) against the stride-p convolution + bias + position table in fp32"""
    dt = pick(torch.float16, torch.bfloat16)
    ps = pick(14, 14, 8, 10, 12, 16)
    G = ri(1, 12)
    R = G * ps + 2 * ri(0, (ps - 1) // 2)                                   # even sides; a ragged border the VALID convolution ignores
    T, N = ri(1, 6), 4 * ri(64, 320)
    if not ops.patch_embed_fusable(T, R, ps, N):
        return f"patch_embed16 (not fusable: T={T} R={R} p={ps} N={N})", 0.0, 0.0
    P = G * G
    pix, w = rnd((T, 3, R, R), 1.0, dt), rnd((N, 3, ps, ps), 0.05, dt)
    b, pos = (rnd((N,), 0.1) if ri(0, 1) else None), (rnd((P, N), 0.5) if ri(0, 1) else None)
    want = F.conv2d(pix.float().to(DEV), w.float().to(DEV), stride=ps).flatten(2).transpose(1, 2).reshape(-1, N).cpu()
    if b is not None:
        want = want + b
    if pos is not None:
        want = want + pos.repeat(T, 1)
    rows = w.reshape(N, 3 * ps, ps)
    wf = torch.zeros((N, (3 * ps + 3) // 4 * 4, 16), dtype=dt)
    wf[:, :3 * ps, :8] = rows[:, :, :8]
    wf[:, :3 * ps, 16 - (ps - 8):] = rows[:, :, 8:]
    buf = torch.full((T * 3 * R * R + 128,), float("nan"), dtype=dt, device=DEV)      # NaN guards around the pixels: a loader chunk past a row's end would show
    dpix = buf[64:64 + T * 3 * R * R].view(T, 3, R, R)
    dpix.copy_(pix)
    out = torch.full((T * P, N), float("nan"), dtype=dt, device=DEV)
    ops.patch_embed16(dpix, wf.reshape(N, -1).contiguous().to(DEV), out, patch=ps, bias=None if b is None else b.to(DEV), pos=None if pos is None else pos.to(DEV))
    got = out.float().cpu()
    if not torch.isfinite(got).all():
        return f"patch_embed16 {dt} T={T} R={R} p={ps} N={N}: non-finite output", float("inf"), 0.0
    excess = ((got - want).abs() - (EPS[dt] * want.abs().clamp(min=1.0) + 1e-3)).max().item()
    return f"patch_embed16 {dt} T={T} R={R} p={ps} N={N} bias={b is not None} pos={pos is not None}", max(excess, 0.0), 0.0


def fuzz_preprocess():
    """vitg_preprocess_u8 against the host twin of the real SiglipImageProcessor (Pillow bicubic + transformers' rescale / normalise): bit-exact"""
    import numpy as np
    from videoitg_amd.image_processor import SiglipImageProcessor
    T, H, W = ri(1, 4), pick(ri(1, 40), ri(1, 400), ri(100, 1100)), pick(ri(1, 40), ri(1, 400), ri(100, 1300))
    S = pick(ri(1, 64), 56, 224, 384, 448, ri(30, 500))
    fr = torch.randint(0, 256, (T, H, W, 3), generator=gen, dtype=torch.uint8)
    if ri(0, 1):                                                            # smooth content beside noise: long runs of equal bytes
        fr = (fr.float().cumsum(2) / torch.arange(1, W + 1)[None, None, :, None]).round().clamp(0, 255).to(torch.uint8)
    want = torch.from_numpy(np.stack(SiglipImageProcessor(size=S).preprocess(fr.numpy(), return_tensors="np")["pixel_values"]))
    try:
        got = ops.preprocess_u8(fr.to(DEV), S, torch.float32).cpu()
    except Exception as e:
        return f"preprocess_u8 T={T} {H}x{W} -> {S}: {type(e).__name__}: {str(e)[:120]}", float("inf"), 0.0
    return f"preprocess_u8 T={T} {H}x{W} -> {S}", 0.0 if torch.equal(got, want) else float("inf"), 0.0


def fuzz_fp8():
    """the per-token e4m3 quantiser against its arithmetic in torch (bit-exact bytes and scales: scale = amax * fp32(1/448), q = RNE(x * (1 / scale))), then
    vitg_gemm_fp8 on those bytes against the fp32 product of the dequantised operands"""
    dt = pick(torch.float16, torch.bfloat16)
    M, N, K = pick(ri(1, 40), ri(1, 2500)), 8 * ri(2, 350), 128 * ri(1, 24)                  # (vitg_gemm_fp8: N % 8, K % 128)
    x = rnd((M, K), pick(0.05, 1.0, 30.0), pick(torch.float32, dt))
    x[:, ri(0, K - 1)] *= pick(1.0, 50.0)
    if ri(0, 5) == 0:
        x[ri(0, M - 1)] = 0                                                 # an all-zero token: scale 1, bytes 0
    a8, sa = ops.quant_rows_fp8(x.to(DEV))
    xf = x.float()
    amax = xf.abs().amax(-1, keepdim=True)
    sc = torch.where(amax > 0, amax * torch.tensor(1.0 / 448.0, dtype=torch.float32), torch.ones_like(amax))
    q = (xf * (1.0 / sc)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    ok = torch.equal(a8[:, :K].cpu().view(torch.float8_e4m3fn).view(torch.uint8), q.view(torch.uint8)) and torch.equal(sa.cpu(), sc.view(-1))
    if not ok:
        return f"quant_rows_fp8 {x.dtype} M={M} K={K}: bytes / scales differ", float("inf"), 0.0
    w = rnd((N, K), K ** -0.5)
    w8, sw = ops.quant_rows_fp8(w.to(DEV))
    wq = w8[:, :K].cpu().view(torch.float8_e4m3fn).float() * sw.cpu()[:, None]
    ad_ = (q.float() * sc).to(DEV)
    ref = (ad_ @ wq.to(DEV).t()).cpu()
    # the f8 MFMA is not an fp32 dot product: inside a 128-deep block the products are aligned to the largest one and ~13 bits below it survive
    # (tools/r06/fp8_f32_probe.py, gpurun_out/r06_c34: error up to 1.5e-4 of sum |terms| with one heavy column, 4e-6 on benign operands - against e4m3's own
    # 2^-4 per operand that is nothing, but it is what an fp32-accurate reference sees).  Bound: 4e-4 of the largest sum |terms| of the call.
    noise = 4e-4 * (ad_.abs() @ wq.to(DEV).abs().t()).max().item()
    b = rnd((N,), 0.1) if ri(0, 1) else None
    if b is not None:
        ref = ref + b
    if ri(0, 1):
        out = torch.full((M, N), float("nan"), dtype=dt, device=DEV)
        ops.gemm_fp8(a8, sa, w8, sw, out, mode=ops.GEMM_STORE16, out_dtype=dt, bias=None if b is None else b.to(DEV))
        return f"gemm_fp8 store16 {dt} M={M} N={N} K={K}", (out.float().cpu() - ref).abs().max().item(), EPS[dt] * ref.abs().max().item() + 2e-3 + noise
    r = rnd((M, N))
    out = r.to(DEV)
    ops.gemm_fp8(a8, sa, w8, sw, out, mode=ops.GEMM_F32, out_dtype=dt, bias=None if b is None else b.to(DEV), res=out)
    return f"gemm_fp8 f32 M={M} N={N} K={K}", (out.cpu() - ref - r).abs().max().item(), noise + 1e-4


def fuzz_misc():
    """the small kernels at the path's edges: embedding gather (exact), row gather (exact), per-frame mean + score head, patchify (exact copy into the im2col
    layout), the grounding loss"""
    kind = pick("embed", "gather", "pool_score", "patchify", "bce")
    if kind == "embed":
        dt = pick(torch.float16, torch.bfloat16)
        V, D, n = ri(1, 2000), 4 * ri(1, 1000), ri(1, 300)
        table = rnd((V, D), 1.0, dt)
        ids = torch.randint(0, V, (n,), generator=gen)
        out = torch.full((n, D + pick(0, 4)), float("nan"), device=DEV)
        ops.embed_gather(ids.to(DEV), table.to(DEV), out)
        return f"embed_gather {dt} V={V} D={D} n={n}", 0.0 if torch.equal(out[:, :D].cpu(), table[ids].float()) else float("inf"), 0.0
    if kind == "gather":
        dt = pick(torch.float32, torch.float16)
        B, R_, W = ri(1, 5), ri(1, 400), 8 * ri(1, 250)                     # (rows move as 16-byte chunks: width and strides multiples of 16 bytes)
        src = rnd((B * R_, W + pick(0, 8, 16)), 1.0, dt)
        idx = torch.randint(0, R_, (ri(1, R_),), generator=gen, dtype=torch.int32)
        out = torch.zeros((B * idx.numel(), W), dtype=dt, device=DEV)
        ops.gather_rows(src.to(DEV), idx.to(DEV), R_, B, out, W)
        want = torch.cat([src[b * R_ + idx.long(), :W] for b in range(B)], 0)
        return f"gather_rows {dt} B={B} rows={R_} W={W} n={idx.numel()}", 0.0 if torch.equal(out.cpu(), want) else float("inf"), 0.0
    if kind == "pool_score":
        T, tok, D = ri(1, 600), ri(1, 300), 4 * ri(1, 900)
        if T * tok * D > 4e7:
            T = max(1, int(4e7 // (tok * D)))
        h = rnd((T * tok, D))
        w, b = rnd((D,), 0.1), rnd((1,), 1.0)
        lg = torch.empty(T, device=DEV)
        ops.pool_score(strided(h, pick(0, 4)), T, tok, D, w.to(DEV), b.to(DEV), lg)
        want = (h.double().view(T, tok, D).mean(1) @ w.double() + b.double()).float()
        return f"pool_score T={T} tokens={tok} D={D}", (lg.cpu() - want).abs().max().item(), 2e-6 * math.sqrt(D) * max(1.0, want.abs().max().item()) + 1e-5
    if kind == "patchify":
        dt, odt = pick(torch.float32, torch.float16, torch.bfloat16), pick(torch.float16, torch.bfloat16)
        ps, G, T = pick(14, 14, 16, 8), ri(1, 12), ri(1, 5)
        R = G * ps + ri(0, ps - 1)
        kpad = ops.pad64(3 * ps * ps) + pick(0, 64)
        pix = rnd((T, 3, R, R), 1.0, dt)
        out = ops.patchify(pix.to(DEV), ps, kpad, odt)
        want = pix[:, :, :G * ps, :G * ps].reshape(T, 3, G, ps, G, ps).permute(0, 2, 4, 1, 3, 5).reshape(T * G * G, 3 * ps * ps).to(odt)
        ok = torch.equal(out[:, :3 * ps * ps].cpu(), want) and bool((out[:, 3 * ps * ps:] == 0).all())
        return f"patchify {dt}->{odt} T={T} R={R} p={ps} kpad={kpad}", 0.0 if ok else float("inf"), 0.0
    n = ri(1, 3000)
    lg = rnd((n,), pick(0.5, 3.0, 20.0))
    y = (torch.rand((n,), generator=gen) < pick(0.0, 0.02, 0.3, 1.0)).float()
    loss, pw = ops.bce_loss(lg.to(DEV), y.to(DEV))
    pos = y.sum()
    wpos = min(5.0, math.sqrt((n - pos.item()) / max(1.0, pos.item())))
    want = F.binary_cross_entropy_with_logits(lg.double(), y.double(), pos_weight=torch.tensor(wpos, dtype=torch.float64)).item()
    err = abs(loss.item() - want) / max(abs(want), 1e-6) + abs(float(pw) - wpos)
    return f"bce_loss n={n} pos={int(pos.item())}", err, 2e-5


def fuzz_splitk():
    """the K-split forms (fp32 and 16-bit output) against the plain GEMM of the same call: equal up to fp32 summation order, and deterministic"""
    dt = pick(torch.float16, torch.bfloat16)
    M, N, K = ri(1, 3000), 4 * ri(16, 900), 64 * ri(2, 150)
    splits = ri(2, min(6, K // 64))
    a, w = rnd((M, K), 1.0, dt).to(DEV), rnd((N, K), K ** -0.5, dt).to(DEV)
    b = rnd((N,), 0.1).to(DEV) if ri(0, 1) else None
    ws = torch.empty((splits * M * N,), device=DEV)
    x = rnd((M, N)).to(DEV)
    plain = x.clone()
    ops.gemm(a, w, plain, mode=ops.GEMM_F32, bias=b, res=plain)
    o1, o2 = x.clone(), x.clone()
    try:
        ops.gemm_splitk(a, w, o1, ws, splits, bias=b, res=o1)
        ops.gemm_splitk(a, w, o2, ws, splits, bias=b, res=o2)
    except Exception as e:
        if "SHAPE" in str(e).upper() or "shape" in str(e):
            return f"gemm_splitk refused M={M} N={N} K={K} splits={splits}", 0.0, 0.0
        raise
    det = torch.equal(o1, o2)
    err = (o1 - plain).abs().max().item()
    return f"gemm_splitk {dt} M={M} N={N} K={K} splits={splits} bias={b is not None}", err if det else float("inf"), 2e-4 * math.sqrt(K)


FAMILIES = [("gemm", fuzz_gemm), ("attention", fuzz_attention), ("norm", fuzz_norm), ("rank", fuzz_rank), ("pool_rope", fuzz_pool_rope), ("split", fuzz_split),
            ("patch_embed", fuzz_patch_embed), ("preprocess", fuzz_preprocess), ("splitk", fuzz_splitk), ("fp8", fuzz_fp8), ("misc", fuzz_misc)]


def main():
    bad_total = 0
    for name, fn in FAMILIES:
        if only and name not in only:
            continue
        t0, n, bad, worst = time.time(), 0, [], 0.0
        while time.time() - t0 < budget:
            try:
                what, err, tol = fn()
            except Exception as e:                                          # a refusal of a shape this tool thought legal is a finding too
                what, err, tol = f"{name}: {type(e).__name__}: {str(e)[:200]}", float("inf"), 0.0
            torch.cuda.synchronize()
            n += 1
            if not (err <= tol):
                bad.append((what, err, tol))
            elif tol > 0:
                worst = max(worst, err / tol)
        print(f"{name}: {n} cases in {time.time() - t0:.0f} s, {len(bad)} outside their bound (worst passing case at {worst:.2f} of its bound)", flush=True)
        for what, err, tol in bad[:12]:
            print(f"    FAIL {what}: err {err:.3e} > {tol:.3e}", flush=True)
        bad_total += len(bad)
    print(f"fuzz: {'clean' if not bad_total else f'{bad_total} failures'} (seed {seed}, {budget:.0f} s per family)", flush=True)
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
