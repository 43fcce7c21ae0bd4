#!/usr/bin/env python3
"""GPU: elementwise precision of every operand-split kernel on BENIGN data (no heavy columns to hide behind): error of (hi + lo) against fp64,
relative to the rms of the output, plus the share of lo values that are exactly zero / subnormal (a flushed conversion would show as 100 % zero)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd.torch_ops import K

dev = "cuda:0"
torch.manual_seed(0)


def joined(t, w, d):
    return t[:, :d].double() + t[:, w:w + d].double()


def split_pair(x, width):
    out = torch.zeros((x.shape[0], 2 * width), dtype=torch.float16, device=dev)
    hi = x.half()
    out[:, :x.shape[1]] = hi
    out[:, width:width + x.shape[1]] = (x - hi.float()).half()
    return out


def report(name, got, ref, lo):
    e = (got - ref).abs()
    sub = ((lo != 0) & (lo.abs() < 2.0 ** -14)).float().mean().item()
    zero = (lo == 0).float().mean().item()
    print(f"{name:42s} max {e.max().item():.2e}  rms {e.pow(2).mean().sqrt().item():.2e}  / rms(ref) {ref.pow(2).mean().sqrt().item():.2e}  -> rel rms {(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():.2e}"
          f"   lo: {zero:.0%} zero, {sub:.0%} subnormal")


for sc in (1.0, 0.05):
    M, D = 1000, 1152
    x = torch.randn(M, D, device=dev) * sc
    w, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
    out = torch.zeros(M, 2 * D, dtype=torch.float16, device=dev)
    K.norm_split(x, w, b, out, D, 1e-6)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-6)
    report(f"layernorm_split (x ~ {sc:g})", joined(out, D, D), ref, out[:, D:])
    # GEMM with split output
    Kd, N = 384, 768
    a = torch.randn(M, Kd, device=dev) * sc
    wt = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).half()
    a2 = split_pair(a, Kd)
    o2 = torch.zeros(M, 2 * N, dtype=torch.float16, device=dev)
    for act, an in ((0, "none"), (1, "gelu_tanh")):
        K.gemm_split16(a2, torch.cat([wt, wt], 1).contiguous(), o2, mode=K.GEMM_STORE16, lo_off=N, act=act)
        pre = joined(a2, Kd, Kd) @ wt.double().t()
        ref = pre if act == 0 else 0.5 * pre * (1.0 + torch.tanh(0.7978845608028654 * (pre + 0.044715 * pre ** 3)))
        report(f"gemm_split16 act={an} (a ~ {sc:g})", joined(o2, N, N), ref, o2[:, N:])
    # attention
    for hd, H, KV, S in ((72, 2, 2, 16), (72, 4, 4, 1024), (128, 4, 2, 523)):
        wq, wk = H * hd, KV * hd
        q, k, v = (torch.randn(S, wq, device=dev) * sc), (torch.randn(S, wk, device=dev) * sc), torch.randn(S, wk, device=dev) * sc
        q2, k2, v2 = split_pair(q, wq), split_pair(k, wk), split_pair(v, wk)
        op = K.pad64(wq)
        o2 = torch.zeros(S, 2 * op, dtype=torch.float16, device=dev)
        K.attention_split(q2, k2, v2, o2, lo=(wq, wk, wk, op), batch=1, heads=H, kv_heads=KV, seq=S, head_dim=hd, q_strides=(0, 2 * wq, hd),
                          k_strides=(0, 2 * wk, hd), v_strides=(0, 2 * wk, hd), o_strides=(0, 2 * op, hd), scale=hd ** -0.5)
        qe = joined(q2, wq, wq).view(S, H, hd).transpose(0, 1)
        ke = joined(k2, wk, wk).view(S, KV, hd).transpose(0, 1).repeat_interleave(H // KV, 0)
        ve = joined(v2, wk, wk).view(S, KV, hd).transpose(0, 1).repeat_interleave(H // KV, 0)
        ref = (torch.softmax(qe @ ke.transpose(-1, -2) * hd ** -0.5, -1) @ ve).transpose(0, 1).reshape(S, wq)
        report(f"attention_split hd {hd} S {S} (~ {sc:g})", joined(o2, op, wq), ref, o2[:, op:op + wq])
