#!/usr/bin/env python3
"""hd-128 attention: the one-wave-per-SIMD kernel (attention_w4.hip: attn_sp = 4; research forms 5 in-place score
tiles, 6 / 7 staging from step 0 / 16 - profiling library) against the 8-wave software-pipelined kernel (attn_sp = 0, the default), same call, alternating.  Correctness first (fp64 reference at small S incl. ragged tails; default kernel at the LLM shape)."""
import os, sys, torch
os.environ.setdefault("VITG_PROFILING_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd import ops

def run(qkv, o, B, H, HKV, S, D, sp):
    ops.set_option("attn_sp", sp)
    ld = (H + 2 * HKV) * D
    ops.attention(qkv, qkv[:, H * D:], qkv[:, (H + HKV) * D:], o, batch=B, heads=H, kv_heads=HKV, seq=S, head_dim=D,
                  q_strides=(S * ld, ld, D), k_strides=(S * ld, ld, D), v_strides=(S * ld, ld, D), o_strides=(S * H * D, H * D, D),
                  scale=D ** -0.5, variant=0)

def ref(qkv, B, H, HKV, S, D):
    x = qkv.double().view(B, S, H + 2 * HKV, D)
    q, k, v = x[:, :, :H], x[:, :, H:H + HKV], x[:, :, H + HKV:]
    g = H // HKV
    k = k.repeat_interleave(g, 2); v = v.repeat_interleave(g, 2)
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * D ** -0.5
    return torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v).reshape(B * S, H * D)

def main():
    ok = True
    for dt in (torch.float16, torch.bfloat16):
        for (B, H, HKV, S, scale) in [(1, 4, 2, 1000, 1.0), (2, 4, 4, 257, 1.0), (1, 4, 1, 64, 1.0), (1, 2, 1, 63, 1.0), (1, 2, 1, 128, 1.0), (1, 2, 1, 100, 1.0), (1, 2, 1, 192, 1.0), (1, 2, 1, 130, 1.0), (1, 2, 1, 256, 1.0), (1, 2, 1, 1, 1.0), (1, 4, 2, 1300, 6.0), (1, 2, 2, 2049, 1.0)]:
            D = 128
            torch.manual_seed(S)
            qkv = (torch.randn((B * S, (H + 2 * HKV) * D), device="cuda") * scale).to(dt)
            r = ref(qkv, B, H, HKV, S, D)
            errs = []
            for sp in (0, 4, 4):
                o = torch.full((B * S, H * D), float("nan"), dtype=dt, device="cuda")
                run(qkv, o, B, H, HKV, S, D, sp)
                errs.append((o.double() - r).abs().max().item())
            tol = 2e-3 if dt == torch.float16 else 1.6e-2
            good = all(e == e and e < tol * max(1.0, scale) ** 2.5 for e in errs)
            ok &= good
            print(f"{str(dt):15s} B={B} H={H}/{HKV} S={S} scale={scale}: max|err| sp0 {errs[0]:.2e}  w4 {errs[1]:.2e}  w4 {errs[2]:.2e}  {'ok' if good else 'FAIL'}", flush=True)
    print("CORRECTNESS", "PASS" if ok else "FAIL", flush=True)

    B, H, HKV, S, D = 1, 28, 4, 12860, 128
    for dt in (torch.float16,):
        qkv = torch.randn((B * S, (H + 2 * HKV) * D), device="cuda").to(dt)
        outs = {}
        for sp in (0, 4, 4):
            o = torch.empty((B * S, H * D), dtype=dt, device="cuda")
            run(qkv, o, B, H, HKV, S, D, sp)
            outs[sp] = o.float()
        print("LLM shape: max|w4 - sp|", (outs[4] - outs[0]).abs().max().item(), flush=True)
        o = torch.empty((B * S, H * D), dtype=dt, device="cuda")
        for rnd in range(4):
            line = []
            for sp in (0, 4, 5, 6, 7):
                for _ in range(2):
                    run(qkv, o, B, H, HKV, S, D, sp)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(qkv, o, B, H, HKV, S, D, sp)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                line.append(f"sp={sp}: {ms:6.3f} ms {4.0 * B * H * S * S * D / ms / 1e9:7.1f} TF/s")
            print(f"round {rnd}: " + "   ".join(line), flush=True)
    # per-rank shapes of the context-parallel LLM (queries 1660, keys 12860)
    ops.set_option("attn_sp", 0)


if __name__ == "__main__":
    main()
