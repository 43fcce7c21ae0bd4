#!/usr/bin/env python3
"""Round 3: what the tower GEMMs fetch through L2 misses and what the output stores cost (profiling library, results stay correct).

Variants of gemm_flags on the step's tower shapes with their real epilogues, interleaved in one process (median of rounds):
    base   shipped kernel
    sc1    output stores write-through, the L2 drops the line (bits 22-23 = 1)
    nt     non-temporal output stores (= 2)
    sc0sc1 (= 3)
    sync   the 32 blocks of an XCD rendezvous once per round of the tile list (bit 26): keeps them inside the K-window the L2 holds
    sync+sc1
  python tools/gemm_l2_ab.py [rounds]          timing table
  python tools/gemm_l2_ab.py pmc <variant> <shape>   one variant, few launches: run under `rocprofv3 --pmc FETCH_SIZE` / WRITE_SIZE
"""
import os, sys, statistics
os.environ["VITG_PROFILING_LIB"] = "1"
os.environ["VITG_PROFILING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoitg_amd import ops

dt = torch.float16
VARIANTS = {"base": 0, "sc1": 1 << 22, "nt": 2 << 22, "sc0sc1": 3 << 22, "sync": 1 << 26, "sync+sc1": (1 << 26) | (1 << 22)}
SHAPES = {"vit_qkv": (524288, 3456, 1152, 0, 0), "vit_fc1": (524288, 4352, 1152, 0, ops.ACT_GELU_TANH),
          "vit_out": (524288, 1152, 1152, 1, 0), "vit_fc2": (524288, 1152, 4352, 1, 0),
          "llm_gu": (12860, 37888, 3584, 2, 0), "llm_down": (12860, 3584, 18944, 1, 0)}


def setup(name):
    M, N, K, mode, act = SHAPES[name]
    a = torch.randn((M, K), device="cuda").to(dt)
    w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(dt)
    bias = None if mode == 2 else torch.randn(N, device="cuda") * 0.02
    out = torch.zeros((M, N // 2 if mode == 2 else N), dtype=torch.float32 if mode == 1 else dt, device="cuda")
    fn = lambda: ops.gemm(a, w, out, mode=mode, bias=bias, act=act, res=out if mode == 1 else None)
    return fn, 2.0 * M * N * K, out


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


if len(sys.argv) > 1 and sys.argv[1] == "pmc":
    var, shape = sys.argv[2], sys.argv[3]
    fn, fl, _ = setup(shape)
    ops.set_option("gemm_flags", VARIANTS[var])
    for _ in range(4): fn()
    torch.cuda.synchronize()
    sys.exit(0)

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for shape in SHAPES:
    fn, fl, out = setup(shape)
    # results of every variant must equal the shipped kernel's (mode 1 accumulates into out: compare on a fresh zero tensor)
    ref = None
    if SHAPES[shape][3] != 1:
        ops.set_option("gemm_flags", 0); fn(); torch.cuda.synchronize(); ref = out[:4096].clone()
    res = {v: [] for v in VARIANTS}
    for r in range(rounds):
        for v, fl_ in VARIANTS.items():
            ops.set_option("gemm_flags", fl_)
            res[v].append(timeit(fn))
            if ref is not None and r == 0:
                assert torch.equal(out[:4096], ref), (shape, v)
    ops.set_option("gemm_flags", 0)
    base = statistics.median(res["base"])
    print(f"{shape:9s} " + "  ".join(f"{v} {fl / statistics.median(t) / 1e9:6.0f} ({base / statistics.median(t) - 1:+.1%})" for v, t in res.items()) + "  TF/s", flush=True)
    del fn, out
