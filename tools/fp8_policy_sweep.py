#!/usr/bin/env python3
"""Round 3, VERDICT item 5: which layer GEMMs can run on the fp8 weight path?  Per-GEMM precision policies on the headline workload
(512 x 448^2, complete-pass oracle fixture): frames/s, max |dscore| and Top-32 overlap against the fp32 oracle, and against the 16-bit
HIP result.  Output = the rows of the Pareto table in profiles/r03_fp8_pareto.md.
    python tools/fp8_policy_sweep.py [case]"""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videoitg_amd import synth
from videoitg_amd.config import ITGConfig
from videoitg_amd.engine import SelectorEngine
from videoitg_amd.synth_device import packed_random_weights
case = sys.argv[1] if len(sys.argv) > 1 else "c2_512x448"
rec = json.load(open(os.path.join(ROOT, "tests", "golden", f"full_{case}.json")))
cfg = ITGConfig(); dev = torch.device("cuda:0")
ids, pix = synth.fixture_inputs(cfg, rec)          # (planted-signal fixtures, e.g. c2p_512x448, rebuild their own pixels)
want = torch.tensor(rec["logits"]); pix = pix.half().to(dev)
POLICIES = [("16-bit (fp16, tower stream fp16)", False),
            ("llm.gu+down", "llm.gu,llm.down"), ("llm (all four)", "llm"), ("llm.qkv+o", "llm.qkv,llm.o"),
            ("vit.fc1", "vit.fc1"), ("vit.qkv", "vit.qkv"), ("vit.fc1+fc2 (MX hand-over)", "vit.fc1,vit.fc2"), ("vit.out", "vit.out"), ("vit.fc2", "vit.fc2"),
            ("vit.qkv+fc1", "vit.qkv,vit.fc1"), ("vit.qkv+fc1 + llm", "vit.qkv,vit.fc1,llm"), ("vit.fc1 + llm", "vit.fc1,llm"),
            ("vit (all four)", "vit"), ("all eight", True)]
ref16 = None
print(f"case {case}: oracle rank-{rec['k']}/{rec['k'] + 1} score gap {rec['gap_k_k1']}")
oord, osc = rec["order"], rec["scores"]
print("oracle score gaps rank k / k+1: " + ", ".join(f"k={k}: {osc[oord[k - 1]] - osc[oord[k]]:.2e}" for k in (1, 2, 4, 8, 16, 32)))
print("| policy (GEMMs on e4m3 weights + e4m3 activations) | ms/step | frames/s | max dscore vs oracle | Top-32 overlap vs oracle | Top-1/2/4/8/16 kept vs oracle | max dscore vs 16-bit HIP | Top-32 overlap vs 16-bit HIP |")
print("|---|---|---|---|---|---|---|---|")
for name, fp8 in POLICIES:
    eng = SelectorEngine(packed_random_weights(cfg, torch.float16, dev, seed=rec["weight_seed"], fp8=fp8))
    lg = eng.forward(ids, None, [pix])[0].view(-1).float().cpu()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): eng.forward(ids, None, [pix])
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    sc = torch.sigmoid(lg)
    ds = (sc - torch.sigmoid(want)).abs().max().item()
    mine = torch.sort(sc, descending=True, stable=True).indices.tolist()
    top = set(mine[:rec["k"]])
    ladder = "/".join(str(len(set(mine[:k]) & set(oord[:k]))) for k in (1, 2, 4, 8, 16))
    if ref16 is None:
        ref16 = (sc.clone(), set(top))
    d16 = (sc - ref16[0]).abs().max().item()
    dlg = (lg - want).abs().max().item()
    print(f"| {name} | {ms:.1f} | {rec['frames'] / ms * 1e3:.0f} | {ds:.2e} (logit {dlg:.2e}) | {len(top & set(rec['topk']))}/{rec['k']} | {ladder} | {d16:.2e} | {len(top & ref16[1])}/{rec['k']} |", flush=True)
    eng.release_workspace(); del eng; torch.cuda.empty_cache()
