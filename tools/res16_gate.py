#!/usr/bin/env python3
"""Round 3, VERDICT item 3: A/B of the tower's 16-bit residual stream against the fp32 one.  Gate: every committed complete-pass fixture
(tests/golden/full_*.json) within 1e-3 of the oracle's scores with an identical Top-32 set; then the step time of both, alternating.
    python tools/res16_gate.py [fp16]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoitg_amd import ops, synth
from videoitg_amd.config import ITGConfig
from videoitg_amd.engine import SelectorEngine
from videoitg_amd.synth_device import packed_random_weights

dt = torch.float16
dev = torch.device("cuda:0")
cfg = ITGConfig()
gdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
engs = {}
rows = []
for fn in sorted(os.listdir(gdir)):
    if not (fn.startswith("full_") and fn.endswith(".json")):
        continue
    rec = json.load(open(os.path.join(gdir, fn)))
    seed = rec["weight_seed"]
    if seed not in engs:
        engs.clear(); torch.cuda.empty_cache()
        engs[seed] = SelectorEngine(packed_random_weights(cfg, dt, dev, seed=seed))
    eng = engs[seed]
    ids, pix = synth.workload_inputs(cfg, rec["frames"], rec["resolution"], rec["prompt_len"], rec["pixel_seed"])
    pix = pix.to(dt).to(dev)
    want = torch.tensor(rec["logits"])
    line = f"{fn:28s}"
    for r16 in (False, True):
        eng.residual16 = r16
        got = eng.forward(ids, None, [pix])[0].view(-1)
        _, order = ops.rank(got)
        got = got.float().cpu()
        ds = (torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item()
        same = sorted(order.cpu().tolist()[:rec["k"]]) == rec["topk"]
        line += f"   {'res16' if r16 else 'fp32 '} max|dscore| {ds:.2e} top{rec['k']} {'same' if same else 'DIFF'} (gap {rec['gap_k_k1']})"
    print(line, flush=True)
    eng.release_workspace()
# step time, alternating, headline workload
eng = next(iter(engs.values()))
ids, pix = synth.workload_inputs(cfg, 512, 448, 60)
pix = pix.to(dt).to(dev)
for rnd_ in range(3):
    for r16 in (False, True):
        eng.residual16 = r16
        eng.forward(ids, None, [pix]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.forward(ids, None, [pix])
        torch.cuda.synchronize()
        print(f"round {rnd_} {'res16' if r16 else 'fp32 '}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per step", flush=True)
