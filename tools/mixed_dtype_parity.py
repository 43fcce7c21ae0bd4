#!/usr/bin/env python3
"""Which half of the path carries the bf16 score error?  Runs the headline workload with the tower in one 16-bit type and the
LLM in the other (features cross over as fp32) and compares every combination with the complete fp32 oracle pass
(tests/golden/full_c2_512x448.json).   python tools/mixed_dtype_parity.py [case]"""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videoitg_amd import synth
from videoitg_amd.config import ITGConfig
from videoitg_amd.engine import SelectorEngine
from videoitg_amd.synth_device import packed_random_weights

case = sys.argv[1] if len(sys.argv) > 1 else "c2_512x448"
rec = json.load(open(os.path.join(ROOT, "tests", "golden", f"full_{case}.json")))
cfg = ITGConfig()
dev = torch.device("cuda:0")
ids, pix = synth.workload_inputs(cfg, rec["frames"], rec["resolution"], rec["prompt_len"], rec["pixel_seed"])
want = torch.tensor(rec["logits"])
eng = {dt: SelectorEngine(packed_random_weights(cfg, dt, dev, seed=0)) for dt in (torch.float16, torch.bfloat16)}
T = rec["frames"]
G = rec["resolution"] // cfg.vision.patch_size
hw = cfg.pooled_grid(T, G)
for vdt in (torch.float16, torch.bfloat16):
    feats, G_, cmap, pc = eng[vdt].vit_forward(pix.to(vdt).to(dev), hw)
    feats = feats.clone()
    for ldt in (torch.float16, torch.bfloat16):
        t0 = time.perf_counter()
        lg = eng[ldt].score_video(feats, T, G_, ids.view(-1), cmap, pc).float().cpu()
        torch.cuda.synchronize()
        ds = (torch.sigmoid(lg) - torch.sigmoid(want)).abs().max().item()
        top = sorted(torch.sort(torch.sigmoid(lg), descending=True, stable=True).indices[:rec["k"]].tolist()) == rec["topk"]
        print(f"tower {str(vdt)[6:]:9s} llm {str(ldt)[6:]:9s}: max|dscore| {ds:.2e}  max|dlogit| {(lg - want).abs().max().item():.2e}  top{rec['k']} set match {top}")
