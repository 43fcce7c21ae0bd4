#!/usr/bin/env python3
"""Complete fp32 oracle passes over the FULL architecture (26 SigLIP-so400m layers + 28 Qwen2-7B layers) at BASELINE sizes.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fullsize_golden.py CASE [CASE ...]

Runs in the build container (CPU only, 8 vCPU / 62 GB; minutes to half an hour per case) and writes
``tests/golden/full_<case>.json``: seeds, the T oracle logits, sigmoid scores, the descending order, the Top-32 set, the
rank-32/33 score gap, a few intermediate checksums and the oracle's wall time.  Weights are
``videoitg_amd.synth.counter_tensor`` (a stateless integer hash: bit-identical on CPU and GPU), inputs come from
``oracle.fullsize.workload_inputs`` — so the `-m gpu` tests and ``bench.py`` rebuild exactly the same model and inputs on the
MI355X and compare the HIP path with these logits (max |dscore|, Top-32 set), with no oracle run and no reference on the box.

The oracle is pinned to the reference itself at small sizes (make_golden.py -> test_oracle_golden.py, <= 2e-5); this script
extends the same oracle functions, unchanged, to full width and depth (oracle/fullsize.py only streams the weights).
"""
import json
import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import fullsize  # noqa: E402
from videoitg_amd import synth  # noqa: E402
from videoitg_amd.config import ITGConfig  # noqa: E402

# name -> (frames, resolution, prompt tokens).  c2 = BASELINE.json configs[1] (the bench workload), b0 = the reference-native
# 512 x 384^2, c1 = configs[0]'s size (64 frames, N = 16 445), depth16 = full depth/width on few small frames (fast GPU test)
CASES = {
    "c2_512x448": (512, 448, 60),
    "b0_512x384": (512, 384, 60),
    "c1_64x384": (64, 384, 60),
    "depth16_224": (16, 224, 60),
    # robustness: other weights (seed 1), other pixels, a 23-token prompt - the dtype decision must not hinge on one draw
    "depth16_224_s1": (16, 224, 23, 1, 777),
    "depth24_280_s2": (24, 280, 41, 2, 4242),
    # planted per-frame signal (videoitg_amd.synth.planted_inputs: 32 bright frames among dark ones): a case whose Top-32 set is
    # DECIDABLE for the lower-precision tiers (SURVEY.md section 7 hard part 2 iii).  c2p = the bench workload's size
    "c2p_512x448": (512, 448, 60, 0, 1234, 32),
    "depth16p_224": (16, 224, 60, 0, 1234, 4),
    # round 6: planted twins of the reference-native shape and of the decoder-heavy 64-frame video
    "b0p_512x384": (512, 384, 60, 0, 1234, 32),
    "c1p_64x384": (64, 384, 60, 0, 1234, 32),
    # BASELINE.json configs[0]'s own shape (64 uniformly sampled 224^2 frames: 16 x 16 tower grid = the pooled grid, i.e. the projector's NO-POOL branch,
    # mlp_proj.py:48-56 with h == hw) at full depth and width, and its planted twin
    "c0_64x224": (64, 224, 60),
    "c0p_64x224": (64, 224, 60, 0, 1234, 32),
    # more than 512 frames at full depth and width: the projector's chunk branch (mlp_proj.py: T > 512) with a 5 x 5 pooled grid, N = 13 060
    "t520_224": (520, 224, 60),
}
# heavy-tailed weights (videoitg_amd.synth.HeavyTailPlan: outlier channels planted through fc2 / down rows, biases and norm gains the
# way trained towers carry them).  Same sizes, inputs and base draw as the case without the "o".
HEAVY = {"c2o_512x448": "c2_512x448", "depth16o_224": "depth16_224"}
for _k, _v in HEAVY.items():
    CASES[_k] = CASES[_v]
PLANT_SEED = 99


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    for name in sys.argv[1:]:
        T, R, L, wseed, pseed = (CASES[name] + (0, 1234))[:5]
        plant_k = CASES[name][5] if len(CASES[name]) > 5 else 0
        cfg = ITGConfig()
        plan = (synth.HeavyTailPlan.from_json(json.load(open(os.path.join(HERE, "heavy_tail_plan.json")))["plan"])
                if name in HEAVY else None)
        planted = None
        if plant_k:
            ids, pix, planted = synth.planted_inputs(cfg, T, R, L, pseed, plant_k, PLANT_SEED)
        else:
            ids, pix = fullsize.workload_inputs(cfg, T, R, L, pseed)
        t0 = time.time()
        with torch.inference_mode():
            out = fullsize.streaming_forward(cfg, ids, pix, seed=wseed, plan=plan, log=lambda m: print(f"[{name} +{time.time() - t0:.0f}s] {m}", flush=True))
        rk = fullsize.ranking_summary(out["logits"], plant_k or 32)
        rec = dict(case=name, frames=T, resolution=R, prompt_len=L, weight_seed=wseed, pixel_seed=pseed, prompt_seed=4321,
                   llm_tokens=out["N"], **({"plant": dict(k=plant_k, seed=PLANT_SEED, frames=planted, what="planted frames pixels*0.5+0.5, others pixels*0.5-0.5")}
                                           if plant_k else {}), **({"heavy_tail": plan.to_json()} if plan else {}), logits=[float(x) for x in out["logits"].tolist()], **rk, checks=out["checks"],
                   oracle=dict(kind="oracle/videoitg_oracle.py streamed by oracle/fullsize.py, torch CPU fp32",
                               torch=torch.__version__, threads=torch.get_num_threads(), seconds=out["seconds"]))
        path = os.path.join(HERE, f"full_{name}.json")
        with open(path, "w") as f:
            json.dump(rec, f)
        print(f"wrote {path}: logits[:4]={rec['logits'][:4]} gap32/33={rec['gap_k_k1']} seconds={out['seconds']}", flush=True)


if __name__ == "__main__":
    main()
