#!/usr/bin/env python3
"""Run THE REFERENCE (EagleQwenG through the Appendix-B shims of make_golden.py) on the seeded random batch structures of tests/structures.py and record
its logits per prompt: tests/golden/random_structures.npz; and on the seeded random ARCHITECTURES (draw_arch): random_archs.npz (data only).  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_random_structures.py
"""
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as MG                                                     # noqa: E402
import structures as ST                                                      # noqa: E402


def main():
    g, ce = MG.install_shims()
    rec = {"n": np.array(ST.N_STRUCTURES)}
    for seed in range(ST.N_STRUCTURES):
        s = ST.draw(seed)
        cfg = ST.config(s)
        torch.manual_seed(0)
        m = MG.build_reference_model(g, ce, cfg)
        MG.load_synth(m, cfg, seed)
        imgs = ST.images(s)
        with torch.inference_mode():
            out = m(s["ids"], attention_mask=s["mask"], images=imgs)
        assert len(out.logits) == s["B"]
        rec[f"s{seed}_ids"] = s["ids"].numpy()
        rec[f"s{seed}_mask"] = s["mask"].numpy()
        rec[f"s{seed}_T"] = np.array(s["T"])
        rec[f"s{seed}_meta"] = np.array([s["side"], s["budget"], int(s["two_first"])])
        for b, lg in enumerate(out.logits):
            rec[f"s{seed}_logits{b}"] = lg.float().numpy()
        print(s["what"], "->", [tuple(l.shape) for l in out.logits], float(max(l.abs().max() for l in out.logits)), flush=True)
    np.savez_compressed(os.path.join(HERE, "random_structures.npz"), **rec)
    print("random_structures.npz written")
    # ---- random architectures (tests/structures.py: draw_arch) ----
    rec = {"n": np.array(ST.N_ARCHS)}
    for seed in range(ST.N_ARCHS):
        s = ST.draw_arch(seed)
        cfg = ST.arch_config(s)
        torch.manual_seed(0)
        m = MG.build_reference_model(g, ce, cfg)
        MG.load_synth(m, cfg, seed)
        with torch.inference_mode():
            out = m(s["ids"], attention_mask=s["mask"], images=ST.arch_images(s))
        rec[f"a{seed}_ids"] = s["ids"].numpy()
        rec[f"a{seed}_T"] = np.array(s["T"])
        rec[f"a{seed}_arch"] = np.array(json.dumps(s["arch"], sort_keys=True))
        rec[f"a{seed}_logits0"] = out.logits[0].float().numpy()
        print(s["what"], "->", tuple(out.logits[0].shape), float(out.logits[0].abs().max()), flush=True)
    np.savez_compressed(os.path.join(HERE, "random_archs.npz"), **rec)
    print("random_archs.npz written")


if __name__ == "__main__":
    main()
