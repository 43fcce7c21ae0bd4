#!/usr/bin/env python3
"""Known answers of the reference's integer helpers on a SWEEP of seeded random arguments (build container only: runs the reference's own functions from
/root/reference; writes tests/golden/helpers_sweep.json - data only, outputs stored as length + first / last elements + a SHA-1 of the index list).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_helpers_sweep.py

Rows A1 / A1' / A4 / A5 of SURVEY.md section 8 are integer work: the bar is exact equality, and ten hand-picked cases (helpers.json) leave the corners to
chance - frame rates just below / above x.5 (Python's banker's rounding of the stride), clips shorter than one stride, more samples than frames, the
`multiple` padding of get_seq_frames, prompts with several <image> tokens with and without a BOS."""
import ast
import hashlib
import importlib
import json
import os
import random
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                                     # noqa: E402  (the Appendix-B shims)

REF = MG.REF


def digest(xs):
    xs = [int(x) for x in xs]
    return {"n": len(xs), "head": xs[:3], "tail": xs[-3:], "sha1": hashlib.sha1(",".join(map(str, xs)).encode()).hexdigest()}


def outcome(fn, *args):
    try:
        return digest(fn(*args))
    except Exception as e:                                                   # the failure mode is part of the behaviour
        return {"error": type(e).__name__}


class Tok:
    def __init__(self, bos=None):
        self.bos_token_id = bos

    def __call__(self, s):
        ids = [ord(c) for c in s]
        if self.bos_token_id is not None:
            ids = [self.bos_token_id] + ids
        return types.SimpleNamespace(input_ids=ids)


def main():
    MG.install_shims()
    mm = importlib.import_module("eagle.mm_utils")
    src = open(f"{REF}/lmms_eval/models/videoitg.py").read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "get_seq_frames")
    ns = {"np": np}
    exec(compile(ast.Module([fn], []), "<ref:get_seq_frames>", "exec"), ns)
    # infer.py's own pad_sequence (infer.py:26-32) is a local function of the script: take its definition the same way
    isrc = open(f"{REF}/infer.py").read()
    pfn = next((n for n in ast.walk(ast.parse(isrc)) if isinstance(n, ast.FunctionDef) and n.name == "pad_sequence"), None)
    pns = {"torch": torch}
    if pfn is not None:
        exec(compile(ast.Module([pfn], []), "<ref:pad_sequence>", "exec"), pns)
    rnd = random.Random(20260930)
    fps_pool = [1.0, 5.0, 10.0, 12.5, 14.985, 15.0, 23.976, 24.0, 25.0, 29.97, 30.0, 48.0, 50.0, 59.94, 60.0, 119.88, 120.0, 2.5, 3.0, 4.99, 5.01, 0.9, 0.4]
    out = {"get_frame_indices": [], "get_seq_frames": [], "tokenizer_image_token": [], "pad_sequence": []}
    for _ in range(400):
        total = rnd.choice([rnd.randint(1, 40), rnd.randint(1, 2000), rnd.randint(1, 200000)])
        fps = rnd.choice(fps_pool + [round(rnd.uniform(0.3, 130.0), 3)])
        tfps = rnd.choice([1, 2, 2, 0.5, 4, 3])
        nfrm = rnd.choice([1, 8, 16, 32, 64, 512, 512, rnd.randint(1, 600)])
        out["get_frame_indices"].append({"args": [total, fps, tfps, nfrm], "out": outcome(mm.get_frame_indices, total, fps, tfps, nfrm)})
        mult = rnd.choice([1, 1, 2, 8, 16])
        out["get_seq_frames"].append({"args": [total, fps, tfps, nfrm, mult], "out": outcome(ns["get_seq_frames"], total, fps, tfps, nfrm, mult)})
    alphabet = "ab \n?<>image"
    for _ in range(200):
        parts = ["".join(rnd.choice(alphabet) for _ in range(rnd.randint(0, 7))) for _ in range(rnd.randint(1, 4))]
        prompt = "<image>".join(parts)
        bos = rnd.choice([None, None, 1, 151643])
        out["tokenizer_image_token"].append({"prompt": prompt, "bos": bos, "out": outcome(mm.tokenizer_image_token, prompt, Tok(bos), -200)})
    if "pad_sequence" in pns:
        for _ in range(60):
            lens = [rnd.randint(1, 9) for _ in range(rnd.randint(1, 4))]
            seqs = [[rnd.randint(-200, 300) for _ in range(n)] for n in lens]
            side = rnd.choice(["left", "right"])
            pad = rnd.choice([0, -100, 151643])
            tok = types.SimpleNamespace(padding_side=side)
            got = pns["pad_sequence"](tok, [torch.tensor(s) for s in seqs], True, pad)
            out["pad_sequence"].append({"seqs": seqs, "side": side, "pad": pad, "out": got.tolist()})
    out["what"] = {"generator": "tests/golden/make_helpers_sweep.py", "reference": ["eagle/mm_utils.py: get_frame_indices, tokenizer_image_token",
                   "lmms_eval/models/videoitg.py: get_seq_frames", "infer.py: pad_sequence"], "seed": 20260930}
    with open(os.path.join(HERE, "helpers_sweep.json"), "w") as f:
        json.dump(out, f)
    errs = {k: sum(1 for e in v if isinstance(e, dict) and isinstance(e.get("out"), dict) and "error" in e["out"]) for k, v in out.items() if isinstance(v, list)}
    print({k: len(v) for k, v in out.items() if isinstance(v, list)}, "errors recorded:", errs)


if __name__ == "__main__":
    main()
