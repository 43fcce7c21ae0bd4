#!/usr/bin/env python3
"""CPU (build container): WHICH rounding sites must carry split (hi + lo fp16) operands for the heavy-tailed fixture to come under 1e-3?

oracle/quant_model.py "product" rounds where libvideoitg_hip.so rounds; `exact=` lifts named sites to split precision (modelled as exact: hi + lo keeps
22 of fp32's 24 bits).  Sites: norm1 norm2 (norm outputs = A operands of qkv / fc1 / gate-up), qkv (fused-QKV output and the RoPE output = attention's
Q / K / V), p (softmax probabilities = PV's operand), attn_out (A operand of the out projection), act (GELU / SwiGLU output = A operand of fc2 / down),
pool / proj (projector operands).  VERDICT r5 item 2: find the SMALLEST set before building kernels for it.

    PYTHONDONTWRITEBYTECODE=1 python tools/split_site_model.py depth16o_224 [variant ...]        (~8 min per variant on 8 vCPUs)"""
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import quant_model as QM            # noqa: E402
from videoitg_amd import synth                  # noqa: E402
from videoitg_amd.config import ITGConfig       # noqa: E402

GEMM_A = ("norm1", "norm2", "attn_out", "act", "pool", "proj")
VARIANTS = {
    "none": (),
    "mlp": ("norm2", "act"),
    "gemm_a": GEMM_A,
    "gemm_a+qkv": GEMM_A + ("qkv",),
    "attn_internal": ("qkv", "p"),
    "all_but_p": GEMM_A + ("qkv",),
    "all": GEMM_A + ("qkv", "p"),
}


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = ITGConfig()
    case = sys.argv[1] if len(sys.argv) > 1 else "depth16o_224"
    names = sys.argv[2:] or ["none", "mlp", "gemm_a", "gemm_a+qkv", "all"]
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", f"full_{case}.json")))
    plan = synth.fixture_plan(rec)
    ids, pix = synth.fixture_inputs(cfg, rec)
    n_vit = cfg.num_vit_layers_needed()
    vsd = synth.counter_state_dict([s for s in synth.vision_param_specs(cfg) if s[0].startswith("vision_model.embeddings.") or int(s[0].split(".")[3]) < n_vit],
                                   rec["weight_seed"], plan=plan)
    lspecs = list(synth.llm_param_specs(cfg))
    sd = synth.counter_state_dict([s for s in lspecs if s[0] != "model.embed_tokens.weight"], rec["weight_seed"], plan=plan)
    flat = ids.reshape(-1)
    text = flat[flat != -200]
    espec = next(s for s in lspecs if s[0] == "model.embed_tokens.weight")
    small = synth.counter_rows(rec["weight_seed"], espec[0], espec[1], text, espec[2], espec[3])
    if plan is not None:
        small = plan.apply({"model.embed_tokens.weight": small})["model.embed_tokens.weight"]
    sd["model.embed_tokens.weight"] = small
    remap = flat.clone()
    remap[flat != -200] = torch.arange(text.numel())
    want = torch.tensor(rec["logits"])
    print(f"| {case}: split sites | max abs dlogit | rms dlogit | max abs dscore |\n|---|---|---|---|", flush=True)
    for name in names:
        with torch.inference_mode():
            got = QM.forward(cfg, sd, vsd, remap.unsqueeze(0), None, [pix], dtype=torch.float16, mode="product", frame_chunk=16,
                             exact=VARIANTS[name])[0].view(-1).float()
        dl = got - want
        print(f"| {name} ({', '.join(VARIANTS[name]) or '-'}) | {dl.abs().max().item():.2e} | {dl.pow(2).mean().sqrt().item():.2e} | "
              f"{(torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item():.2e} |", flush=True)


if __name__ == "__main__":
    main()
