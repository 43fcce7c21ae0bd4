#!/usr/bin/env python3
"""CPU (build container): WHICH rounding sites must carry split (hi + lo fp16) operands for the heavy-tailed fixture to come under 1e-3?

oracle/quant_model.py "product" rounds where libvideoitg_hip.so rounds; `exact=` lifts named sites to split precision (modelled as exact: hi + lo keeps
22 of fp32's 24 bits).  Sites: norm1 norm2 (norm outputs = A operands of qkv / fc1 / gate-up), qkv (fused-QKV output and the RoPE output = attention's
Q / K / V), p (softmax probabilities = PV's operand), attn_out (A operand of the out projection), act (GELU / SwiGLU output = A operand of fc2 / down),
pool / proj (projector operands).  VERDICT r5 item 2: find the SMALLEST set before building kernels for it.

    PYTHONDONTWRITEBYTECODE=1 python tools/split_site_model.py depth16o_224 [variant ...]        (~25 min per variant on 8 vCPUs)
    python tools/split_site_model.py --device cuda c2o_512x448 [variant ...]                      (GPU box: torch fp32 on the device, ~1 min per variant)

With --device cuda the same torch-fp32 model runs on the MI355X through torch's own fp32 GEMMs (analysis only - this is the rounding MODEL, not the
product path); the variant "all" (nothing rounded) then measures how far torch-fp32-on-GPU is from the committed CPU oracle logits."""
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


class _F:
    """torch.nn.functional with the patch-embedding convolution (stride = kernel) restated as im2col + matmul: no MIOpen on the GPU box"""

    def __getattr__(self, name):
        return getattr(torch.nn.functional, name)

    @staticmethod
    def conv2d(x, w, b, stride):
        T, C, R, _ = x.shape
        p = stride
        G = R // p
        cols = x[:, :, :G * p, :G * p].reshape(T, C, G, p, G, p).permute(0, 2, 4, 1, 3, 5).reshape(T * G * G, C * p * p)
        y = cols @ w.reshape(w.shape[0], -1).t() + b
        return y.reshape(T, G, G, -1).permute(0, 3, 1, 2)


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = ITGConfig()
    argv = sys.argv[1:]
    dev = "cpu"
    if argv[:1] == ["--device"]:
        dev, argv = argv[1], argv[2:]
        QM.F = _F()
    case = argv[0] if argv else "depth16o_224"
    names = argv[1:] or ["none", "mlp", "gemm_a", "gemm_a+qkv", "all"]
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", f"full_{case}.json")))
    plan = synth.fixture_plan(rec)
    ids, pix = synth.fixture_inputs(cfg, rec)
    n_vit = cfg.num_vit_layers_needed()
    vsd = synth.counter_state_dict([s for s in synth.vision_param_specs(cfg) if s[0].startswith("vision_model.embeddings.") or int(s[0].split(".")[3]) < n_vit],
                                   rec["weight_seed"], device=dev, plan=plan)
    lspecs = list(synth.llm_param_specs(cfg))
    sd = synth.counter_state_dict([s for s in lspecs if s[0] != "model.embed_tokens.weight"], rec["weight_seed"], device=dev, plan=plan)
    flat = ids.reshape(-1)
    text = flat[flat != -200]
    espec = next(s for s in lspecs if s[0] == "model.embed_tokens.weight")
    small = synth.counter_rows(rec["weight_seed"], espec[0], espec[1], text, espec[2], espec[3])
    if plan is not None:
        small = plan.apply({"model.embed_tokens.weight": small})["model.embed_tokens.weight"]
    sd["model.embed_tokens.weight"] = small.to(dev)
    remap = flat.clone()
    remap[flat != -200] = torch.arange(text.numel())
    remap, pix = remap.to(dev), pix.to(dev)
    want = torch.tensor(rec["logits"])
    print(f"| {case}: split sites | max abs dlogit | rms dlogit | max abs dscore |\n|---|---|---|---|", flush=True)
    for name in names:
        with torch.inference_mode(), torch.device(dev):
            got = QM.forward(cfg, sd, vsd, remap.unsqueeze(0), None, [pix], dtype=torch.float16, mode="product", frame_chunk=16,
                             exact=VARIANTS[name])[0].view(-1).float().cpu()
        dl = got - want
        print(f"| {name} ({', '.join(VARIANTS[name]) or '-'}) | {dl.abs().max().item():.2e} | {dl.pow(2).mean().sqrt().item():.2e} | "
              f"{(torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item():.2e} |", flush=True)


if __name__ == "__main__":
    main()
