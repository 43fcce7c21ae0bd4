#!/usr/bin/env python3
"""Round 5, VERDICT r4 item 6: can ANY e4m3 tier select the oracle's Top-32 on the headline workload, and would per-channel smoothing change it?

    python tools/fp8_error_floor.py gpu [case ...]      # on the MI355X: per-frame error anatomy of the named fp8 tier (decoder MLP on e4m3)
    python tools/fp8_error_floor.py channels            # CPU (oracle): per-channel dynamic range of the operands smoothing would rebalance

`gpu`: the 16-bit engine and the fp8_llm_mlp engine on the same fixture; per-frame score / logit error against the complete fp32 oracle pass: max, rms,
the error of the frames around the Top-K boundary, which frames swapped, and the expected number of boundary swaps for an error of that rms given how
densely the oracle's scores are packed around rank K (a tier keeps the SET only if its error at the boundary frames is below half the local score gap).
`channels`: what SmoothQuant-style scale migration (activation channel j / s_j into the RMSNorm gain, weight column j * s_j) has to work with: the
spread of per-channel |max| of the gate/up operand (post-attention RMSNorm output) and of the down operand (silu(gate) * up) over a small complete
pass of the full-width model.  s_j = sqrt(amax_x_j / amax_w_j) is the identity when every channel has the same range.
"""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videoitg_amd import synth  # noqa: E402
from videoitg_amd.config import ITGConfig  # noqa: E402


def gpu(cases):
    from videoitg_amd.engine import SelectorEngine
    from videoitg_amd.synth_device import packed_random_weights
    cfg, dev = ITGConfig(), torch.device("cuda:0")
    for case in cases:
        rec = json.load(open(os.path.join(ROOT, "tests", "golden", f"full_{case}.json")))
        ids, pix = synth.fixture_inputs(cfg, rec)
        pix = pix.half().to(dev)
        want = torch.tensor(rec["logits"])
        ws = torch.sigmoid(want)
        k, oord = rec["k"], rec["order"]
        srt = ws[oord]
        print(f"\n## {case}: oracle rank-{k}/{k + 1} score gap {rec['gap_k_k1']}; score gaps around the boundary (ranks {k - 4}..{k + 4}): "
              + ", ".join(f"{(srt[i] - srt[i + 1]).item():.1e}" for i in range(k - 5, k + 4)))
        for name, fp8 in (("16-bit", False), ("fp8_llm_mlp (decoder gate/up + down on e4m3, MX hand-over)", "llm.gu,llm.down")):
            eng = SelectorEngine(packed_random_weights(cfg, torch.float16, dev, seed=rec["weight_seed"], fp8=fp8, plan=synth.fixture_plan(rec)))
            lg = eng.forward(ids, None, [pix])[0].view(-1).float().cpu()
            sc = torch.sigmoid(lg)
            de, dl = sc - ws, lg - want
            mine = torch.sort(sc, descending=True, stable=True).indices.tolist()
            kept = set(mine[:k]) & set(oord[:k])
            lost, gained = sorted(set(oord[:k]) - kept), sorted(set(mine[:k]) - set(oord[:k]))
            near = oord[k - 4:k + 4]
            sigma = de.pow(2).mean().sqrt().item()
            # expected boundary swaps for independent per-frame errors of this rms: pairs (a in Top-k, b outside) whose oracle gap is below the
            # error of their difference (sigma * sqrt(2)) swap with probability Phi(-gap / (sigma sqrt 2))
            exp_swaps = 0.0
            for a in oord[:k]:
                for b in oord[k:k + 64]:
                    g = (ws[a] - ws[b]).item()
                    exp_swaps += 0.5 * math.erfc(g / (2.0 * sigma + 1e-30))
            print(f"| {name} | max dscore {de.abs().max().item():.2e} | rms dscore {sigma:.2e} | max dlogit {dl.abs().max().item():.2e} | rms dlogit "
                  f"{dl.pow(2).mean().sqrt().item():.2e} | Top-{k} kept {len(kept)}/{k} (lost {lost}, gained {gained}) | dscore of the 8 boundary frames "
                  f"{[round(de[f].item(), 5) for f in near]} | expected boundary pair swaps at this rms {exp_swaps:.2f} |", flush=True)
            eng.release_workspace()
            del eng
            torch.cuda.empty_cache()


def channels():
    from oracle import videoitg_oracle as O
    cfg = ITGConfig()
    v = cfg.vision
    torch.set_num_threads(os.cpu_count() or 1)
    T, R, L = 4, 224, 60
    ids, pix = synth.workload_inputs(cfg, T, R, L)
    vspecs, lspecs = list(synth.vision_param_specs(cfg)), list(synth.llm_param_specs(cfg))
    n_vit = cfg.num_vit_layers_needed()
    with torch.inference_mode():
        vsd = synth.counter_state_dict([s for s in vspecs if s[0].startswith("vision_model.embeddings.") or int(s[0].split(".")[3]) < n_vit], 0)
        feats = O.siglip_tower(pix.float(), vsd, patch=v.patch_size, num_heads=v.num_attention_heads, num_layers=v.num_hidden_layers,
                               select_layer=cfg.mm_vision_select_layer, eps=v.layer_norm_eps, frame_chunk=4)
        del vsd
        sd = synth.counter_state_dict([s for s in lspecs if not s[0].startswith("model.layers.") and s[0] != "model.embed_tokens.weight"], 0)
        visual = O.mlp_projector(feats, sd, cfg.vision_token_num)
        text = ids.reshape(-1)[ids.reshape(-1) != O.IMAGE_TOKEN_INDEX]
        espec = next(s for s in lspecs if s[0] == "model.embed_tokens.weight")
        small = synth.counter_rows(0, espec[0], espec[1], text, espec[2], espec[3])
        remap = ids.reshape(-1).clone()
        remap[remap != O.IMAGE_TOKEN_INDEX] = torch.arange(text.numel())
        x, _, _ = O.splice(remap.unsqueeze(0), None, [visual], {"model.embed_tokens.weight": small}, max_length=None, padding_side="right")
        cos, sin = O.rope_tables(x.shape[1], cfg.head_dim, cfg.rope_theta)
        print("| decoder layer | operand | per-channel amax: min / median / max | max / median | SmoothQuant s_j = sqrt(amax_x / amax_w): min / max |")
        print("|---|---|---|---|---|")
        for i in range(4):
            sdi = synth.counter_state_dict([s for s in lspecs if s[0].startswith(f"model.layers.{i}.")], 0)
            p = f"model.layers.{i}."
            x_attn = O.qwen2_layer(x, sdi, i, num_heads=cfg.num_attention_heads, num_kv_heads=cfg.num_key_value_heads, eps=cfg.rms_norm_eps,
                                   cos=cos, sin=sin, key_mask=None)
            # the MLP's operands, restated from the layer's own pieces: h = RMSNorm(x + attn) needs the mid-layer stream; the oracle's layer function
            # returns only its end, so the gate/up operand is taken from the layer INPUT normed by the post-attention gain (same statistics: one more
            # residual add of the same scale), and f from it
            h = O.rms_norm(x, sdi[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)[0]
            gate, up = h @ sdi[p + "mlp.gate_proj.weight"].t(), h @ sdi[p + "mlp.up_proj.weight"].t()
            f = torch.nn.functional.silu(gate) * up
            for name, act, w in (("gate/up operand (RMSNorm output)", h, torch.cat([sdi[p + "mlp.gate_proj.weight"], sdi[p + "mlp.up_proj.weight"]])),
                                 ("down operand (silu(gate) * up)", f, sdi[p + "mlp.down_proj.weight"])):
                ax, aw = act.abs().amax(0), w.abs().amax(0)
                s = (ax / aw).sqrt()
                s = s / s.median()
                print(f"| {i} | {name} | {ax.min().item():.2f} / {ax.median().item():.2f} / {ax.max().item():.2f} | {(ax.max() / ax.median()).item():.2f} | "
                      f"{s.min().item():.2f} / {s.max().item():.2f} (relative to the median) |", flush=True)
            x = x_attn


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "gpu"
    if mode == "channels":
        channels()
    else:
        gpu(sys.argv[2:] or ["c2_512x448", "c2p_512x448"])
