#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports /root/reference's ``EagleQwenG`` through the harness-side shims of SURVEY.md Appendix B
(nothing under /root/reference is modified), loads seeded synthetic weights from
``videoitg_amd.synth`` into it, runs the forward on seeded inputs and writes, per case, a small ``.npz``
with the inputs that are not re-derivable from the seed, the logits and intermediate check values.
It also records known answers of the reference's pure-Python helpers (``eagle/mm_utils.py``,
``lmms_eval/models/videoitg.py:get_seq_frames``) in ``helpers.json``.

Fixtures are data only (inputs / expected outputs); no reference source is copied.
"""
import importlib
import json
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from videoitg_amd.config import ITGConfig, VisionConfig, tiny_config  # noqa: E402
from videoitg_amd import synth  # noqa: E402


# ------------------------------------------------------------------ shims
def install_shims():
    sys.path.insert(0, REF)

    def stub(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m
        return m

    stub("decord", VideoReader=object, cpu=lambda *a, **k: None)
    stub("av")
    stub("av.codec")
    stub("av.codec.context", CodecContext=object)
    for n, p in (("eagle", "eagle"), ("eagle.model", "eagle/model")):
        pkg = types.ModuleType(n)
        pkg.__path__ = [f"{REF}/{p}"]
        sys.modules[n] = pkg
    stub("eagle.model.multimodal_encoder.multi_backbone_channel_concatenation_encoder",
         MultiBackboneChannelConcatenationVisionTower=object)
    import transformers.masking_utils as mu
    import transformers.models.qwen2.modeling_qwen2 as mq
    mq.Qwen2FlashAttention2 = mq.Qwen2Attention                 # class removed after 4.47
    mq.create_causal_mask = mu.create_bidirectional_mask        # reference attention is bidirectional
    g = importlib.import_module("eagle.model.language_model.grounding_qwen2")
    ce = importlib.import_module("eagle.model.multimodal_encoder.clip_encoder")
    return g, ce


def build_reference_model(g, ce, cfg: ITGConfig):
    from transformers import SiglipImageProcessor, SiglipVisionConfig, SiglipVisionModel
    v = cfg.vision
    vcfg = SiglipVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
                              num_hidden_layers=v.num_hidden_layers, num_attention_heads=v.num_attention_heads,
                              image_size=v.image_size, patch_size=v.patch_size, layer_norm_eps=v.layer_norm_eps)
    vcfg._attn_implementation = "eager"
    ce.SiglipVisionConfig = type("C", (), {"from_pretrained": staticmethod(lambda n, **k: vcfg)})
    ce.SiglipVisionModel = type("M", (), {"from_pretrained": staticmethod(lambda n, **k: SiglipVisionModel(vcfg))})
    ce.SiglipImageProcessor = type("P", (), {"from_pretrained": staticmethod(
        lambda n, **k: SiglipImageProcessor(size={"height": v.image_size, "width": v.image_size}))})

    class Cfg(g.EagleQwenGConfig):
        def __setattr__(self, k, val):
            if k == "rope_scaling" and val is None:      # grounding_qwen2.py:75 would wipe rope params in 5.x
                return
            super().__setattr__(k, val)

    rcfg = Cfg(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
               num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
               num_key_value_heads=cfg.num_key_value_heads, max_position_embeddings=cfg.max_position_embeddings,
               rope_theta=cfg.rope_theta, rms_norm_eps=cfg.rms_norm_eps,
               mm_vision_tower=cfg.mm_vision_tower, mm_projector_type=cfg.mm_projector_type,
               mm_use_4_vision_tokens=cfg.mm_use_4_vision_tokens, vision_token_num=cfg.vision_token_num,
               vision_min_num=cfg.vision_min_num, mm_vision_select_layer=cfg.mm_vision_select_layer,
               mm_vision_select_feature=cfg.mm_vision_select_feature, mm_hidden_size=cfg.mm_hidden_size,
               tokenizer_model_max_length=cfg.tokenizer_model_max_length,
               tokenizer_padding_side=cfg.tokenizer_padding_side)
    rcfg._attn_implementation = "eager"
    m = g.EagleQwenG(rcfg)
    m.get_vision_tower().load_model()
    m.eval()
    from transformers.utils.output_capturing import maybe_install_capturing_hooks
    maybe_install_capturing_hooks(m.model)               # once, top-down, before any forward (App. B shim 6)
    assert all(len(l._forward_hooks) == 1 for l in m.get_vision_tower().vision_tower.encoder.layers)
    return m


def tiny_heavy_plan(cfg: ITGConfig) -> "synth.HeavyTailPlan":
    """the massive-activation anatomy of synth.HeavyTailPlan on the toy architecture (round 6): 4 planted channels in each residual stream, a constant
    from a bias in the first tower layers / the projector's last bias + the embedding column, 64x rows in every fc2 / down_proj, norm gains of (4, 1/4)
    and (4, 1/8).  Power-of-two factors and 16-bit-exact constants only: the weights stay 16-bit exact."""
    vc, lc = synth.HeavyTailPlan.channels(cfg)
    n_vit = cfg.num_vit_layers_needed()
    return synth.HeavyTailPlan(vc, lc, vit_norm=[(4.0, 0.25)] * n_vit, llm_norm=[(4.0, 0.125)] * cfg.num_hidden_layers, final_norm=(4.0, 0.125),
                               row_gain=64.0, vit_bias=128.0, vit_bias_layers=(0, 1), llm_bias=1024.0)


def load_synth(m, cfg: ITGConfig, seed: int, plan=None):
    sd = synth.synth_llm_state_dict(cfg, seed)
    vsd = synth.synth_vision_state_dict(cfg, seed)
    if plan is not None:
        sd, vsd = plan.apply(sd), plan.apply(vsd)
    own = m.state_dict()
    mapped = {}
    for k, t in sd.items():
        assert k in own, k
        mapped[k] = t
    vkeys = [k for k in own if "vision_tower" in k]
    for k, t in vsd.items():
        tail = k[len("vision_model."):]
        cand = [o for o in vkeys if o.endswith(tail)]
        assert len(cand) == 1, (k, cand)
        mapped[cand[0]] = t
    missing = [k for k in own if k not in mapped]
    # tower tensors the path never uses (last layer is still used by HF forward but not by hidden_states[-2])
    unused = [k for k in missing if ("post_layernorm" in k or ".head." in k)]
    assert set(missing) == set(unused), sorted(set(missing) - set(unused))
    m.load_state_dict(mapped, strict=False)
    return sd, vsd


CASES = {
    # name: (cfg overrides, frames per video, prompt layout)
    # pool 4x4 -> 3x3, prompt after <image>
    "pool_post": dict(cfg=dict(vision_token_num=72), T=[8], ids=[[-200, 5, 6, 7, 8, 9]]),
    # no pooling (HW == G), text before and after <image>
    "nopool_prepost": dict(cfg=dict(vision_token_num=4096), T=[6], ids=[[11, 12, -200, 5, 6, 7]]),
    # odd grid 5x5 -> 3x3, longer prompt
    "oddgrid": dict(cfg=dict(vision_token_num=60, vision=dict(image_size=70)), T=[6],
                    ids=[[-200] + list(range(20, 45))]),
    # batch of two videos of different length, right padding (token type 4)
    "batch2_pad": dict(cfg=dict(vision_token_num=72), T=[8, 5],
                       ids=[[-200, 5, 6, 7, 8, 9], [-200, 30, 31, 0, 0, 0]], pad=0),
    # T > 512 -> chunked ViT (384) and chunked pool (512) branches; 2x2 grid pooled to 1x1
    "chunk520": dict(cfg=dict(vision_token_num=600, vision=dict(image_size=28)), T=[520], ids=[[-200, 5, 6, 7]]),
    # image side not a multiple of the patch (62 = 4*14 + 6, like SigLIP-so400m's 384 = 27*14 + 6): the stride-14 VALID
    # convolution ignores the last 6 rows / columns
    "ragged_image": dict(cfg=dict(vision_token_num=72, vision=dict(image_size=62)), T=[7], ids=[[-200, 5, 6, 7, 8]]),
    # ONE sample with TWO <image> tokens (eagle_archv1.py:296-337): it consumes two entries of `images`, text sits between the
    # visual blocks, and the head (grounding_qwen2.py:153) reshapes ALL its visual rows by images[b].shape[0] = the FIRST clip's
    # frame count - so with two clips of 6 frames each logit pools two consecutive frames' tokens
    "two_images": dict(cfg=dict(vision_token_num=72), T=[6, 6], ids=[[11, -200, 5, 6, -200, 7, 8]], samples=1),
    # the same with clips of different length: each is pooled with ITS OWN grid (4 frames -> 4x4, 8 frames -> 3x3: 64 + 72 rows), the
    # head's groups of 136 / 4 = 34 rows straddle frames and clips
    "two_images_ragged": dict(cfg=dict(vision_token_num=72), T=[4, 8], ids=[[-200, 5, 6, -200, 7]], samples=1),
    # round 6: HEAVY-TAILED weights through the reference itself (tiny_heavy_plan: |x| ~ 10^2 - 10^3 in four channels of each residual stream), 24 frames,
    # a 20-token prompt - the anatomy of tests/golden/full_c2o_512x448.json at a size the reference runs here, so that the precision tiers are
    # compared with THE REFERENCE's fp32 forward on such weights and not only with the oracle's
    "heavy_pool_post": dict(cfg=dict(vision_token_num=216), T=[24], ids=[[-200] + list(range(40, 60))], heavy=True),
}
# degenerate inputs (recorded in helpers.json["forward_errors"]: exception type, or shape / all-NaN flag of the logits): a sample
# without <image> still consumes an entry of `images` (eagle_archv1.py:282-294) and then has no visual rows - the head's mean over
# an empty [T, 0, D] view gives NaN logits; visual rows that do not divide by images[b].shape[0] fail the head's reshape
ERROR_CASES = {
    "imageless": dict(cfg=dict(vision_token_num=72), T=[4], ids=[[5, 6, 7, 8]]),
    "two_images_bad_split": dict(cfg=dict(vision_token_num=72), T=[5, 8], ids=[[-200, 5, 6, -200, 7]]),    # 5*9 + 8*9 = 117 rows, 117 % 5 != 0
}


def run_case(g, ce, name, spec, seed=0):
    cfg = tiny_config(**spec["cfg"])
    torch.manual_seed(0)
    m = build_reference_model(g, ce, cfg)
    plan = tiny_heavy_plan(cfg) if spec.get("heavy") else None
    load_synth(m, cfg, seed, plan)
    R = cfg.vision.image_size
    images = [synth.synth_pixel_values(T, R, seed=1234 + i) for i, T in enumerate(spec["T"])]
    ids = torch.tensor(spec["ids"], dtype=torch.long)
    pad = spec.get("pad")
    mask = ids.ne(pad) if pad is not None else torch.ones_like(ids).bool()
    with torch.inference_mode():
        # tower + projector intermediates through the reference's own modules
        feats = m.get_vision_tower()(torch.cat(images, 0))
        vis = m.get_model().mm_projector(list(torch.split(feats, spec["T"], 0)))
        (_, _, am, _, emb, _, tts) = m.prepare_inputs_labels_for_multimodal(ids, None, mask, None, None, images, None)
        out = m(ids, attention_mask=mask, images=images)
        out2 = m(ids, attention_mask=mask, images=images)
    for a, b in zip(out.logits, out2.logits):
        assert torch.equal(a, b), "reference forward not repeatable (hook duplication?)"
    hs = out.hidden_states           # LLM hidden states per layer (output_hidden_states=True, grounding_qwen2.py:143)
    rec = dict(
        cfg=json.dumps(cfg.to_dict()), seed=seed, T=np.array(spec["T"]), input_ids=ids.numpy(),
        attention_mask=mask.numpy(),
        vit_first=feats[0].numpy(), vit_last=feats[-1].numpy(),
        vit_sum=np.array([feats.double().sum().item(), feats.double().abs().sum().item()]),
        visual0_head=vis[0][:32].numpy(),
        visual_sums=np.array([[x.double().sum().item(), x.double().abs().sum().item()] for x in vis]),
        visual_shapes=np.array([list(x.shape) for x in vis]),
        embeds_shape=np.array(list(emb.shape)), token_types=tts.numpy(),
        embeds_sum=np.array([emb.double().sum().item(), emb.double().abs().sum().item()]),
        layer_sums=np.array([[h.double().sum().item(), h.double().abs().sum().item()] for h in hs]),
        final_row0=hs[-1][0, :4].numpy() if hs is not None else np.zeros(1),
    )
    for b, lg in enumerate(out.logits):
        rec[f"logits{b}"] = lg.numpy()
    if plan is not None:
        rec["heavy_tail"] = json.dumps(plan.to_json())
        rec["stream_absmax"] = np.array([feats.abs().max().item(), max(h.abs().max().item() for h in hs[:-1])])
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
    print(name, "N=", emb.shape[1], "logits0[:4]=", out.logits[0].view(-1)[:4].tolist())


def train_mode_pool(g, ce):
    """in TRAINING mode the projector pools every video to a random grid (mlp_proj.py:51-52).  The reference's forward in
    .train() mode under a fixed torch seed -> the grids it drew (from the visual shapes) and its logits (train_pool.npz)."""
    spec = dict(cfg=dict(vision_token_num=72, vision_min_num=1), T=[8, 5], ids=[[-200, 5, 6, 7, 8, 9], [-200, 30, 31, 0, 0, 0]], pad=0, seed=20)
    cfg = tiny_config(**spec["cfg"])
    torch.manual_seed(0)
    m = build_reference_model(g, ce, cfg)
    load_synth(m, cfg, 0)
    m.train()
    R = cfg.vision.image_size
    images = [synth.synth_pixel_values(T, R, seed=1234 + i) for i, T in enumerate(spec["T"])]
    ids = torch.tensor(spec["ids"], dtype=torch.long)
    mask = ids.ne(spec["pad"])
    torch.manual_seed(spec["seed"])
    with torch.no_grad():
        out = m(ids, attention_mask=mask, images=images)
    n_vis = [int((tt == 3).sum()) for tt in out.hidden_states[0].new_zeros(1)] if False else None
    # the grids drawn: re-run the projector's draw sequence under the same seed (torch.randint(vision_min_num, HW + 1, (1,)) per video, in order)
    import math
    torch.manual_seed(spec["seed"])
    grids = [min(torch.randint(cfg.vision_min_num, math.floor((cfg.vision_token_num / T) ** 0.5) + 1, (1,)).item(), cfg.vision.grid) for T in spec["T"]]
    rec = dict(cfg=json.dumps(cfg.to_dict()), seed=0, torch_seed=spec["seed"], T=np.array(spec["T"]), input_ids=ids.numpy(), attention_mask=mask.numpy(),
               grids=np.array(grids), n_tokens=np.array(list(out.hidden_states[0].shape)))
    for b, lg in enumerate(out.logits):
        rec[f"logits{b}"] = lg.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "train_pool.npz"), **rec)
    print("train_pool grids", grids, "N=", out.hidden_states[0].shape[1], "logits0[:4]=", out.logits[0].view(-1)[:4].tolist())


def forward_errors(g, ce):
    out = {}
    for name, spec in ERROR_CASES.items():
        cfg = tiny_config(**spec["cfg"])
        torch.manual_seed(0)
        m = build_reference_model(g, ce, cfg)
        load_synth(m, cfg, 0)
        images = [synth.synth_pixel_values(T, cfg.vision.image_size, seed=1234 + i) for i, T in enumerate(spec["T"])]
        ids = torch.tensor(spec["ids"], dtype=torch.long)
        try:
            with torch.inference_mode():
                o = m(ids, attention_mask=torch.ones_like(ids).bool(), images=images)
            out[name] = {"T": spec["T"], "ids": spec["ids"], "cfg": spec["cfg"], "error": None,
                         "logits_shapes": [list(l.shape) for l in o.logits], "logits_all_nan": bool(all(l.isnan().all() for l in o.logits))}
        except Exception as e:
            out[name] = {"T": spec["T"], "ids": spec["ids"], "cfg": spec["cfg"], "error": type(e).__name__}
        print("forward error case", name, "->", out[name]["error"])
    return out


def helper_known_answers(g=None, ce=None):
    mm = importlib.import_module("eagle.mm_utils")
    ka = {"get_frame_indices": [], "tokenizer_image_token": [], "get_model_name_from_path": [], "get_seq_frames": []}
    if g is not None:
        ka["forward_errors"] = forward_errors(g, ce)
    for args in [(3000, 30.0, 2, 512), (18000, 29.97, 2, 512), (7200, 24.0, 2, 64), (500, 25.0, 2, 512),
                 (123456, 59.94, 1, 512), (10, 30.0, 2, 512), (1024, 1.0, 2, 512), (999, 23.976, 2, 16),
                 (2500, 25.0, 2, 512), (1537, 30.0, 2, 512)]:
        ka["get_frame_indices"].append({"args": list(args), "out": mm.get_frame_indices(*args)})

    class Tok:
        def __init__(self, bos=None):
            self.bos_token_id = bos

        def __call__(self, s):
            ids = [ord(c) for c in s]
            if self.bos_token_id is not None:
                ids = [self.bos_token_id] + ids
            return types.SimpleNamespace(input_ids=ids)

    for prompt, bos in [("<image>hi\n", None), ("a<image>b", 1), ("<image>", None), ("x<image>y<image>z", None),
                        ("no image", 1), ("<image>q\n", 1)]:
        ka["tokenizer_image_token"].append(
            {"prompt": prompt, "bos": bos, "out": mm.tokenizer_image_token(prompt, Tok(bos), -200)})
    for p in ["nvidia/VideoITG-8B", "x/y/checkpoint-500/", "/a/b/c", "model"]:
        ka["get_model_name_from_path"].append({"path": p, "out": mm.get_model_name_from_path(p)})

    # get_seq_frames lives in the lmms-eval plugin module, which needs loguru/accelerate/decord to import;
    # evaluate just that function's definition (run here from the reference file, nothing is copied).
    import ast
    src = open(f"{REF}/lmms_eval/models/videoitg.py").read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "get_seq_frames")
    ns = {"np": np}
    exec(compile(ast.Module([fn], []), "<ref:get_seq_frames>", "exec"), ns)
    for (total, fps, tfps, nf, mult) in [(18000, 29.97, 1, 512, 1), (123456, 59.94, 1, 512, 1), (300, 30.0, 1, 512, 1),
                                         (7200, 24.0, 1, 64, 1), (1000, 25.0, 1, 512, 8), (20000, 30.0, 2, 512, 1)]:
        try:
            out = ns["get_seq_frames"](total, fps, tfps, nf, mult)
            out = [int(x) for x in out]
        except Exception as e:  # record the failure mode too
            out = {"error": type(e).__name__}
        ka["get_seq_frames"].append({"total": total, "fps": fps, "target_fps": tfps, "num_frames": nf,
                                     "multiple": mult, "out": out})
    with open(os.path.join(HERE, "helpers.json"), "w") as f:
        json.dump(ka, f)
    print("helpers.json written")


def main():
    g, ce = install_shims()
    which = sys.argv[1:] or list(CASES) + ["train_pool"]
    for name in which:
        if name == "helpers":
            continue
        if name == "train_pool":
            train_mode_pool(g, ce)
            continue
        run_case(g, ce, name, CASES[name])
    helper_known_answers(g, ce)


if __name__ == "__main__":
    main()
