"""-m gpu: the whole selector forward through libvideoitg_hip.so vs (a) the committed golden logits produced by the
reference itself and (b) the oracle on the same seeded inputs.  Tolerance: |sigmoid score delta| <= 1e-3
(BASELINE.json north_star) and identical Top-K index sets."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = ["pool_post", "nopool_prepost", "oddgrid", "batch2_pad", "chunk520", "ragged_image", "two_images", "two_images_ragged"]
SCORE_TOL = 1e-3


def run_case(golden_dir, name, dtype, variant=0, q_prescaled=None):
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    z = np.load(os.path.join(golden_dir, f"{name}.npz"))
    cfg = ITGConfig.from_dict(json.loads(str(z["cfg"])))
    seed = int(z["seed"])
    T = z["T"].tolist()
    images = [synth.synth_pixel_values(t, cfg.vision.image_size, seed=1234 + i) for i, t in enumerate(T)]
    sd, vsd = synth.synth_llm_state_dict(cfg, seed), synth.synth_vision_state_dict(cfg, seed)
    if q_prescaled is None:
        pw = PackedWeights.from_state_dicts(cfg, sd, vsd, dtype, "cuda:0")
    else:                                                 # the tower's softmax scale folded into the q projection (default) or applied in the attention kernel
        pw = PackedWeights(cfg=cfg, dtype=dtype, device=torch.device("cuda:0"), vit_q_prescaled=q_prescaled)
        pw.pack_vision_embeddings(vsd)
        for i in range(cfg.num_vit_layers_needed()):
            pw.pack_vision_layer(vsd, i)
        pw.pack_projector_and_head(sd)
        for i in range(cfg.num_hidden_layers):
            pw.pack_llm_layer(sd, i)
    eng = SelectorEngine(pw, attn_variant=variant)
    ids, mask = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    got = eng.forward(ids, mask, [im.cuda() for im in images])
    torch.cuda.synchronize()
    return z, [g.float().cpu() for g in got], T


# Fixed, stated bounds (no model of this implementation's own rounding enters): fp16 - the reference's inference dtype and the
# headline - is held to north_star's 1e-3 on every fixture but chunk520 (a 520-frame toy: ONE pooled token per frame, random weights that
# spread the logits over [-5.9, 10.3]; 1.4e-3 - 1.7e-3 measured, held to 2e-3.  profiles/r04_chunk520.md: the error is flat 16-bit
# activation rounding from the tower on - no kernel, stream width or scale placement moves it - and the REFERENCE's own model.half()
# arithmetic is 2.7e-3 away from this fixture, tools/reference_fp16_error.py); bf16 is the secondary tier (8-bit mantissa on every MFMA operand) and is held to
# 2e-2 on these 2-3-layer random-weight toys (1.5e-2 measured; at full depth with the real widths it measures 1.4e-3,
# tests/test_fullsize_parity.py).
GOLDEN_TOL = {torch.float16: {"chunk520": 2e-3}, torch.bfloat16: {}}
GOLDEN_TOL_DEFAULT = {torch.float16: SCORE_TOL, torch.bfloat16: 2e-2}


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference_golden(golden_dir, name, dtype):
    z, got, T = run_case(golden_dir, name, dtype)
    tol = GOLDEN_TOL[dtype].get(name, GOLDEN_TOL_DEFAULT[dtype])
    worst = 0.0
    for b in range(len(got)):                     # one logits tensor per SAMPLE (a sample may hold several <image> tokens)
        ref = torch.from_numpy(z[f"logits{b}"])
        assert got[b].shape == ref.shape == (T[b], 1)
        assert torch.isfinite(got[b]).all()
        d = (torch.sigmoid(got[b]) - torch.sigmoid(ref)).abs().max().item()
        worst = max(worst, d)
        print(f"{name}[{b}] {dtype}: max|dscore|={d:.2e} max|dlogit|={(got[b] - ref).abs().max().item():.2e} (tol {tol:.1e})")
    assert worst <= tol, (worst, tol)


@pytest.mark.parametrize("name", CASES)
def test_topk_index_sets_match_reference(golden_dir, name):
    """Top-K index SET identical to the reference's for every K whose rank-K / rank-K+1 score gap exceeds the
    score tolerance (ties inside the tolerance are not decidable by any 16-bit run)."""
    from videoitg_amd import ops
    z, got, T = run_case(golden_dir, name, torch.float16)
    checked = 0
    for b in range(len(got)):                     # one logits tensor per SAMPLE (a sample may hold several <image> tokens)
        ref = torch.sigmoid(torch.from_numpy(z[f"logits{b}"]).view(-1))
        sref, oref = torch.sort(ref, descending=True, stable=True)
        _, order = ops.rank(got[b].view(-1).cuda())
        order = order.cpu().long()
        for k in range(1, T[b]):
            if (sref[k - 1] - sref[k]).item() > 2.5 * SCORE_TOL:
                assert set(order[:k].tolist()) == set(oref[:k].tolist()), (name, b, k)
                checked += 1
    assert checked >= 2


@pytest.mark.parametrize("name", ["pool_post", "oddgrid", "two_images"])
def test_q_prescale_on_and_off_both_match_the_reference(golden_dir, name):
    """PackedWeights.vit_q_prescaled: head_dim^-0.5 * log2(e) in the q rows of the tower's qkv weights + vitg_attention variant bit 1 (scores leave the
    matrix pipe ready for exp2) against the scale applied inside the attention kernel: each inside the golden tolerance on its own, and within two
    16-bit roundings of each other"""
    tol = GOLDEN_TOL[torch.float16].get(name, SCORE_TOL)
    outs = {}
    for qp in (True, False):
        z, got, T = run_case(golden_dir, name, torch.float16, q_prescaled=qp)
        for b in range(len(got)):
            ref = torch.from_numpy(z[f"logits{b}"])
            assert (torch.sigmoid(got[b]) - torch.sigmoid(ref)).abs().max().item() <= tol, (qp, name, b)
        outs[qp] = got
    for x, y in zip(outs[True], outs[False]):
        assert (torch.sigmoid(x) - torch.sigmoid(y)).abs().max().item() <= 2 * tol


@pytest.mark.parametrize("name", ["pool_post", "oddgrid"])
def test_tr_read_and_scalar_v_paths_agree(golden_dir, name):
    """same kernel, two ways of reading V^T out of LDS: identical arithmetic, so near-identical logits"""
    from videoitg_amd import ops
    ops.set_option("attn_sp", 2)              # per-tile kernel for both head sizes (the scalar-V cross-check lives there)
    ops.set_option("attn_ab", 2)              # (and not the out-of-phase hd-72 kernel, whose row sums are formed on the matrix pipe from the ROUNDED P)
    try:
        _, a, _ = run_case(golden_dir, name, torch.float16, variant=0)
        _, b, _ = run_case(golden_dir, name, torch.float16, variant=1)
    finally:
        ops.set_option("attn_sp", 0)
        ops.set_option("attn_ab", 0)
    for x, y in zip(a, b):
        assert (x - y).abs().max().item() < 1e-4


@pytest.mark.parametrize("name", ["pool_post", "chunk520"])
def test_attention_loop_structures_agree(golden_dir, name):
    """per-tile kernel (eager rescale) vs software-pipelined kernel (lazy rescale: P relative to a stale max, <= 2^8): the
    same real-number result with different 16-bit roundings of P.  Each must sit inside the golden tolerance on its own,
    and they differ from each other by no more than two such roundings (measured: tools/attn_variants_vs_golden.py)."""
    from videoitg_amd import ops
    tol = GOLDEN_TOL[torch.float16].get(name, SCORE_TOL)
    outs = []
    PROF_LIB = os.environ.get("VITG_PROFILING_LIB", "0") not in ("", "0")
    # 2 = per-tile kernels, 0 = the default (software-pipelined 8-wave loop for hd 128); 4 = the one-wave-per-SIMD kernel: profiling library only since round 6
    for sp in ((2, 4, 0) if PROF_LIB else (2, 0)):
        ops.set_option("attn_sp", sp)
        try:
            z, got, T = run_case(golden_dir, name, torch.float16, variant=0)
        finally:
            ops.set_option("attn_sp", 0)
        for bi in range(len(got)):
            ref = torch.from_numpy(z[f"logits{bi}"])
            assert (torch.sigmoid(got[bi]) - torch.sigmoid(ref)).abs().max().item() <= tol, (sp, name)
        outs.append(got)
    for other in outs[:-1]:
        for x, y in zip(other, outs[-1]):
            assert (torch.sigmoid(x) - torch.sigmoid(y)).abs().max().item() <= 2.0 * tol


def test_model_object_surface(golden_dir):
    """load_pretrained_model / model(...).logits / sigmoid-sort-topk as infer.py:15-21,60-79 does it."""
    from videoitg_amd import synth
    from videoitg_amd.constants import DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX
    from videoitg_amd.mm_utils import get_model_name_from_path, pad_sequence, tokenizer_image_token
    from videoitg_amd.model.builder import load_pretrained_model
    from videoitg_amd import ops
    from oracle import videoitg_oracle as O
    path = "synthetic:tiny:0"
    tok, model, proc, ctx = load_pretrained_model(path, None, get_model_name_from_path(path), device_map="cuda:0")
    model.half().eval().to(torch.device("cuda:0"))
    assert ctx == 2048
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (8, 40, 60, 3), dtype=np.uint8)
    pix = proc.preprocess(frames, return_tensors="pt")["pixel_values"]
    assert pix.shape == (8, 3, 56, 56) and pix.dtype == torch.float32 and pix.min() >= -1 and pix.max() <= 1
    video = pix.half().to("cuda:0")
    prompt = DEFAULT_IMAGE_TOKEN + "which frame shows the cat?" + "\n"
    ids = tokenizer_image_token(prompt, tok, IMAGE_TOKEN_INDEX, return_tensors="pt")
    pad_id = tok.pad_token_id if tok.pad_token_id is not None else tok.eos_token_id
    ids = pad_sequence(tok, [ids], batch_first=True, padding_value=pad_id).to("cuda:0")
    with torch.inference_mode():
        resp = model(ids, attention_mask=ids.ne(pad_id), images=[video])
        logits = resp.logits[0]
        assert logits.shape == (8, 1)
        scores = logits.sigmoid().view(-1)
    cfg = model.config
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    want = O.forward(cfg, sd, vsd, ids.cpu(), None, [pix.half().float()])[0].view(-1)
    assert (scores.float().cpu() - torch.sigmoid(want)).abs().max().item() <= SCORE_TOL
    s2, order = ops.rank(logits.view(-1).float())
    _, worder = O.rank_frames(want)
    frame_idx = list(range(0, 80, 10))
    sel = sorted(frame_idx[i] for i in order[:3].tolist())
    assert sel == O.topk_selection(want, frame_idx, 3)


def test_context_parallel_path_world1_equals_plain_forward(golden_dir):
    """The multi-GPU code path (frame shard + K/V all-gather over RCCL + logits all-gather) with a 1-rank NCCL group
    must reproduce the single-GPU forward bit for bit (same kernels, keys gathered from one rank)."""
    import socket
    import torch.distributed as dist
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    from videoitg_amd.parallel import ContextParallel
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        cfg = tiny_config(vision_token_num=72)
        sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
        eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0"))
        pix = synth.synth_pixel_values(8, cfg.vision.image_size, seed=1234).cuda()
        ids = torch.tensor([[11, 12, -200, 5, 6, 7]])
        a = eng.forward(ids, None, [pix])[0].view(-1).clone()
        b = ContextParallel(0, 1).forward(eng, ids, pix)
        torch.cuda.synchronize()
        assert torch.equal(a.cpu(), b.cpu())
    finally:
        dist.destroy_process_group()


# (frame counts below 16 keep the single-rank reference on the same 128-tile GEMM kernels as the ranks: at 304 rows it would switch to
# the 256-tile kernels, which differ from them in fp32 summation order - a kernel-set difference, not a sharding one)
# (4 and 8 ranks on one GPU run in tests/test_gpu_cp_fullsize.py at full size; the 8-rank toy cost 85 s of process start-up, a 4-rank one 34 s: VERDICT r5 item 6)
# (a 3-rank toy run - a middle rank without text rows - cost 16 s of process start-up; middle ranks run at full size in tests/test_gpu_cp_fullsize.py
# (8 and 4 ranks) and in the gloo CPU tests, so round 6 traded it for the two prompt shapes below; all toy runs share ONE pair of rank processes: conftest cp_toy)
@pytest.mark.parametrize("case", ["2-9", "2-5-image_only", "2-4-no_text_after"])
def test_context_parallel_multi_rank_on_one_gpu(cp_toy, case):
    """WORLD > 1 with the real kernels: two processes share cuda:0 and exchange over gloo (RCCL refuses two ranks on one
    device).  Frame shards are ragged (9 frames / 2 ranks, prompt tokens on the first and last rank), so global positions, the
    padded K|V all-gather + gather-rows compaction and the logits gather are all exercised; round 6: a prompt that is the <image> token alone - no rank
    owns text - and one with nothing after it.  Same arithmetic per token as the single-rank forward -> identical logits."""
    from conftest import CP_TOY_CASES
    got, want = cp_toy[case]
    assert got.shape == want.shape == (CP_TOY_CASES[case]["frames"],)
    assert (got - want).abs().max().item() <= 1e-5


def test_context_parallel_fp8_weight_path_on_one_gpu(cp_toy):
    """BASELINE.json configs[4] names the fp8 weight path at 1 AND 8 GPUs: the context-parallel forward on fp8 weights (per-token
    activation scales are row-local, so sharding the rows changes nothing but the kernels' block partition).  Two ranks on one GPU
    over gloo against the single-rank fp8 forward (ragged: 5 + 4 frames; the full-size fp8 run covers 8 ranks); the bound is the fp8 tier's own (an fp32
    rounding difference in front of a quantiser can move an e4m3 code)."""
    got, want = cp_toy["fp8"]
    assert got.shape == want.shape == (9,) and torch.isfinite(got).all()
    d = (torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item()
    print(f"context-parallel fp8, 2 ranks: max|dscore| vs single-rank fp8 = {d:.2e}")
    assert d <= 5e-2, d


@pytest.mark.parametrize("vit_layers,llm_layers", [(6, 5)])      # ((3, 2) went when test_full_width_random_shapes_equal_the_oracle arrived: six drawn shapes at depth 1 + 1)
def test_full_width_reduced_depth_parity(vit_layers, llm_layers):
    """Real layer widths (SigLIP 1152/4304/16x72, Qwen2 3584/18944, 28 q / 4 kv heads x128, pos-emb interpolation to a
    non-native grid, 16->10 token pool) at reduced depth / frame count so the fp32 oracle finishes on the host: every
    production kernel instantiation (hd 72 and hd 128 attention, 256x256 persistent GEMM incl. K=4352/18944, SwiGLU,
    LN/RMS at 1152/3584) is on this path."""
    from oracle import videoitg_oracle as O
    from videoitg_amd import ops, synth
    from videoitg_amd.config import ITGConfig, VisionConfig
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    cfg = ITGConfig(vocab_size=2048, num_hidden_layers=llm_layers, vision_token_num=1600,
                    vision=VisionConfig(image_size=384, num_hidden_layers=vit_layers))
    T, R = 16, 224                                               # 16x16 grid from a 27x27 table -> bicubic resample
    sd, vsd = synth.synth_llm_state_dict(cfg, 3), synth.synth_vision_state_dict(cfg, 3)
    pix = synth.synth_pixel_values(T, R, seed=77)
    ids = torch.cat([torch.tensor([-200]), synth.synth_prompt_ids(60, cfg.vocab_size)]).unsqueeze(0)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    want = O.forward(cfg, sd, vsd, ids, None, [pix])[0].view(-1)
    eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0"))
    got = eng.forward(ids, None, [pix.cuda()])[0].view(-1).float().cpu()
    d = (torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item()
    print(f"full-width {vit_layers - 1}+{llm_layers} layers: max|dscore|={d:.2e} max|dlogit|={(got - want).abs().max().item():.2e} "
          f"logit range [{want.min():.2f}, {want.max():.2f}]")
    assert d <= SCORE_TOL, d
    _, order = ops.rank(got.cuda())
    sref, oref = torch.sort(torch.sigmoid(want), descending=True, stable=True)
    for k in range(1, T):
        if (sref[k - 1] - sref[k]).item() > 2.5 * SCORE_TOL:
            assert set(order[:k].cpu().tolist()) == set(oref[:k].tolist()), k


@pytest.mark.parametrize("name", ["pool_post", "oddgrid", "chunk520"])
def test_dead_row_skipping_does_not_change_results(golden_dir, name):
    """Running the last tower layer only on the tokens the pool reads must give the same logits as running it on all
    tokens (each output row is computed by the same arithmetic; only rows nothing reads are dropped)."""
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    z = np.load(os.path.join(golden_dir, f"{name}.npz"))
    cfg = ITGConfig.from_dict(json.loads(str(z["cfg"])))
    T = z["T"].tolist()
    images = [synth.synth_pixel_values(t, cfg.vision.image_size, seed=1234 + i).cuda() for i, t in enumerate(T)]
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0"))
    ids, mask = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    G = cfg.vision.grid
    assert eng.pool_sources(G, cfg.pooled_grid(T[0], G)) is not None
    a = [t.clone() for t in eng.forward(ids, mask, images)]
    eng.skip_dead_rows = False
    b = eng.forward(ids, mask, images)
    for x, y in zip(a, b):
        assert (x - y).abs().max().item() < 2e-3


@pytest.fixture(scope="module")
def full_engine():
    from videoitg_amd.config import ITGConfig
    from videoitg_amd.engine import SelectorEngine
    from videoitg_amd.synth_device import packed_random_weights
    eng = SelectorEngine(packed_random_weights(ITGConfig(), torch.float16, "cuda:0", seed=0))
    yield eng


# BASELINE.json configs: [1] 512 x 448^2 (the bench workload), B0 512 x 384^2 (the only reference-native shape: 27x27 grid, no
# pos-emb interpolation, 5x5 pool), [0] = C1 64 x 384^2 (16x16 pool, N = 16 445: the LLM-heavy case)
@pytest.mark.parametrize("T,R", [(512, 448), (512, 384), (64, 384)])
def test_full_size_invariants(full_engine, T, R):
    """BASELINE.json configurations at FULL size (all 26 + 28 layers, real widths): the fp32 oracle cannot run
    this on a host in test time, so parity is carried by size-independent properties:
      * run-to-run bit-identical logits (no atomics, no order-dependent reductions anywhere on the path);
      * the same video through DIFFERENT kernels (128x128 register-staged GEMM instead of the 256-wide LDS-DMA ones; per-tile
        attention loop instead of the software-pipelined one) agrees to the 16-bit rounding level: a kernel that is wrong
        at sizes the small fixtures do not reach (K = 4352 / 18 944, S = 1024 / 12 860, ragged last tiles) shows up here;
      * frame order of the ranking is a permutation, scores sorted, Top-32 stable across the kernel sets wherever the score
        gap exceeds the tolerance."""
    from videoitg_amd import ops
    from videoitg_amd import synth
    eng = full_engine
    cfg = eng.cfg
    g = torch.Generator().manual_seed(1234)
    pix = (torch.rand((T, 3, R, R), generator=g) * 2 - 1).to(torch.float16).cuda()
    ids = torch.cat([torch.tensor([-200]), synth.synth_prompt_ids(60, cfg.vocab_size)]).view(1, -1)

    def run():
        out = eng.forward(ids, None, [pix])[0].view(-1).float()
        torch.cuda.synchronize()
        return out.cpu()
    a = run()
    assert a.shape == (T,) and torch.isfinite(a).all()
    assert torch.equal(a, run())                                      # determinism
    ops.set_option("gemm_tile", 128)
    ops.set_option("attn_sp", 2)
    try:
        b = run()
    finally:
        ops.set_option("gemm_tile", 0)
        ops.set_option("attn_sp", 0)
    # random-init weights put the logits far from 0 (saturated sigmoids), so the comparison is made on the logits: 4e-3 in logit
    # is what 1e-3 in score means where the sigmoid is steepest
    dl = (a - b).abs().max().item()
    print(f"full size: max|dlogit| between kernel sets = {dl:.2e} (logit mean {a.mean().item():.2f}, std {a.std().item():.2f}), "
          f"max|dscore| = {(torch.sigmoid(a) - torch.sigmoid(b)).abs().max().item():.2e}")
    assert dl <= 4e-3, dl                                             # two independent 16-bit roundings of the same real result
    scores, order = ops.rank(a.cuda())
    order = order.cpu().long()
    assert sorted(order.tolist()) == list(range(T))
    ranked = scores.cpu()[order]                                      # scores come back in frame order; `order` ranks them
    assert (ranked[:-1] >= ranked[1:]).all()
    srt, oa = torch.sort(a, descending=True, stable=True)
    ob = torch.sort(b, descending=True, stable=True)[1]
    for k in (1, 8, 32):
        if k < T and (srt[k - 1] - srt[k]).item() > 2 * dl + 1e-6:
            assert set(oa[:k].tolist()) == set(ob[:k].tolist()), k


@pytest.mark.parametrize("T,R,prompt", [(1, 384, 60), (7, 448, 1), (22, 384, 0), (23, 384, 17), (200, 384, 60), (513, 384, 5), (600, 392, 60)])
def test_full_width_shape_sweep(full_engine, T, R, prompt):
    """Frame counts on both sides of every branch of the pooled-grid rule (T <= 22: no pooling at 27x27; 23: 26x26; 200: 9x9;
    > 512: chunked tower / pool, grid 5 -> 4 at 600), a lone frame, an empty prompt, and a resolution whose grid is not the
    checkpoint's (392 = 28 * 14: position embeddings interpolated).  Real widths and depths; checks shape, finiteness,
    run-to-run determinism and agreement of the two kernel sets."""
    from videoitg_amd import ops, synth
    eng = full_engine
    cfg = eng.cfg
    g = torch.Generator().manual_seed(77 + T)
    pix = (torch.rand((T, 3, R, R), generator=g) * 2 - 1).to(torch.float16).cuda()
    ids = torch.cat([torch.tensor([-200]), synth.synth_prompt_ids(prompt, cfg.vocab_size)]).view(1, -1) if prompt else torch.tensor([[-200]])

    def run():
        out = eng.forward(ids, None, [pix])[0]
        torch.cuda.synchronize()
        return out.float().cpu()
    a = run()
    assert a.shape == (T, 1) and torch.isfinite(a).all()
    assert torch.equal(a, run())
    ops.set_option("gemm_tile", 128)
    ops.set_option("attn_sp", 2)
    ops.set_option("attn_ab", 2)
    try:
        b = run()
    finally:
        for k in ("gemm_tile", "attn_sp", "attn_ab"):
            ops.set_option(k, 0)
    assert (a - b).abs().max().item() <= 4e-3


def test_infer_entry_point_runs_offline():
    """python infer.py --model synthetic:tiny --video synthetic: the reference's entry-point flow (sample -> preprocess ->
    prompt -> model -> sigmoid/sort -> Top-K ascending) end to end on the GPU with the synthetic decoder stand-in."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    with tempfile.TemporaryDirectory() as vis:
        out = subprocess.run([sys.executable, os.path.join(root, "infer.py"), "--model", "synthetic:tiny", "--video", "synthetic", "--topk", "8",
                              "--save-dir", vis], capture_output=True, text=True, timeout=600, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        import ast
        lines = out.stdout.strip().splitlines()
        sel = ast.literal_eval(next(l for l in lines if l.startswith("[")))
        assert len(sel) == 8 and sel == sorted(sel) and len(set(sel)) == 8
        # infer.py:81-94 of the reference: the selected frames as JPEGs, frame_<rank>_idx<original index>.jpg
        assert sorted(os.listdir(vis)) == [f"frame_{i:03d}_idx{f}.jpg" for i, f in enumerate(sel)]


@pytest.mark.parametrize("container", ["avi", "mp4"])
def test_infer_entry_point_on_a_real_video_file(container):
    """(round 6: also an .mp4 whose track is Motion-JPEG as MP4 muxers label it - 'mp4v' + esds object type 0x6C - through the built-in demux)
    python infer.py --video clip.avi on a REAL file (infer.py:44-51, 81-94): a 20 s MJPEG-in-AVI clip written here, decoded by
    the built-in reader (decord is not in the image), sampled at 2 fps by get_frame_indices, scored on the GPU; the Top-K are original frame
    indices out of the sampler's list and the saved JPEGs are those frames of the file."""
    import ast
    import os
    import subprocess
    import sys
    import tempfile
    import numpy as np
    from PIL import Image
    from videoitg_amd import mm_utils, video_io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    T, fps = 600, 29.97
    clip = np.zeros((T, 72, 96, 3), np.uint8)
    for i in range(T):
        clip[i] = (i * 7) % 200 + 20
        clip[i, 10:30, (i % 60):(i % 60) + 20] = (240, 30, 60)
    with tempfile.TemporaryDirectory() as tmp:
        path, vis = os.path.join(tmp, f"clip.{container}"), os.path.join(tmp, "vis")
        if container == "avi":
            video_io.write_avi(path, clip, fps=fps, codec="MJPG")
        else:
            video_io.write_mov(path, clip, fps=fps, codec="mp4v-jpeg", per_chunk=25)
        out = subprocess.run([sys.executable, os.path.join(root, "infer.py"), "--model", "synthetic:tiny", "--video", path, "--topk", "6",
                              "--num-frames", "32", "--save-dir", vis], capture_output=True, text=True, timeout=600, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        sel = ast.literal_eval(next(l for l in out.stdout.strip().splitlines() if l.startswith("[")))
        sampled = mm_utils.get_frame_indices(T, fps, 2, 32)
        assert len(sel) == 6 and sel == sorted(sel) and set(sel) <= set(sampled)
        assert sorted(os.listdir(vis)) == [f"frame_{i:03d}_idx{f}.jpg" for i, f in enumerate(sel)]
        for i, f in enumerate(sel):                              # the saved frame IS frame f of the file (grey level of its lower half)
            im = np.asarray(Image.open(os.path.join(vis, f"frame_{i:03d}_idx{f}.jpg")))
            assert abs(int(im[60, 48, 1]) - int(clip[f, 60, 48, 1])) <= 6, (i, f)


def test_graph_captured_forward_replays_bit_identically(golden_dir):
    """HIP-graph capture of the per-video forward (BASELINE.json configs[3]): replay with NEW pixel values and NEW prompt
    token values (same layout) must equal the eager forward bit for bit; a different layout is refused."""
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.engine import GraphedForward, PackedWeights, SelectorEngine
    cfg = tiny_config(vision_token_num=72)
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0"))
    ids0 = torch.tensor([[11, 12, -200, 5, 6, 7]])
    pix0 = synth.synth_pixel_values(8, cfg.vision.image_size, seed=1).half().cuda()
    gf = GraphedForward(eng, ids0, pix0)
    for seed, ids in [(1, ids0), (2, torch.tensor([[30, 31, -200, 40, 41, 42]])), (3, torch.tensor([[9, 8, -200, 7, 6, 5]]))]:
        pix = synth.synth_pixel_values(8, cfg.vision.image_size, seed=seed).half().cuda()
        got = gf(ids, pix)
        want = eng.forward(ids, None, [pix])[0].view(-1)
        torch.cuda.synchronize()
        assert torch.equal(got, want), seed
    with pytest.raises(ValueError):
        gf(torch.tensor([[11, -200, 5, 6, 7, 8]]), pix0)                 # <image> moved
    with pytest.raises(ValueError):
        gf(ids0, pix0[:6])                                               # fewer frames


def test_degenerate_samples_as_the_reference(golden_dir):
    """A sample without <image> -> NaN logits of shape [images[b].shape[0], 1]; visual rows that do not divide by images[b].shape[0]
    -> RuntimeError: what the reference itself does (tests/golden/helpers.json "forward_errors", recorded by running it)."""
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    ka = json.load(open(os.path.join(golden_dir, "helpers.json")))["forward_errors"]
    for name, rec in ka.items():
        cfg = tiny_config(**rec["cfg"])
        sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
        eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0"))
        images = [synth.synth_pixel_values(t, cfg.vision.image_size, seed=1234 + i).cuda() for i, t in enumerate(rec["T"])]
        ids = torch.tensor(rec["ids"])
        if rec["error"]:
            with pytest.raises(RuntimeError):
                eng.forward(ids, None, images)
        else:
            lg = eng.forward(ids, None, images)
            assert [list(l.shape) for l in lg] == rec["logits_shapes"]
            assert all(bool(l.isnan().all()) for l in lg) == rec["logits_all_nan"]


def test_mixed_batch_of_single_and_double_image_samples_equals_the_oracle(golden_dir):
    """B = 2: sample 0 holds two <image> tokens (clips 0 and 1), sample 1 one (clip 2); the head of sample b reshapes by images[b].shape[0]
    (the reference's indexing, grounding_qwen2.py:153).  Compared with the oracle, which is pinned to the reference on the two-image
    fixtures."""
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    from oracle import videoitg_oracle as O
    cfg = tiny_config(vision_token_num=72)
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    T = [6, 6, 8]
    images = [synth.synth_pixel_values(t, cfg.vision.image_size, seed=4321 + i) for i, t in enumerate(T)]
    ids = torch.tensor([[11, -200, 5, 6, -200, 7, 8], [-200, 30, 31, 32, 0, 0, 0]])
    mask = torch.tensor([[1] * 7, [1, 1, 1, 1, 0, 0, 0]]).bool()
    want = O.forward(cfg, sd, vsd, ids, mask, images)
    eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0"))
    got = eng.forward(ids, mask, [im.cuda() for im in images])
    assert len(got) == len(want) == 2
    for g, w in zip(got, want):
        assert g.shape == w.shape
        assert (torch.sigmoid(g.float().cpu()) - torch.sigmoid(w)).abs().max().item() <= SCORE_TOL


@pytest.mark.parametrize("name", CASES)
def test_residual16_tower_stream_matches_reference_golden(golden_dir, name):
    """the tower's residual stream in fp16 (the reference's own arithmetic: model.half()) instead of fp32: same goldens, same bound"""
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    z = np.load(os.path.join(golden_dir, f"{name}.npz"))
    cfg = ITGConfig.from_dict(json.loads(str(z["cfg"])))
    T = z["T"].tolist()
    images = [synth.synth_pixel_values(t, cfg.vision.image_size, seed=1234 + i).cuda() for i, t in enumerate(T)]
    sd, vsd = synth.synth_llm_state_dict(cfg, int(z["seed"])), synth.synth_vision_state_dict(cfg, int(z["seed"]))
    eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0"))
    eng.residual16 = True
    ids, mask = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    got = [g.float().cpu() for g in eng.forward(ids, mask, images)]
    assert "vit_x16" in eng._ws and "vit_x" not in eng._ws                  # the 16-bit stream really ran
    tol = GOLDEN_TOL[torch.float16].get(name, SCORE_TOL)
    for b in range(len(got)):
        ref = torch.from_numpy(z[f"logits{b}"])
        assert got[b].shape == ref.shape
        assert (torch.sigmoid(got[b]) - torch.sigmoid(ref)).abs().max().item() <= tol


def _rccl_worker(rank, world, frames, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    from videoitg_amd.parallel import ContextParallel
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        cfg = tiny_config(vision_token_num=72)
        sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
        eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, dev))
        pix = synth.synth_pixel_values(frames, cfg.vision.image_size, seed=1234).to(dev)
        ids = torch.tensor([[11, 12, -200, 5, 6, 7]])
        cp = ContextParallel(rank, world)
        fr = cp.local_frames(frames)
        a = cp.forward(eng, ids, pix[fr.start:fr.stop].clone(), total_frames=frames)
        b = cp.forward(eng, ids, pix[fr.start:fr.stop].clone(), total_frames=frames)     # asynchronous exchanges: must be repeatable
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        if rank == 0:
            q.put(a.float().cpu())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device (the one-GPU boxes run the "
                                                           "same plan over gloo, test_context_parallel_multi_rank_on_one_gpu)")
@pytest.mark.parametrize("world,frames", [(2, 9), (2, 8), (8, 19), (8, 64), (4, 13)])
def test_context_parallel_over_rccl(world, frames):
    """One rank per GPU over RCCL: K|V written in place into the gather buffer, the all-gather launched asynchronously under the Q
    projection, ragged shards (9 frames / 2 ranks, prompt tokens on the first and last rank).  Runs the moment a multi-GPU box exists;
    logits must equal the single-rank forward (same arithmetic per token)."""
    import torch.multiprocessing as mp
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    if torch.cuda.device_count() < world:                       # (the 8- and 4-rank cases run the moment a node shows that many GPUs)
        pytest.skip(f"{world} ranks need {world} GPUs, {torch.cuda.device_count()} visible")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, frames, 29600 + frames + world, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, p.exitcode
    cfg = tiny_config(vision_token_num=72)
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0"))
    pix = synth.synth_pixel_values(frames, cfg.vision.image_size, seed=1234).cuda()
    want = eng.forward(torch.tensor([[11, 12, -200, 5, 6, 7]]), None, [pix])[0].view(-1).float().cpu()
    assert (got - want).abs().max().item() <= 1e-5


def test_training_mode_pooled_grids_match_the_reference(golden_dir):
    """model.train() pools every video to the random grid the reference's projector would draw under the same torch seed
    (mlp_proj.py:51-52); logits against the reference's own .train()-mode forward (train_pool.npz).  Forward values only."""
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    from videoitg_amd.model.grounding_qwen2 import EagleQwenG
    z = np.load(os.path.join(golden_dir, "train_pool.npz"))
    cfg = ITGConfig.from_dict(json.loads(str(z["cfg"])))
    T = z["T"].tolist()
    images = [synth.synth_pixel_values(t, cfg.vision.image_size, seed=1234 + i).cuda() for i, t in enumerate(T)]
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    ids, mask = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0"))
    got = eng.forward(ids, mask, images, pooled_grids=z["grids"].tolist())
    for b in range(len(T)):
        ref = torch.from_numpy(z[f"logits{b}"])
        assert got[b].shape == ref.shape
        assert (torch.sigmoid(got[b].float().cpu()) - torch.sigmoid(ref)).abs().max().item() <= SCORE_TOL
    # the model object draws the grids itself in training mode - same calls to torch's global generator as the reference's projector
    model = EagleQwenG(cfg, sd, vsd, torch_dtype=torch.float16, device="cuda:0")
    model.train()
    torch.manual_seed(int(z["torch_seed"]))
    out = model(ids, attention_mask=mask, images=images)
    for b in range(len(T)):
        assert torch.equal(out.logits[b].float().cpu(), got[b].float().cpu())
    model.eval()
    ev = model(ids, attention_mask=mask, images=images)                 # eval rule: [3, 3] - a different pooled grid for the 5-frame video
    assert ev.logits[1].shape == got[1].shape and not torch.equal(ev.logits[1].float().cpu(), got[1].float().cpu())


@pytest.mark.parametrize("seed", list(range(16)))
def test_randomly_drawn_batch_structures_equal_the_reference(golden_dir, seed):
    """Structures nobody chose (round 6, tests/structures.py): batch of 1-3 prompts, padding on either side (the mask decides, eagle_archv1.py:272), text
    of random length around the <image> token, clips of 1-11 frames, optionally two clips in the first prompt, a token budget that pools (3 x 3, 2 x 2,
    1 x 1) or does not, image sides 56 / 70 - run through THE REFERENCE (tests/golden/make_random_structures.py -> random_structures.npz; the oracle agrees
    with it to 2e-5, tests/test_oracle_golden.py).  The operand-split tier - whose rounding noise is ~1e-6 here - is held to 2e-4 in LOGIT, so a
    structural slip (a token one row off, a mask bit) cannot hide behind rounding; the fp16 default to north_star's 1e-3."""
    import structures as ST
    from videoitg_amd import synth
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    z = np.load(os.path.join(golden_dir, "random_structures.npz"))
    s = ST.draw(seed)
    assert np.array_equal(z[f"s{seed}_ids"], s["ids"].numpy()) and np.array_equal(z[f"s{seed}_mask"], s["mask"].numpy()) and z[f"s{seed}_T"].tolist() == s["T"]
    cfg, images, ids, mask, B, T, what = ST.config(s), ST.images(s), s["ids"], s["mask"], s["B"], s["T"], s["what"]
    want = [torch.from_numpy(z[f"s{seed}_logits{b}"]) for b in range(B)]
    sd, vsd = synth.synth_llm_state_dict(cfg, seed), synth.synth_vision_state_dict(cfg, seed)
    # the structural gate first: the split tier, whose own rounding is ~1e-6 on these toys
    eng2 = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0", vit_q_prescaled=False), operand_split=True)
    got2 = eng2.forward(ids, mask, [im.half().cuda() for im in images])
    assert len(got2) == B, what
    assert [tuple(g2.shape) for g2 in got2] == [tuple(wl.shape) for wl in want], what
    worst = max((g2.float().cpu() - wl).abs().max().item() for g2, wl in zip(got2, want))
    print(f"{what}: split tier max |dlogit| vs the reference = {worst:.2e}")
    assert worst <= 2e-4, what
    # the fp16 default on these 2-3-layer random-weight toys: held to the toy-class bound the reference's own `chunk520` golden has above (2e-3: flat 16-bit
    # activation rounding where a frame is one or a few pooled tokens and the logits spread over +-10 measured up to 1.8e-3 on these draws - seeds 2, 7, 15 -
    # and the reference's own model.half() arithmetic is 2.7e-3 from chunk520, tools/reference_fp16_error.py); north_star's 1e-3 is asserted at full depth
    # and width (tests/test_fullsize_parity.py) and on the hand-written goldens.  Reported per draw.
    eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0"))
    got = eng.forward(ids, mask, [im.cuda() for im in images])
    worst16 = 0.0
    for gl, wl in zip(got, want):
        assert gl.shape == wl.shape, what
        worst16 = max(worst16, (torch.sigmoid(gl.float().cpu()) - torch.sigmoid(wl)).abs().max().item())
    print(f"{what}: fp16 default max |dscore| vs the reference = {worst16:.2e}")
    assert worst16 <= GOLDEN_TOL[torch.float16]["chunk520"], what


@pytest.mark.parametrize("seed", list(range(12)))
def test_randomly_drawn_architectures_equal_the_reference(golden_dir, seed):
    """Architectures nobody chose (round 6, tests/structures.py: draw_arch): tower widths 72 .. 288 with FFN widths that are multiples of 4 only, 2-4 layers,
    image sides 56 / 62 / 70 / 84; decoders of 2 .. 14 query heads over 1-2 KV heads (GQA groups 1 / 2 / 3 / 7), FFN widths that are multiples of 8 only, 1-3
    layers, two rope bases - every padding rule of the engine (pad64 of widths, 32-row SwiGLU interleave, K multiples of 64) meets sizes it was not tuned on.
    Expected values from THE REFERENCE (random_archs.npz).  Operand-split tier <= 2e-4 in logit = the structural gate; fp16 default at the toy-class 2e-3."""
    import structures as ST
    from videoitg_amd import synth
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    z = np.load(os.path.join(golden_dir, "random_archs.npz"))
    s = ST.draw_arch(seed)
    assert json.loads(str(z[f"a{seed}_arch"])) == json.loads(json.dumps(s["arch"])) and np.array_equal(z[f"a{seed}_ids"], s["ids"].numpy())
    cfg, images, what = ST.arch_config(s), ST.arch_images(s), s["what"]
    want = torch.from_numpy(z[f"a{seed}_logits0"])
    sd, vsd = synth.synth_llm_state_dict(cfg, seed), synth.synth_vision_state_dict(cfg, seed)
    eng2 = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0", vit_q_prescaled=False), operand_split=True)
    got2 = eng2.forward(s["ids"], s["mask"], [im.half().cuda() for im in images])[0].float().cpu()
    assert got2.shape == want.shape, what
    e2 = (got2 - want).abs().max().item()
    eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0"))
    got = eng.forward(s["ids"], s["mask"], [im.cuda() for im in images])[0].float().cpu()
    e16 = (torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item()
    print(f"{what}: split tier max |dlogit| = {e2:.2e}, fp16 default max |dscore| = {e16:.2e}")
    assert e2 <= 2e-4, what
    assert e16 <= GOLDEN_TOL[torch.float16]["chunk520"], what


@pytest.mark.parametrize("content", ["black", "white", "zeros", "checker", "one_hot_frame", "fp16_extremes"])
def test_degenerate_pixel_content_equals_the_oracle(content):
    """frames a decoder really delivers at the edges: all-black / all-white clips (every patch identical: the tower's LayerNorm sees rows that differ by
    the position table only), mid-grey zeros, a +-1 checkerboard at pixel pitch (the highest spatial frequency a patch can hold), one bright frame among
    black ones, and values at the edge of what the processor can emit (+-1 exactly, 16-bit exact).  Oracle on the same tensors; split tier as the
    structural gate, fp16 default at the toy-class bound."""
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    from oracle import videoitg_oracle as O
    cfg = tiny_config(vision_token_num=72)
    R, T = cfg.vision.image_size, 8
    pix = torch.zeros((T, 3, R, R))
    if content == "black":
        pix -= 1.0
    elif content == "white":
        pix += 1.0
    elif content == "checker":
        yy, xx = torch.meshgrid(torch.arange(R), torch.arange(R), indexing="ij")
        pix += ((yy + xx) % 2 * 2 - 1).float()
    elif content == "one_hot_frame":
        pix -= 1.0
        pix[5] = 1.0
    elif content == "fp16_extremes":
        g = torch.Generator().manual_seed(3)
        pix = (torch.randint(0, 2, (T, 3, R, R), generator=g).float() * 2 - 1) * torch.tensor([1.0, 2.0 ** -8, 0.99609375])[torch.randint(0, 3, (T, 3, R, R), generator=g)]
    sd, vsd = synth.synth_llm_state_dict(cfg, 1), synth.synth_vision_state_dict(cfg, 1)
    ids = torch.tensor([[7, -200, 5, 6, 9]])
    want = O.forward(cfg, sd, vsd, ids, None, [pix])[0]
    assert torch.isfinite(want).all()
    eng2 = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0", vit_q_prescaled=False), operand_split=True)
    got2 = eng2.forward(ids, None, [pix.half().cuda()])[0].float().cpu()
    eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0"))
    got = eng.forward(ids, None, [pix.cuda()])[0].float().cpu()
    e2, e16 = (got2 - want).abs().max().item(), (torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item()
    print(f"{content}: split tier max |dlogit| = {e2:.2e}, fp16 default max |dscore| = {e16:.2e}")
    assert torch.isfinite(got).all() and torch.isfinite(got2).all()
    assert e2 <= 2e-4 and e16 <= GOLDEN_TOL[torch.float16]["chunk520"], content


@pytest.mark.parametrize("seed", list(range(6)))
def test_full_width_random_shapes_equal_the_oracle(seed):
    """the REAL layer widths (SigLIP 1152 / 4304 / 16 x 72, Qwen2 3584 / 18944, 28 q / 4 kv heads) at depth 1 + 1 on drawn frame counts, image sides (native,
    ragged, tiny: the position table is resampled to every grid), budgets and prompt lengths: ragged M against the production tile shapes at the production
    N and K.  Oracle on the host; split tier <= 2e-4 in logit as the structural gate, fp16 default <= north_star's 1e-3."""
    from oracle import videoitg_oracle as O
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig, VisionConfig
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    g = torch.Generator().manual_seed(300 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g).item())       # noqa: E731
    T, R = ri(1, 10), (56, 70, 126, 224, 230, 384)[ri(0, 5)]
    cfg = ITGConfig(vocab_size=2048, num_hidden_layers=1, vision_token_num=(64, 400, 1600, 16384)[ri(0, 3)], vision=VisionConfig(image_size=384, num_hidden_layers=2))
    sd, vsd = synth.synth_llm_state_dict(cfg, seed), synth.synth_vision_state_dict(cfg, seed)
    pix = synth.synth_pixel_values(T, R, seed=900 + seed)
    n_pre, n_post = ri(0, 6), ri(0, 24)
    ids = torch.cat([synth.synth_prompt_ids(n_pre, cfg.vocab_size), torch.tensor([-200]), synth.synth_prompt_ids(n_post, cfg.vocab_size)]).unsqueeze(0)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    want = O.forward(cfg, sd, vsd, ids, None, [pix])[0].view(-1)
    what = f"seed {seed}: T={T} R={R} grid {R // 14} -> {cfg.pooled_grid(T, R // 14)}, prompt {n_pre}+{n_post}, budget {cfg.vision_token_num}"
    eng2 = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0", vit_q_prescaled=False), operand_split=True)
    got2 = eng2.forward(ids, None, [pix.half().cuda()])[0].view(-1).float().cpu()
    eng2.release_workspace()
    del eng2
    eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0"))
    got = eng.forward(ids, None, [pix.cuda()])[0].view(-1).float().cpu()
    e2, e16 = (got2 - want).abs().max().item(), (torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item()
    print(f"{what}: split tier max |dlogit| = {e2:.2e}, fp16 default max |dscore| = {e16:.2e}")
    assert e2 <= 2e-4, what
    assert e16 <= SCORE_TOL, what
