"""-m gpu: hipGraph-captured per-video forward (BASELINE.json configs[3]), prompt-length buckets, workspace growth, and the
torch.ops.videoitg.* registration."""
import torch
import pytest

pytestmark = pytest.mark.gpu


def tiny_engine(dtype=torch.float16, fp8=False):
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    cfg = tiny_config()
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    return cfg, SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, dtype, "cuda:0", fp8=fp8))


def test_graph_replay_on_the_fp8_weight_path():
    """configs[3] x configs[4]: the captured forward on fp8 weights (per-token quantisers, MX block-scaled MLP hand-over) replays bit for bit
    what the eager fp8 forward computes, for every prompt length of the bucket"""
    from videoitg_amd import synth
    from videoitg_amd.engine import GraphCache
    cfg, eng = tiny_engine(fp8=True)
    assert eng.fp8_mx
    cache = GraphCache(eng, bucket=32)
    pix = synth.synth_pixel_values(8, cfg.vision.image_size, seed=5).half().cuda()
    for L in (7, 32, 19):
        ids = torch.cat([torch.tensor([-200]), synth.synth_prompt_ids(L, cfg.vocab_size, seed=L)])
        want = eng.forward(ids.unsqueeze(0), None, [pix])[0].view(-1).clone()
        got = cache(ids, pix)
        assert torch.isfinite(got).all() and torch.equal(got, want), (L, (got - want).abs().max().item())
    assert cache.captures == 1


def test_one_capture_serves_every_prompt_length_of_its_bucket():
    """Prompts of 1..32 text tokens share ONE captured graph (rows beyond the real prompt are padding the attention kernels
    exclude by a device-side key count); every replay must equal the eager, unpadded forward bit for bit - and a second
    bucket triggers exactly one more capture."""
    from videoitg_amd import synth
    from videoitg_amd.engine import GraphCache
    cfg, eng = tiny_engine()
    cache = GraphCache(eng, bucket=32)
    pix = synth.synth_pixel_values(8, cfg.vision.image_size, seed=5).half().cuda()
    for L in (7, 32, 1, 19, 33, 40, 7):
        ids = torch.cat([torch.tensor([-200]), synth.synth_prompt_ids(L, cfg.vocab_size, seed=L)])
        want = eng.forward(ids.unsqueeze(0), None, [pix])[0].view(-1).clone()
        got = cache(ids, pix)
        assert torch.equal(got, want), (L, (got - want).abs().max().item())
    assert cache.captures == 2
    # text BEFORE <image> changes the layout key -> its own capture, same equality
    ids = torch.cat([torch.tensor([3, 4]), torch.tensor([-200]), synth.synth_prompt_ids(9, cfg.vocab_size)])
    assert torch.equal(cache(ids, pix), eng.forward(ids.unsqueeze(0), None, [pix])[0].view(-1))
    assert cache.captures == 3
    # different pixels through the same capture
    pix2 = synth.synth_pixel_values(8, cfg.vision.image_size, seed=6).half().cuda()
    assert torch.equal(cache(ids, pix2), eng.forward(ids.unsqueeze(0), None, [pix2])[0].view(-1))
    assert cache.captures == 3


def test_graph_cache_evicts_least_recently_used():
    from videoitg_amd import synth
    from videoitg_amd.engine import GraphCache
    cfg, eng = tiny_engine()
    cache = GraphCache(eng, bucket=8, max_entries=2)
    pix = synth.synth_pixel_values(4, cfg.vision.image_size, seed=5).half().cuda()
    mk = lambda L: torch.cat([torch.tensor([-200]), synth.synth_prompt_ids(L, cfg.vocab_size)])
    for L in (3, 12, 3, 20, 3):          # buckets 8, 16, 8 (hit), 24 (evicts 16), 8 (hit)
        cache(mk(L), pix)
    assert cache.captures == 3
    cache(mk(12), pix)                   # 16 was evicted -> recapture
    assert cache.captures == 4


def test_model_object_replays_graphs_when_asked(golden_dir):
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.model.grounding_qwen2 import EagleQwenG
    cfg = tiny_config()
    m = EagleQwenG(cfg, synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0), torch.float16, "cuda:0").eval()
    pix = synth.synth_pixel_values(8, cfg.vision.image_size, seed=5)
    ids = torch.tensor([[-200, 5, 6, 7, 8, 9]])
    eager = m(ids, attention_mask=torch.ones_like(ids).bool(), images=[pix.half().cuda()]).logits[0].clone()
    m.use_graphs = True
    for _ in range(2):
        out = m(ids, attention_mask=torch.ones_like(ids).bool(), images=[pix.half().cuda()]).logits
        assert out[0].shape == (8, 1) and torch.equal(out[0], eager)
    assert m._graphs.captures == 1


def test_full_size_graph_replay_equals_eager():
    """the captured forward at FULL depth and width (26 + 28 layers, 64 frames x 384^2 -> N = 16 445 with a 60-token prompt in a
    64-token bucket): replay == eager, bit for bit, and a second video through the same capture too"""
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig
    from videoitg_amd.engine import GraphCache, SelectorEngine
    from videoitg_amd.synth_device import packed_random_weights
    cfg = ITGConfig()
    eng = SelectorEngine(packed_random_weights(cfg, torch.bfloat16, "cuda:0", seed=0))
    cache = GraphCache(eng, bucket=64)
    for seed, L in ((1234, 60), (99, 41)):
        ids, pix = synth.workload_inputs(cfg, 64, 384, L, pix_seed=seed)
        pix = pix.bfloat16().cuda()
        want = eng.forward(ids, None, [pix])[0].view(-1).clone()
        got = cache(ids.view(-1), pix)
        assert torch.isfinite(got).all() and torch.equal(got, want), (got - want).abs().max().item()
    assert cache.captures == 1
    eng.release_workspace()


def test_batch_of_eight_videos_through_one_capture_at_the_headline_workload(golden_dir):
    """BASELINE.json configs[3] at its own size: 8 different 512-frame x 448^2 videos - other pixels, prompts of other
    lengths inside one 64-token bucket - through ONE captured forward (one video per GPU is the reference's data parallelism: whole documents per
    rank, lmms_eval/utils.py:689-695, lmms_eval/models/videoitg.py:274-328).  Each replay equals the eager forward of the same document bit for
    bit, there is one capture, and the fixture video (tests/golden/full_c2_512x448.json: a complete fp32 oracle pass) comes out of the graph
    within north_star's 1e-3 with the oracle's Top-32."""
    import json
    import os
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig
    from videoitg_amd.engine import GraphCache, SelectorEngine
    from videoitg_amd.synth_device import packed_random_weights
    from videoitg_amd.torch_ops import K
    rec = json.load(open(os.path.join(golden_dir, "full_c2_512x448.json")))
    cfg = ITGConfig()
    eng = SelectorEngine(packed_random_weights(cfg, torch.float16, "cuda:0", seed=rec["weight_seed"]))
    cache = GraphCache(eng, bucket=64)
    videos = [(rec["pixel_seed"], rec["prompt_len"])] + [(5000 + 17 * i, L) for i, L in enumerate((33, 64, 41, 57, 36, 49, 62))]
    seen = []
    for vi, (seed, L) in enumerate(videos):
        if vi == 0:                                      # the fixture video: the oracle's own inputs
            ids, pix = synth.workload_inputs(cfg, rec["frames"], rec["resolution"], L, pix_seed=seed)
            pix = pix.half().cuda()
        else:                                            # seven other documents: own prompt, own pixels drawn on the device (the seeded low-pass CPU
            ids = torch.cat([torch.tensor([-200]), synth.synth_prompt_ids(L, cfg.vocab_size, seed=seed)]).unsqueeze(0)      # generator cost 3 s per video)
            g = torch.Generator(device="cuda").manual_seed(seed)
            pix = torch.empty((rec["frames"], 3, rec["resolution"], rec["resolution"]), device="cuda").uniform_(-1.0, 1.0, generator=g).half()
        want = eng.forward(ids, None, [pix])[0].view(-1).clone()
        got = cache(ids.view(-1), pix).clone()
        assert torch.isfinite(got).all() and torch.equal(got, want), (seed, L, (got - want).abs().max().item())
        seen.append(got.float().cpu())
        del pix
    assert cache.captures == 1
    for a in range(len(seen)):                                            # eight different documents, not one replayed eight times
        for b in range(a + 1, len(seen)):
            assert not torch.equal(seen[a], seen[b])
    want = torch.tensor(rec["logits"])
    ds = (torch.sigmoid(seen[0]) - torch.sigmoid(want)).abs().max().item()
    _, order = K.rank(seen[0].cuda())
    print(f"C4 fixture video out of the graph: max|dscore|={ds:.2e}")
    assert ds <= 1e-3 and sorted(order.cpu().tolist()[:rec["k"]]) == rec["topk"]
    eng.release_workspace()


def test_workspace_is_bounded_over_varying_prompts_and_frame_counts():
    """one buffer per name, grown to the largest shape seen - memory after a sweep over prompt lengths and frame
    counts equals the memory of the largest configuration alone (no per-shape accumulation)."""
    from videoitg_amd import synth
    cfg, eng = tiny_engine()
    R = cfg.vision.image_size

    def run(T, L):
        ids = torch.cat([torch.tensor([-200]), synth.synth_prompt_ids(L, cfg.vocab_size)]).unsqueeze(0)
        return eng.forward(ids, None, [synth.synth_pixel_values(T, R, seed=T).half().cuda()])[0]
    first = run(5, 9).clone()
    sweep = [(3, 4), (12, 50), (7, 33), (1, 1), (12, 49), (9, 2), (5, 9)]
    for T, L in sweep:
        out = run(T, L)
    high = eng.workspace_bytes()
    n_buffers = len(eng.workspace_tensors())
    for _ in range(3):                        # the same documents again, any order: nothing grows, no new buffers appear
        for T, L in reversed(sweep):
            out = run(T, L)
    assert eng.workspace_bytes() == high and len(eng.workspace_tensors()) == n_buffers
    # ... and the high-water mark is what the per-name maxima need, not the sum over the distinct shapes seen
    alone = 0
    for T, L in sweep:
        eng.release_workspace()
        run(T, L)
        alone = max(alone, eng.workspace_bytes())
    assert high <= 2 * alone
    for T, L in sweep:                        # grow everything again, then the first document once more
        run(T, L)
    out = run(5, 9)
    assert torch.equal(out, first)           # views of larger buffers (row stride unchanged) give the same bits


def test_torch_ops_are_registered_and_equal_the_raw_binding():
    from videoitg_amd import ops, torch_ops
    for name in torch_ops.OP_NAMES:
        assert hasattr(torch.ops.videoitg, name), name
    a = (torch.randn(300, 192, device="cuda") * 0.5).half()
    w = (torch.randn(264, 192, device="cuda") * 0.1).half()
    bias = torch.randn(264, device="cuda")
    o1 = torch.empty(300, 264, dtype=torch.float16, device="cuda")
    o2 = torch.empty_like(o1)
    ops.gemm(a, w, o1, mode=ops.GEMM_STORE16, bias=bias, act=ops.ACT_GELU_TANH)
    torch.ops.videoitg.gemm(a, w, o2, ops.GEMM_STORE16, None, bias, ops.ACT_GELU_TANH)
    assert torch.equal(o1, o2)
    x = torch.randn(50, 256, device="cuda")
    g = torch.ones(256, device="cuda")
    y1 = torch.empty(50, 256, dtype=torch.float16, device="cuda")
    y2 = torch.empty_like(y1)
    ops.rmsnorm(x, g, y1, 256, 1e-6)
    torch.ops.videoitg.rmsnorm(x, g, y2, 256, 1e-6)
    assert torch.equal(y1, y2)
    s1, i1 = ops.rank(x[0])
    s2, i2 = torch.ops.videoitg.rank(x[0])
    assert torch.equal(s1, s2) and torch.equal(i1, i2)
    with pytest.raises(NotImplementedError):
        torch.ops.videoitg.rmsnorm(x.cpu(), g.cpu(), y1.cpu(), 256, 1e-6)      # no CPU kernel is registered: loud failure
    with pytest.raises(RuntimeError):
        torch.ops.videoitg.gemm(a[:, :100].contiguous(), w[:, :100].contiguous(), o2, ops.GEMM_STORE16)   # K % 64 != 0: VITG_ERR_SHAPE -> RuntimeError


def test_model_survives_accelerate_prepare_model():
    """lmms_eval/models/videoitg.py:193-205 wraps the model with accelerator.prepare_model(model, evaluation_mode=True)"""
    from accelerate import Accelerator
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.model.grounding_qwen2 import EagleQwenG
    cfg = tiny_config()
    m = EagleQwenG(cfg, synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0), torch.float16, "cuda:0").eval()
    pix = synth.synth_pixel_values(8, cfg.vision.image_size, seed=5).half().cuda()
    ids = torch.tensor([[-200, 5, 6, 7, 8, 9]]).cuda()
    want = m(ids, attention_mask=ids.ne(0), images=[pix]).logits[0].clone()
    acc = Accelerator()
    wrapped = acc.prepare_model(m, evaluation_mode=True)
    wrapped.eval()
    wrapped.tie_weights()
    with torch.inference_mode():
        got = wrapped(ids, attention_mask=ids.ne(0), images=[pix]).logits[0]
    assert torch.equal(got.float(), want.float())
    assert wrapped.config is cfg and wrapped.get_vision_tower().image_processor is not None
