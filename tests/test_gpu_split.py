"""-m gpu: the operand-split precision tier (videoitg_amd/csrc/split.hip, SelectorEngine(operand_split=True)).

Every MFMA operand the default path rounds to one 16-bit value is carried as an fp16 pair x = hi + lo (22 bits).  Bounds below were WRITTEN BEFORE
THE FIRST GPU RUN of this tier: the kernels are held to fp32-grade errors against fp64 torch references (a few 1e-6 relative - 2^-22 operand
splitting plus fp32 accumulation), the whole forward to 2e-4 in logit on the reference's own golden fixtures and to north_star's 1e-3 in score WITH the
Top-K set on the heavy-tailed complete-pass fixtures the 16-bit tiers miss (tests/test_fullsize_parity.py: HEAVY)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
KERNEL_REL = 4e-6          # |got - ref| <= KERNEL_REL * (scale of the row) : fp32-grade, ~100x below one fp16 rounding (4.9e-4)


def split_pair(x32: torch.Tensor, width: int) -> torch.Tensor:
    """fp32 [M, D] -> fp16 [M, 2 * width] = [hi | lo], pad columns zero (the layout of csrc/split.hip, built with torch ops)"""
    M, D = x32.shape
    out = torch.zeros((M, 2 * width), dtype=torch.float16, device=x32.device)
    hi = x32.half()
    out[:, :D] = hi
    out[:, width:width + D] = (x32 - hi.float()).half()
    return out


def joined(t: torch.Tensor, width: int, d: int) -> torch.Tensor:
    return t[:, :d].double() + t[:, width:width + d].double()


def heavy(x: torch.Tensor, cols=(3, 17), gain=700.0) -> torch.Tensor:
    """a few columns two to three orders above the rest - and many values below fp16's smallest normal lo (|x| < 0.125: lo is subnormal)"""
    x = x.clone()
    for c in cols:
        if c < x.shape[1]:
            x[:, c] *= gain
    x[:, 1::5] *= 0.01
    return x


@pytest.mark.parametrize("rms", [False, True])
@pytest.mark.parametrize("M,D", [(37, 144), (1030, 1152), (260, 3584)])
def test_norm_split(M, D, rms):
    from videoitg_amd.torch_ops import K
    torch.manual_seed(M + D)
    x = heavy(torch.randn(M, D, device=DEV))
    w, b = torch.randn(D, device=DEV), torch.randn(D, device=DEV)
    Dp = K.pad64(D)
    out = torch.full((M, 2 * Dp + 8), 7.0, dtype=torch.float16, device=DEV)[:, :2 * Dp]       # a strided view: row stride != width
    K.norm_split(x, w, None if rms else b, out, D, 1e-6)
    xd = x.double()
    ref = (xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-6) * w.double()) if rms else \
        torch.nn.functional.layer_norm(xd, (D,), w.double(), b.double(), 1e-6)
    got = joined(out, Dp, D)
    scale = ref.abs().amax(-1, keepdim=True).clamp_min(1.0)
    assert ((got - ref).abs() / scale).max().item() <= KERNEL_REL
    assert torch.equal(out[:, :D], got.float().half()) or ((out[:, :D].double() - got).abs() <= got.abs() * 2.0 ** -11 + 1e-7).all()   # hi = fp16 of the value
    assert (out[:, D:Dp] == 0).all() and (out[:, Dp + D:] == 0).all()


def test_rope_split():
    from videoitg_amd.torch_ops import K
    torch.manual_seed(3)
    N, heads, hd, extra = 333, 6, 128, 2
    width = (heads + extra) * hd
    x = heavy(torch.randn(N, width, device=DEV))
    buf = split_pair(x, width)
    fr = torch.outer(torch.arange(N, dtype=torch.float32), 1.0 / (1e6 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd)))
    cos, sin = fr.cos().to(DEV).contiguous(), fr.sin().to(DEV).contiguous()
    x0 = joined(buf, width, width)
    K.rope_split_(buf, cos, sin, heads, hd, width)
    got = joined(buf, width, width)
    xr = x0.view(N, heads + extra, hd)
    c, s = cos.double()[:, None, :], sin.double()[:, None, :]
    ref = xr.clone()
    a, b = xr[:, :heads, :hd // 2], xr[:, :heads, hd // 2:]
    ref[:, :heads, :hd // 2] = a * c - b * s
    ref[:, :heads, hd // 2:] = b * c + a * s
    ref = ref.view(N, width)
    assert ((got - ref).abs() / ref.abs().amax(-1, keepdim=True)).max().item() <= KERNEL_REL
    assert torch.equal(got[:, heads * hd:], x0[:, heads * hd:])                     # heads past `heads` (the V heads) untouched


@pytest.mark.parametrize("T,G,HW", [(3, 27, 5), (2, 16, 16), (4, 32, 5), (1, 4, 3)])
def test_bilinear_pool_split(T, G, HW):
    from videoitg_amd.torch_ops import K
    torch.manual_seed(G)
    D = 144
    x = heavy(torch.randn(T * G * G, D, device=DEV))
    Dp = K.pad64(D)
    out = torch.full((T * HW * HW, 2 * Dp), 3.0, dtype=torch.float16, device=DEV)
    K.bilinear_pool_split(x, out, T, G, HW, D)
    xd = x.double().view(T, G, G, D).permute(0, 3, 1, 2)
    ref = xd if HW == G else torch.nn.functional.interpolate(xd, size=(HW, HW), mode="bilinear", align_corners=False)
    ref = ref.permute(0, 2, 3, 1).reshape(T * HW * HW, D)
    got = joined(out, Dp, D)
    scale = ref.abs().amax(-1, keepdim=True).clamp_min(1.0)
    # the interpolation WEIGHTS are fp32 values in the kernel, in ATen's fp32 upsample (what the oracle runs) and in the reference: against the fp64
    # evaluation that costs ~1e-5 of the row's scale; against the fp32 evaluation of the same expression the split output is exact to fp32 rounding
    assert ((got - ref).abs() / scale).max().item() <= 5e-5
    x32 = x.view(T, G, G, D).permute(0, 3, 1, 2)
    ref32 = (x32 if HW == G else torch.nn.functional.interpolate(x32, size=(HW, HW), mode="bilinear", align_corners=False)).permute(0, 2, 3, 1).reshape(T * HW * HW, D)
    assert ((got - ref32.double()).abs() / scale).max().item() <= 1e-6
    assert (out[:, D:Dp] == 0).all() and (out[:, Dp + D:] == 0).all()


def _act(v, act):
    if act == 1:
        return 0.5 * v * (1.0 + torch.tanh(0.7978845608028654 * (v + 0.044715 * v ** 3)))
    if act == 2:
        return torch.nn.functional.gelu(v)
    return v


@pytest.mark.parametrize("M,N,K_,act,bias", [(300, 384, 192, 0, True), (1024, 768, 320, 1, True), (520, 256, 128, 2, True), (77, 1160, 64, 0, False),
                                             (2048, 4352, 1152, 1, True)])
def test_gemm_split16_store(M, N, K_, act, bias):
    """split A operand (K-concatenated against [W | W]) -> (hi, lo) output pairs, against fp64"""
    from videoitg_amd.torch_ops import K
    torch.manual_seed(M + N)
    a = heavy(torch.randn(M, K_, device=DEV) * 0.7)
    w = (torch.randn(N, K_, device=DEV) * K_ ** -0.5).half()
    b = torch.randn(N, device=DEV) if bias else None
    a2, w2 = split_pair(a, K_), torch.cat([w, w], 1).contiguous()
    Np = K.pad64(N)
    out = torch.full((M, 2 * Np), 5.0, dtype=torch.float16, device=DEV)
    K.gemm_split16(a2, w2, out, mode=K.GEMM_STORE16, lo_off=Np, bias=b, act=act)
    a_eff = joined(a2, K_, K_)                                  # the operand the kernel really saw (22-bit a)
    pre = a_eff @ w.double().t() + (b.double() if bias else 0.0)
    ref = _act(pre, act)
    got = joined(out, Np, N)
    scale = (a_eff.abs() @ w.double().abs().t() + 1.0)          # the size of the sum of |terms|: what fp32 accumulation error scales with
    err = ((got - ref).abs() / scale).max().item()
    assert err <= KERNEL_REL, err
    # and it is fp32-grade where one 16-bit operand is not: the same product from rounded-once operands is >= 20x further from fp64
    once = _act(a.half().double() @ w.double().t() + (b.double() if bias else 0.0), act)
    ref32 = _act(a.double() @ w.double().t() + (b.double() if bias else 0.0), act)
    assert (joined(out, Np, N) - ref32).abs().max() * 20 <= (once - ref32).abs().max() + 1e-9
    assert (out[:, N:Np] == 5.0).all() or Np == N              # columns outside the two planes are not written


def test_gemm_split16_swiglu_and_f32_consumer():
    """gate/up -> silu(gate) * up as (hi, lo) pairs, consumed by the fp32-residual GEMM through the ordinary vitg_gemm with K = 2 Kp"""
    from videoitg_amd.torch_ops import K
    torch.manual_seed(11)
    M, D, F = 700, 256, 640
    x = heavy(torch.randn(M, D, device=DEV), gain=40.0)            # silu(gate) * up has to stay inside fp16's range, as on the default path
    g, u = (torch.randn(F, D, device=DEV) * D ** -0.5).half(), (torch.randn(F, D, device=DEV) * D ** -0.5).half()
    gu = torch.cat([g.view(F // 32, 1, 32, D), u.view(F // 32, 1, 32, D)], 1).reshape(2 * F, D)       # rows [32 gate | 32 up] (engine.pack_llm_layer)
    down = (torch.randn(D, F, device=DEV) * F ** -0.5).half()
    x2 = split_pair(x, D)
    f2 = torch.zeros((M, 2 * F), dtype=torch.float16, device=DEV)
    K.gemm_split16(x2, torch.cat([gu, gu], 1).contiguous(), f2, mode=K.GEMM_SWIGLU, lo_off=F)
    xe = joined(x2, D, D)
    fref = torch.nn.functional.silu(xe @ g.double().t()) * (xe @ u.double().t())
    fgot = joined(f2, F, F)
    assert ((fgot - fref).abs() / fref.abs().amax(-1, keepdim=True)).max().item() <= 2e-6
    res = torch.randn(M, D, device=DEV)
    out = res.clone()
    K.gemm(f2, torch.cat([down, down], 1).contiguous(), out, mode=K.GEMM_F32, res=out)
    ref = res.double() + fgot @ down.double().t()
    assert ((out.double() - ref).abs() / ref.abs().amax(-1, keepdim=True)).max().item() <= 2e-6


def _attn_ref(q, k, v, scale):
    rep = q.shape[1] // k.shape[1]
    k, v = k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1)
    p = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * scale, -1)
    return torch.matmul(p, v)


@pytest.mark.parametrize("hd,B,H,KV,S,Sk", [(72, 3, 4, 4, 300, 300), (72, 2, 16, 16, 1024, 1024), (128, 1, 4, 2, 333, 333), (128, 1, 28, 4, 700, 1500),
                                             (128, 1, 2, 1, 61, 61)])
def test_attention_split(hd, B, H, KV, S, Sk):
    """3-term QK^T, split P, 3-term PV against an fp64 softmax attention on the operands the kernel saw; GQA, ragged lengths, keys != queries"""
    from videoitg_amd.torch_ops import K
    torch.manual_seed(hd + S)
    wq, wk = H * hd, KV * hd
    # (scores stay below ~10^2: at |s| ~ 10^4 - two 700x columns meeting - ANY fp32 softmax is 1e-3 off the fp64 one, and so is this one)
    q = heavy(torch.randn(B * S, wq, device=DEV), cols=(5, hd + 9), gain=12.0) * 0.5
    k = heavy(torch.randn(B * Sk, wk, device=DEV), cols=(5, 40), gain=12.0) * 0.5
    v = heavy(torch.randn(B * Sk, wk, device=DEV), cols=(7,))
    q2, k2, v2 = split_pair(q, wq), split_pair(k, wk), split_pair(v, wk)
    op = K.pad64(wq)
    o2 = torch.zeros((B * S, 2 * op), dtype=torch.float16, device=DEV)
    scale = hd ** -0.5
    K.attention_split(q2, k2, v2, o2, lo=(wq, wk, wk, op), batch=B, heads=H, kv_heads=KV, seq=S, kv_seq=Sk, head_dim=hd,
                      q_strides=(S * 2 * wq, 2 * wq, hd), k_strides=(Sk * 2 * wk, 2 * wk, hd), v_strides=(Sk * 2 * wk, 2 * wk, hd),
                      o_strides=(S * 2 * op, 2 * op, hd), scale=scale)
    qe = joined(q2, wq, wq).view(B, S, H, hd).transpose(1, 2)
    ke = joined(k2, wk, wk).view(B, Sk, KV, hd).transpose(1, 2)
    ve = joined(v2, wk, wk).view(B, Sk, KV, hd).transpose(1, 2)
    ref = _attn_ref(qe, ke, ve, scale).transpose(1, 2).reshape(B * S, wq)
    got = joined(o2, op, wq)
    err = ((got - ref).abs() / ve.abs().amax()).max().item()
    assert err <= 1e-5, err
    # the 16-bit kernel on the same (rounded-once) operands, for scale: it has to be far worse, or the test proves nothing
    o16 = torch.zeros((B * S, op), dtype=torch.float16, device=DEV)
    K.attention(q.half(), k.half(), v.half(), o16, batch=B, heads=H, kv_heads=KV, seq=S, kv_seq=Sk, head_dim=hd,
                q_strides=(S * wq, wq, hd), k_strides=(Sk * wk, wk, hd), v_strides=(Sk * wk, wk, hd), o_strides=(S * op, op, hd), scale=scale)
    ref32 = _attn_ref(q.double().view(B, S, H, hd).transpose(1, 2), k.double().view(B, Sk, KV, hd).transpose(1, 2),
                      v.double().view(B, Sk, KV, hd).transpose(1, 2), scale).transpose(1, 2).reshape(B * S, wq)
    e16 = (o16[:, :wq].double() - ref32).abs().max().item()
    e22 = (got - ref32).abs().max().item()
    assert e22 * 20 <= e16, (e22, e16)


# ---- the whole forward ----------------------------------------------------------------------------------------------------------------------------
GOLDEN = ["pool_post", "nopool_prepost", "oddgrid", "batch2_pad", "chunk520", "ragged_image", "two_images", "two_images_ragged", "heavy_pool_post"]
GOLDEN_LOGIT_TOL = 2e-4        # on fixtures the REFERENCE ITSELF produced (tests/golden/make_golden.py); the fp16 default sits at 1e-3 .. 1e-2 in logit
# History: the first build of the tier MISSED this bound on chunk520 (4.65e-4, gpurun_out/r06_c4) - which is what exposed the compiler-contraction bug
# described at ELEM_REL below (one output element in ~10^4 off by an fp16 ulp).  With the values pinned every fixture is at 2e-6 .. 1.3e-5
# (gpurun_out/r06_c9), the level at which the oracle itself agrees with the reference (<= 9e-6, tests/test_oracle_golden.py).
GOLDEN_LOGIT_TOL_CASE = {}
GOLDEN_SCORE_TOL = 2e-5


def split_engine(cfg, sd, vsd):
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    pw = PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, DEV, vit_q_prescaled=False)
    return SelectorEngine(pw, operand_split=True)


@pytest.mark.parametrize("name", GOLDEN)
def test_split_tier_matches_reference_golden(golden_dir, name):
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig
    z = np.load(os.path.join(golden_dir, f"{name}.npz"))
    cfg = ITGConfig.from_dict(json.loads(str(z["cfg"])))
    seed = int(z["seed"])
    T = z["T"].tolist()
    images = [synth.synth_pixel_values(t, cfg.vision.image_size, seed=1234 + i).half().to(DEV) for i, t in enumerate(T)]
    sd, vsd = synth.synth_llm_state_dict(cfg, seed), synth.synth_vision_state_dict(cfg, seed)
    if "heavy_tail" in z:                              # heavy_pool_post: the REFERENCE's forward on weights with massive-activation channels
        plan = synth.HeavyTailPlan.from_json(json.loads(str(z["heavy_tail"])))
        sd, vsd = plan.apply(sd), plan.apply(vsd)
    eng = split_engine(cfg, sd, vsd)
    got = eng.forward(torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]), images)
    torch.cuda.synchronize()
    worst = worst_s = 0.0
    assert len(got) == len(T) or len(got) >= 1
    for b, g in enumerate(got):                       # one logits tensor per SAMPLE
        want = torch.from_numpy(z[f"logits{b}"]).float().view(-1)
        assert g.numel() == want.numel() and torch.isfinite(g).all()
        g = g.float().cpu().view(-1)
        worst = max(worst, (g - want).abs().max().item())
        worst_s = max(worst_s, (torch.sigmoid(g) - torch.sigmoid(want)).abs().max().item())
    print(f"operand-split tier vs the reference's golden logits, {name}: max |dlogit| = {worst:.2e}, max |dscore| = {worst_s:.2e}")
    assert worst <= GOLDEN_LOGIT_TOL_CASE.get(name, GOLDEN_LOGIT_TOL) and worst_s <= GOLDEN_SCORE_TOL, (name, worst, worst_s)


def _full_case(golden_dir, name):
    path = os.path.join(golden_dir, f"full_{name}.json")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    return json.load(open(path))


# (max |dscore|, max |dlogit|) STATED before the first run: north_star's 1e-3 with a 10x margin, and the logit bound that implies it everywhere
FULL = {"depth16o_224": (1e-4, 2e-3), "c2o_512x448": (1e-4, 4e-3), "c2_512x448": (1e-4, 2e-3), "c0_64x224": (1e-4, 2e-3)}     # (c0: the no-pool branch, bound stated before its first run)


@pytest.mark.parametrize("name", list(FULL))
def test_split_tier_holds_north_star_on_heavy_tailed_weights(golden_dir, name):
    """<= 1e-3 in score AND the oracle's Top-K set on the heavy-tailed complete-pass fixtures (c2o: the headline workload with
    massive-activation channels, where the fp16 default measures 2.9e-3 / 31 of 32 and the oracle's own rank-32/33 gap is 1.5e-4), through the tier
    that splits every MFMA operand.  The benign headline fixture runs through the same tier for scale."""
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig
    from videoitg_amd.engine import SelectorEngine
    from videoitg_amd.synth_device import packed_random_weights
    from videoitg_amd.torch_ops import K
    rec = _full_case(golden_dir, name)
    cfg = ITGConfig()
    eng = SelectorEngine(packed_random_weights(cfg, torch.float16, DEV, seed=rec["weight_seed"], plan=synth.fixture_plan(rec), vit_q_prescaled=False),
                         operand_split=True)
    ids, pix = synth.fixture_inputs(cfg, rec)
    got = eng.forward(ids, None, [pix.half().to(DEV)])[0].view(-1)
    torch.cuda.synchronize()
    _, order = K.rank(got)
    order, got = order.cpu().tolist(), got.float().cpu()
    want = torch.tensor(rec["logits"])
    ds, dl = (torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item(), (got - want).abs().max().item()
    k = rec["k"]
    print(f"operand-split tier, full {name}: max|dscore|={ds:.2e} max|dlogit|={dl:.2e} top{k} kept {len(set(order[:k]) & set(rec['topk']))} "
          f"(oracle rank-{k}/{k + 1} gap {rec['gap_k_k1']})")
    tol_s, tol_l = FULL[name]
    assert ds <= tol_s and dl <= tol_l, (name, ds, dl)
    assert ds <= 1e-3                                                        # north_star, for the record
    if rec["gap_k_k1"] is None or rec["gap_k_k1"] > 2 * ds:
        assert sorted(order[:k]) == rec["topk"], name
    assert order[:8] == rec["order"][:8] or ds > 0                           # (reported above; the set assertion is the gate)
    eng.release_workspace()


def test_split_tier_refuses_what_it_does_not_cover():
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    cfg = tiny_config()
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    with pytest.raises(ValueError, match="vit_q_prescaled"):
        SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, DEV, vit_q_prescaled=True), operand_split=True)
    with pytest.raises(ValueError, match="float16"):
        SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.bfloat16, DEV, vit_q_prescaled=False), operand_split=True)
    with pytest.raises(ValueError, match="fp8"):
        SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, DEV, fp8="llm", vit_q_prescaled=False), operand_split=True)
    eng = split_engine(cfg, sd, vsd)
    with pytest.raises(NotImplementedError):
        eng.llm_forward(torch.zeros(8, cfg.hidden_size, device=DEV), kv_len=torch.tensor([4], dtype=torch.int32, device=DEV))


# ---- elementwise precision on BENIGN data -----------------------------------------------------------------------------------------------------------
# The heavy columns above make the error SCALE large, so they cannot see an fp16-sized error in a single ordinary element.  That is exactly what the first
# build of this tier had: hipcc (-ffp-contract=fast) fused the multiply that produces a value into its two uses differently - the stored hi was
# cvt(fl32(a * b)), the hi that lo was taken against v_fma_mix_f16(a, b) - and near a tie (hi, lo) summed to x +- ulp(hi): one element in ~10^4 off by
# exactly 2^-13 .. 2^-17 (tools/r06/split_attn_debug.py).  Every output ELEMENT is therefore held to 2e-5 of the output's rms (the outliers were 1e-3).
ELEM_REL = 2e-5


def _elementwise_ok(got, ref):
    e = (got - ref).abs().max().item()
    # (floor: an unscaled lo below 2^-14 is a subnormal with a quantum of 2^-24 - outputs of overall scale ~1e-3 are exact to that, not to 22 bits)
    return e <= max(ELEM_REL * ref.pow(2).mean().sqrt().item(), 1.2e-7), e


@pytest.mark.parametrize("sc", [1.0, 0.05])
def test_every_output_element_is_fp32_grade_on_benign_data(sc):
    from videoitg_amd.torch_ops import K
    torch.manual_seed(5)
    M, D = 1000, 1152
    x = torch.randn(M, D, device=DEV) * sc
    w, b = torch.randn(D, device=DEV), torch.randn(D, device=DEV)
    out = torch.zeros(M, 2 * D, dtype=torch.float16, device=DEV)
    K.norm_split(x, w, b, out, D, 1e-6)
    ok, e = _elementwise_ok(joined(out, D, D), torch.nn.functional.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-6))
    assert ok, ("layernorm", e)
    Kd, N = 384, 768
    a = torch.randn(M, Kd, device=DEV) * sc
    wt = (torch.randn(N, Kd, device=DEV) * Kd ** -0.5).half()
    bias = torch.randn(N, device=DEV) * sc
    a2 = split_pair(a, Kd)
    o2 = torch.zeros(M, 2 * N, dtype=torch.float16, device=DEV)
    for act in (0, 1, 2):
        K.gemm_split16(a2, torch.cat([wt, wt], 1).contiguous(), o2, mode=K.GEMM_STORE16, lo_off=N, bias=bias, act=act)
        ok, e = _elementwise_ok(joined(o2, N, N), _act(joined(a2, Kd, Kd) @ wt.double().t() + bias.double(), act))
        assert ok, ("gemm_split16", act, e)
    for hd, H, KV, S in ((72, 2, 2, 16), (72, 4, 4, 1024), (128, 4, 2, 523), (128, 1, 1, 128), (128, 2, 1, 192)):
        wq, wk = H * hd, KV * hd
        q, k, v = torch.randn(S, wq, device=DEV) * sc, torch.randn(S, wk, device=DEV) * sc, torch.randn(S, wk, device=DEV) * sc
        q2, k2, v2 = split_pair(q, wq), split_pair(k, wk), split_pair(v, wk)
        op = K.pad64(wq)
        ob = torch.zeros(S, 2 * op, dtype=torch.float16, device=DEV)
        K.attention_split(q2, k2, v2, ob, lo=(wq, wk, wk, op), batch=1, heads=H, kv_heads=KV, seq=S, head_dim=hd, q_strides=(0, 2 * wq, hd),
                          k_strides=(0, 2 * wk, hd), v_strides=(0, 2 * wk, hd), o_strides=(0, 2 * op, hd), scale=hd ** -0.5)
        ref = _attn_ref(joined(q2, wq, wq).view(1, S, H, hd).transpose(1, 2), joined(k2, wk, wk).view(1, S, KV, hd).transpose(1, 2),
                        joined(v2, wk, wk).view(1, S, KV, hd).transpose(1, 2), hd ** -0.5).transpose(1, 2).reshape(S, wq)
        ok, e = _elementwise_ok(joined(ob, op, wq), ref)
        assert ok, ("attention_split", hd, S, e)


def test_operand_split_through_the_operator_surface():
    """the tier is ONE switch on the reference-side surface: load_pretrained_model(..., operand_split=True) -> model(...).logits as infer.py:15-21,60-72 uses
    it, against the oracle on the same weights and inputs; the default model on the same call is the fp16 tier (further away)"""
    from oracle import videoitg_oracle as O                          # checker
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.model.builder import load_pretrained_model
    cfg = tiny_config()
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    pix = synth.synth_pixel_values(8, cfg.vision.image_size, seed=1234)
    ids = torch.tensor([[-200, 5, 6, 7, 8, 9]])
    want = O.forward(cfg, sd, vsd, ids, None, [pix])[0].view(-1)
    errs = {}
    for split in (True, False):
        tok, model, proc, ctx = load_pretrained_model("synthetic:tiny:0", None, "VideoITG-tiny", device_map="cuda:0", operand_split=split)
        model = model.half().eval()
        assert model.operand_split is split
        got = model(ids, attention_mask=torch.ones_like(ids).bool(), images=[pix.half().to(DEV)]).logits[0].view(-1).float().cpu()
        assert model.engine().operand_split is split
        errs[split] = (got - want).abs().max().item()
    print(f"operator surface: operand_split max|dlogit| = {errs[True]:.2e}, fp16 default {errs[False]:.2e}")
    assert errs[True] <= 5e-5 and errs[True] * 10 <= errs[False]


def test_bilinear_pool_split_on_compact_rows_and_the_compact_last_layer():
    """the split tier skips the last tower layer's dead rows like the default path: the pool then reads COMPACT rows through cmap.  (a) pool on compact
    input == pool on dense input, bit for bit; (b) the engine with and without skip_dead_rows agrees to fp32 rounding"""
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.torch_ops import K
    torch.manual_seed(9)
    T, G, HW, D = 3, 27, 5, 144
    cfg = tiny_config()
    eng = split_engine(cfg, synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0))
    idx, cmap = eng.pool_sources(G, HW)
    Pc = idx.numel()
    x = torch.randn(T * G * G, D, device=DEV)
    xc = x.view(T, G * G, D)[:, idx.long()].reshape(T * Pc, D).contiguous()
    Dp = K.pad64(D)
    a, b = torch.zeros((T * HW * HW, 2 * Dp), dtype=torch.float16, device=DEV), torch.zeros((T * HW * HW, 2 * Dp), dtype=torch.float16, device=DEV)
    K.bilinear_pool_split(x, a, T, G, HW, D)
    K.bilinear_pool_split(xc, b, T, G, HW, D, cmap, Pc)
    assert Pc in (81, 100) and torch.equal(a, b)              # (9 or 10 source rows per axis: the 13.0 coordinate lists row 14 with weight 0)
    pix = synth.synth_pixel_values(6, cfg.vision.image_size, seed=3).half().to(DEV)
    ids = torch.tensor([[-200, 5, 6, 7]])
    outs = []
    for skip in (True, False):
        eng.skip_dead_rows = skip
        outs.append(eng.forward(ids, None, [pix])[0].view(-1).float().cpu())
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-5


def test_operand_split_tier_context_parallel_on_one_gpu(cp_toy):
    """BASELINE configs[2] in the operand-split tier: two ranks share cuda:0 over gloo, frame-sharded tower (compact last layer, GLOBAL pooled grid),
    K|V exchanged as (hi, lo) PAIRS through the blocking all-gather, 3-term attention of the local queries against all keys - against the one-rank forward of
    the same tier (same arithmetic per token: fp32 rounding apart) and, through it, the oracle (ragged: 5 + 4 frames, prompt tokens on both ends; the run
    shares its rank processes with the other toy context-parallel cases: conftest cp_toy)"""
    got, want = cp_toy["split"]
    assert got.shape == want.shape == (9,) and torch.isfinite(got).all()
    d = (got - want).abs().max().item()
    print(f"operand-split tier, context-parallel x2: max|dlogit| vs one rank = {d:.2e}")
    assert d <= 2e-5, d


def test_heavy_tailed_weights_against_the_reference_itself(golden_dir):
    """tests/golden/heavy_pool_post.npz is the REFERENCE's own fp32 forward (tests/golden/make_golden.py) on toy-size weights with massive-activation
    channels (|x| max 989 in the tower stream, 1 328 in the decoder's).  CPU rounding models against it (oracle/quant_model.py): the product's 16-bit sites
    1.9e-3 in logit / 4.7e-4 in score, the reference's OWN model.half() arithmetic 5.8e-3 / 1.3e-3 - it misses north_star's 1e-3 against its own fp32 path.
    Measured here: the fp16 default inside 1e-3 in score at this size, the operand-split tier >= 50x closer (bounds: 2e-4 logit as for every golden)."""
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    z = np.load(os.path.join(golden_dir, "heavy_pool_post.npz"))
    cfg = ITGConfig.from_dict(json.loads(str(z["cfg"])))
    plan = synth.HeavyTailPlan.from_json(json.loads(str(z["heavy_tail"])))
    sd, vsd = plan.apply(synth.synth_llm_state_dict(cfg, 0)), plan.apply(synth.synth_vision_state_dict(cfg, 0))
    pix = synth.synth_pixel_values(int(z["T"][0]), cfg.vision.image_size, seed=1234).half().to(DEV)
    ids, mask = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    want = torch.from_numpy(z["logits0"]).float().view(-1)
    res = {}
    for tier, eng in (("fp16", SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, DEV))), ("split", split_engine(cfg, sd, vsd))):
        got = eng.forward(ids, mask, [pix])[0].view(-1).float().cpu()
        res[tier] = ((got - want).abs().max().item(), (torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item())
    print(f"heavy-tailed reference golden: fp16 default dlogit {res['fp16'][0]:.2e} dscore {res['fp16'][1]:.2e}; operand-split dlogit {res['split'][0]:.2e} "
          f"dscore {res['split'][1]:.2e}")
    assert res["fp16"][1] <= 1e-3
    assert res["split"][0] <= 2e-4 and res["split"][0] * 50 <= res["fp16"][0]


def test_split_entries_refuse_bad_arguments():
    """C-ABI argument checks of the round-6 entry points: nothing is launched on a bad shape / alignment (VITG_ERR_* -> VitgError)"""
    from videoitg_amd._lib import VitgError
    from videoitg_amd.torch_ops import K
    x = torch.randn(8, 144, device=DEV)
    w = torch.ones(144, device=DEV)
    out = torch.zeros(8, 2 * 192, dtype=torch.float16, device=DEV)
    K.norm_split(x, w, None, out, 144, 1e-6)                                   # fine
    with pytest.raises((VitgError, RuntimeError)):
        K.norm_split(x, w, None, torch.zeros(8, 200, dtype=torch.float16, device=DEV), 144, 1e-6)          # two planes of 144 do not fit 200 columns
    with pytest.raises((VitgError, RuntimeError)):
        K.norm_split(x.view(-1)[1:1 + 7 * 144].view(7, 144), w, None, out[:7], 144, 1e-6)                  # rows not 16-B aligned
    a2 = torch.zeros(64, 128, dtype=torch.float16, device=DEV)
    w2 = torch.zeros(256, 128, dtype=torch.float16, device=DEV)
    o2 = torch.zeros(64, 512, dtype=torch.float16, device=DEV)
    K.gemm_split16(a2, w2, o2, mode=K.GEMM_STORE16, lo_off=256)               # fine
    with pytest.raises((VitgError, RuntimeError)):
        K.gemm_split16(a2, w2, o2, mode=K.GEMM_STORE16, lo_off=128)           # planes would overlap
    with pytest.raises((VitgError, RuntimeError)):
        K.gemm_split16(a2, w2, o2, mode=K.GEMM_F32, lo_off=256)               # pairs are a 16-bit-output form
    q = torch.zeros(64, 2 * 3 * 128, dtype=torch.float16, device=DEV)
    o = torch.zeros(64, 2 * 128, dtype=torch.float16, device=DEV)
    kw = dict(batch=1, heads=1, kv_heads=1, seq=64, q_strides=(0, 768, 128), k_strides=(0, 768, 128), v_strides=(0, 768, 128), o_strides=(0, 256, 128), scale=0.1)
    K.attention_split(q, q[:, 128:], q[:, 256:], o, lo=(384, 384, 384, 128), head_dim=128, **kw)       # fine
    with pytest.raises((VitgError, RuntimeError)):
        K.attention_split(q, q[:, 128:], q[:, 256:], o, lo=(384, 384, 384, 128), head_dim=64, **kw)    # head dims 72 and 128 only
    with pytest.raises((VitgError, RuntimeError)):
        K.attention_split(q, q[:, 128:], q[:, 256:], o, lo=(383, 384, 384, 128), head_dim=128, **kw)   # lo plane offset not 16-B aligned
