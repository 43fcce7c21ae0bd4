"""Labelled-evaluation branch of the selector forward (/root/reference/eagle/model/language_model/
grounding_qwen2.py:159-181): BCE-with-logits with pos_weight = min(5, sqrt(neg / max(1, pos)))."""
import pytest
import torch

from oracle import videoitg_oracle as O

CASES = [(512, 0.06, 3.0), (64, 0.5, 1.0), (8, 0.0, 2.0), (200, 1.0, 2.0), (300, 0.004, 6.0), (1, 1.0, 1.0), (8192, 0.1, 4.0)]


def _data(n, p, scale, seed=0):
    g = torch.Generator().manual_seed(seed + n)
    return torch.randn(n, generator=g) * scale, (torch.rand(n, generator=g) < p).float()


def _reference_call(x, y):
    """the third-party op the reference calls, with its weight rule"""
    pos = y.sum()
    neg = y.numel() - pos
    pw = torch.sqrt(neg / max(1, pos))
    return torch.nn.BCEWithLogitsLoss(pos_weight=torch.min(torch.tensor(5.0), pw))(x, y)


@pytest.mark.parametrize("n,p,scale", CASES)
def test_oracle_is_torch_bce_with_capped_pos_weight(n, p, scale):
    x, y = _data(n, p, scale)
    assert abs(O.grounding_loss(x, y).item() - _reference_call(x, y).item()) <= 1e-6 * max(1.0, abs(_reference_call(x, y).item()))


def test_weight_rule_known_answers():
    y = torch.zeros(100)
    y[:4] = 1                       # neg/pos = 24 -> sqrt = 4.899
    x = torch.zeros(100)            # softplus(0) = ln 2 for every element
    want = (96 * 0.6931471805599453 + 4 * (24 ** 0.5) * 0.6931471805599453) / 100
    assert abs(O.grounding_loss(x, y).item() - want) < 1e-6
    y[:4] = 0
    y[0] = 1                        # neg/pos = 99 -> capped at 5
    want = (99 + 5) * 0.6931471805599453 / 100
    assert abs(O.grounding_loss(x, y).item() - want) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n,p,scale", CASES)
def test_gpu_loss_matches_oracle(n, p, scale):
    from videoitg_amd import ops
    x, y = _data(n, p, scale)
    loss, pw = ops.bce_loss(x.cuda(), y.cuda())
    want = O.grounding_loss(x, y).item()
    # fp32, different summation order: relative 1e-5
    assert abs(loss.item() - want) <= 1e-5 * max(1.0, abs(want))
    pos = y.sum().item()
    assert abs(pw.item() - min(5.0, ((n - pos) / max(1.0, pos)) ** 0.5)) < 1e-5


@pytest.mark.gpu
def test_model_forward_with_grounding_labels():
    """model(..., grounding_labels=[...]) -> .loss scalar, .logits flat fp32 (grounding_qwen2.py:160,175-181)"""
    from videoitg_amd.model.builder import load_pretrained_model
    from videoitg_amd import synth
    _, model, _, _ = load_pretrained_model("synthetic:tiny:0", None, "videoitg-tiny", device_map="cuda:0")
    model = model.half().eval()
    cfg = model.config
    T = [6, 4]
    images = [synth.synth_pixel_values(t, cfg.vision.image_size, seed=50 + i).cuda() for i, t in enumerate(T)]
    ids = torch.tensor([[-200, 5, 6, 7], [-200, 9, 10, 11]])
    mask = torch.ones_like(ids, dtype=torch.bool)
    labels = [torch.tensor([1, 0, 0, 1, 0, 0.0]), torch.tensor([0, 0, 1, 0.0])]
    plain = model(input_ids=ids, attention_mask=mask, images=images)
    out = model(input_ids=ids, attention_mask=mask, images=images, grounding_labels=labels)
    flat = torch.cat([l.float().view(-1) for l in plain.logits]).cpu()
    assert out.logits.shape == (10,) and out.logits.dtype == torch.float32
    assert torch.equal(out.logits.cpu(), flat)
    want = O.grounding_loss(flat, torch.cat(labels)).item()
    assert abs(out.loss.item() - want) <= 1e-5 * max(1.0, abs(want))
