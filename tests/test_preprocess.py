"""Frame preprocessing: oracle and host twin against vectors made with the real
SiglipImageProcessor + Pillow (tests/golden/make_preprocess_golden.py); GPU kernel against the oracle, bit-exact."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as PO
from videoitg_amd.synth import synth_video_frames

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL = ["down", "up", "same", "odd"]
HD = ["hd720", "hd1080"]


def _frames(spec):
    seed, T, H, W, S = [int(v) for v in spec]
    return synth_video_frames(seed, T, H, W), S


@pytest.mark.parametrize("name", SMALL)
def test_oracle_matches_real_processor_small(name):
    z = np.load(os.path.join(GOLD, "preprocess_small.npz"))
    fr, S = _frames(z[f"{name}_spec"])
    assert np.array_equal(PO.pil_bicubic_resize(fr[0], S, S), z[f"{name}_u8"])            # integer resize: bit-exact
    got = PO.preprocess(fr, S)
    assert got.dtype == np.float32 and np.array_equal(got, z[f"{name}_out"])               # same float32 bits


@pytest.mark.parametrize("name", HD)
def test_oracle_matches_real_processor_hd(name):
    z = np.load(os.path.join(GOLD, "preprocess_hd.npz"))
    fr, S = _frames(z[f"{name}_spec"])
    u8 = PO.pil_bicubic_resize(fr[0], S, S)
    assert zlib.crc32(np.ascontiguousarray(u8).tobytes()) == int(z[f"{name}_crc"][0])
    out = PO.preprocess(fr, S)
    assert np.array_equal(out.reshape(-1)[z[f"{name}_idx"]], z[f"{name}_val"])
    assert out.astype(np.float64).sum() == float(z[f"{name}_sum"][0])


@pytest.mark.parametrize("name", SMALL)
def test_host_twin_matches_real_processor(name):
    from videoitg_amd.image_processor import SiglipImageProcessor
    z = np.load(os.path.join(GOLD, "preprocess_small.npz"))
    fr, S = _frames(z[f"{name}_spec"])
    got = SiglipImageProcessor(size=S).preprocess(fr, return_tensors="pt")["pixel_values"]
    assert got.dtype == torch.float32 and np.array_equal(got.numpy(), z[f"{name}_out"])


def test_resample_coeffs_properties():
    """fixed-point rows sum to 2^22 within the per-tap rounding; windows stay inside the image; identity when sizes match"""
    for n_in, n_out in [(1280, 384), (40, 56), (56, 56), (33, 42), (1920, 448)]:
        b, k = PO.resample_coeffs(n_in, n_out)
        assert (b[:, 0] >= 0).all() and (b[:, 0] + b[:, 1] <= n_in).all()
        assert np.abs(k.sum(1) - (1 << 22)).max() <= k.shape[1]
    img = np.random.default_rng(0).integers(0, 256, (9, 7, 3), dtype=np.uint8)
    assert np.array_equal(PO.pil_bicubic_resize(img, 9, 7), img)


def test_normalise_lut_is_the_transformers_arithmetic():
    lut = PO.normalise_lut()
    assert lut[0] == -1.0 and lut[255] == 1.0 and lut.dtype == np.float32
    u = np.arange(256)
    want = ((u.astype(np.float64) * (1 / 255)).astype(np.float32) - np.float32(0.5)) / np.float32(0.5)
    assert np.array_equal(lut, want)


# ---- GPU --------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", SMALL + HD)
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_gpu_preprocess_bit_exact(name, out_dtype):
    from videoitg_amd import ops
    z = np.load(os.path.join(GOLD, "preprocess_small.npz" if name in SMALL else "preprocess_hd.npz"))
    fr, S = _frames(z[f"{name}_spec"])
    want = torch.from_numpy(PO.preprocess(fr, S)).to(out_dtype)          # .half() of the fp32 result = what infer.py:55 feeds the model
    got = ops.preprocess_u8(torch.from_numpy(fr).cuda(), S, out_dtype)
    assert got.shape == (fr.shape[0], 3, S, S) and got.dtype == out_dtype
    assert torch.equal(got.cpu(), want)


@pytest.mark.gpu
def test_gpu_preprocess_strided_batch_and_errors():
    from videoitg_amd import ops
    from videoitg_amd._lib import VitgError
    fr = synth_video_frames(5, 6, 48, 64)
    big = torch.from_numpy(fr).cuda()
    got = ops.preprocess_u8(big[1:5], 56, torch.float32)                # a view into a larger clip
    assert torch.equal(got.cpu(), torch.from_numpy(PO.preprocess(fr[1:5], 56)))
    with pytest.raises((VitgError, ValueError, TypeError)):
        ops.preprocess_u8(big.float(), 56, torch.float32)               # not uint8
    assert ops.preprocess_u8(big[:0], 56, torch.float32).shape == (0, 3, 56, 56)


@pytest.mark.gpu
def test_gpu_image_processor_front_end_matches_host_twin():
    from videoitg_amd.image_processor import SiglipImageProcessor
    fr = synth_video_frames(9, 4, 120, 160)
    host = SiglipImageProcessor(size=56).preprocess(fr, return_tensors="pt")["pixel_values"]
    dev = SiglipImageProcessor(size=56).preprocess(torch.from_numpy(fr).cuda(), return_tensors="pt")["pixel_values"]
    assert dev.is_cuda and torch.equal(dev.cpu(), host)


# ---- property test: the restatement against Pillow itself on arbitrary sizes (Pillow is the third-party library the reference's
# processor calls; it ships in this image, so beyond the committed vectors the oracle can be pinned on the fly) ----------------
try:
    from hypothesis import given, settings, strategies as st
    from PIL import Image as _PILImage
    _HAVE = True
except Exception:                                       # pragma: no cover
    _HAVE = False


@pytest.mark.skipif(not _HAVE, reason="hypothesis / Pillow not importable")
def test_oracle_resize_equals_pillow_on_random_sizes():
    @settings(max_examples=40, deadline=None)
    @given(h=st.integers(1, 97), w=st.integers(1, 97), oh=st.integers(1, 80), ow=st.integers(1, 80), seed=st.integers(0, 2 ** 16))
    def prop(h, w, oh, ow, seed):
        img = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(_PILImage.fromarray(img).resize((ow, oh), resample=_PILImage.BICUBIC))
        assert np.array_equal(PO.pil_bicubic_resize(img, oh, ow), ref)
    prop()


@pytest.mark.gpu
def test_gpu_preprocess_equals_oracle_on_odd_sizes():
    """sizes the fixtures do not hold: 1-pixel inputs, extreme down-scaling (long filter windows), widths whose rows are not 4-byte
    aligned, output sides that are not multiples of 4 (byte-wise vertical pass)"""
    from videoitg_amd import ops
    for (T, H, W, S) in [(2, 1, 1, 8), (1, 5, 301, 9), (3, 211, 7, 33), (2, 64, 64, 30), (1, 17, 1000, 28), (2, 33, 35, 34)]:
        fr = synth_video_frames(100 + H + W, T, H, W)
        want = torch.from_numpy(PO.preprocess(fr, S))
        got = ops.preprocess_u8(torch.from_numpy(fr).cuda(), S, torch.float32)
        assert torch.equal(got.cpu(), want), (T, H, W, S)


TALL = [(201, 2, 21), (200, 2, 21), (301, 3, 8), (300, 3, 8), (349, 3, 21), (308, 3, 224), (509, 5, 56), (1101, 11, 64), (1100, 11, 64), (1001, 10, 1000),
        (3, 700, 16), (101, 1, 8), (296, 2, 448), (435, 4, 471), (400, 2, 401), (400, 2, 399), (435, 4, 434), (283, 1, 371)]


@pytest.mark.skipif(not _HAVE, reason="hypothesis / Pillow not importable")
@pytest.mark.parametrize("H,W,S", TALL)
def test_pillow_resamples_very_tall_images_vertical_pass_first(H, W, S):
    """Found by tools/fuzz_kernels.py (round 6): Pillow (12.2 here) runs the VERTICAL pass first when an image more than 100 times taller than wide is
    reduced in height - exactly from H = 100 W + 1, only for S < H, never for wide images - and each pass rounds to 8 bits, so the order shows (up to 20 grey levels at
    308 x 3 -> 224).  The oracle restates the rule; here it is pinned against the library on both sides of the switch, and against the real
    SiglipImageProcessor path through the host twin."""
    from videoitg_amd.image_processor import SiglipImageProcessor
    img = np.random.default_rng(H * 7 + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    ref = np.asarray(_PILImage.fromarray(img).resize((S, S), resample=_PILImage.BICUBIC))
    assert np.array_equal(PO.pil_bicubic_resize(img, S, S), ref)
    twin = SiglipImageProcessor(size=S).preprocess(img[None], return_tensors="np")["pixel_values"]
    assert np.array_equal(np.stack(twin), PO.preprocess(img[None], S))
    if H > 100 * W and S < H and W > 1 and W != S:     # the order matters on these draws (a 1-pixel-wide image is constant along x: either order): the test would notice a regression to horizontal-first
        hfirst = PO._pass(PO._pass(img, *PO.resample_coeffs(W, S), axis=1), *PO.resample_coeffs(H, S), axis=0)
        assert not np.array_equal(hfirst, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,S", TALL)
def test_gpu_preprocess_on_very_tall_images(H, W, S):
    """the product path on the geometry where Pillow swaps its passes: `ops.preprocess_u8` transposes around the horizontal-first kernel; the C entry
    itself refuses the geometry (include/videoitg_hip.h) instead of returning the other order's pixels"""
    from videoitg_amd import _lib, ops
    fr = np.random.default_rng(H * 7 + W).integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    want = torch.from_numpy(PO.preprocess(fr, S))
    dev = torch.from_numpy(fr).cuda()
    for dt in (torch.float32, torch.float16):
        got = ops.preprocess_u8(dev, S, dt)
        assert got.is_contiguous() and torch.equal(got.cpu(), want.to(dt)), (H, W, S, dt)
    if H > 100 * W and S < H and 1 < W != S:
        import ctypes as C
        need = C.c_long(0)
        _lib.call("vitg_preprocess_tmp_bytes", 2, H, W, S, C.byref(need))
        tmp = torch.empty(max(need.value, 1), dtype=torch.uint8, device="cuda")
        out = torch.empty((2, 3, S, S), device="cuda")
        with pytest.raises(_lib.VitgError):
            _lib.call("vitg_preprocess_u8", dev.data_ptr(), 2, H, W, dev.stride(0), dev.stride(1), out.data_ptr(), S, ops.dt_code(torch.float32),
                      tmp.data_ptr(), need.value, None)
