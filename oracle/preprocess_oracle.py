"""CPU restatement of the frame preprocessing in front of the selector.  TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.

Reference call sites: /root/reference/infer.py:55 and /root/reference/lmms_eval/models/videoitg.py:282,286 call
``image_processor.preprocess(frames, return_tensors="pt")["pixel_values"]``; the object is the third-party
``SiglipImageProcessor`` of transformers (pinned 4.47.1 in the reference's requirements.txt:18; 5.15.0 is installed here,
same arithmetic) created at /root/reference/eagle/model/multimodal_encoder/clip_encoder.py:115, which resizes with Pillow
(``Image.resize((S, S), resample=BICUBIC)``).  Neither library is vendored in the reference, so their published algorithms
are restated here and PINNED against the libraries themselves: tests/golden/make_preprocess_golden.py runs the real
SiglipImageProcessor / Pillow in this container and commits inputs-by-seed + outputs (tests/golden/preprocess_*.npz).

Pillow's resampling (src/libImaging/Resample.c), 8-bit path, restated:
  * per output coordinate xx: center = (xx + 0.5) * scale, support = 2.0 * max(scale, 1)  (bicubic, a = -0.5),
    window [xmin, xmax) = [int(center - support + 0.5), int(center + support + 0.5)) clipped to the image,
    weights w(x) = bicubic((x + xmin - center + 0.5) / max(scale, 1)), normalised to sum 1 in double precision;
  * weights are converted to fixed point with PRECISION_BITS = 22: k = int(w * 2^22 + (w < 0 ? -0.5 : 0.5)) (truncation
    toward zero of the offset value);
  * each pass accumulates  ss = 2^21 + sum(pixel * k)  in int32 and stores clip8(ss >> 22); the horizontal pass runs first
    and its uint8 result is the input of the vertical pass (images more than 100 times taller than wide whose height is reduced: the vertical pass first).
Then (transformers image_transforms.rescale / normalize): x = float32(float64(u8) * (1/255)); y = (x - 0.5f) / 0.5f in
float32; HWC -> CHW.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int):
    """-> (bounds int32 [out, 2] = (xmin, count), kk int32 [out, ksize]) exactly as Pillow's precompute_coeffs +
    normalize_coeffs_8bpc."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        n = xmax - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        bounds[xx] = (xmin, n)
        for x, v in enumerate(w):
            kk[xx, x] = int(v * (1 << PRECISION_BITS) + (-0.5 if v < 0 else 0.5))       # C cast: truncation toward zero
    return bounds, kk


def _pass(img: np.ndarray, bounds, kk, axis: int) -> np.ndarray:
    """one resampling pass over `axis` of a uint8 [H, W, C] image"""
    out_size = bounds.shape[0]
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.tensordot(kk[xx, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_bicubic_resize(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [H, W, 3] -> uint8 [out_h, out_w, 3], bit-exact Pillow ``resize((out_w, out_h), BICUBIC)``"""
    h, w, _ = img.shape
    if h > 100 * w and out_h < h:
        # Pillow 12.2 (the library this oracle is pinned to) runs the VERTICAL pass first when an image more than 100 times taller than wide is reduced in
        # height - found by tools/fuzz_kernels.py in round 6 (349 x 3 -> 21: the two orders differ by up to 20 grey levels, each pass rounds to 8 bits) and
        # pinned against the library in tests/test_preprocess.py: exactly from h = 100 w + 1, only when out_h < h, never for wide images (rule fitted on
        # 2 000 random geometries on which the two orders differ, no violation)
        img = _pass(img, *resample_coeffs(h, out_h), axis=0)
        if w != out_w:
            img = _pass(img, *resample_coeffs(w, out_w), axis=1)
        return img
    if w != out_w:
        img = _pass(img, *resample_coeffs(w, out_w), axis=1)         # horizontal first
    if h != out_h:
        img = _pass(img, *resample_coeffs(h, out_h), axis=0)
    return img


def normalise_lut(rescale_factor: float = 1 / 255, mean: float = 0.5, std: float = 0.5) -> np.ndarray:
    """the 256 possible output values: float32(float64(u) * factor), then (x - mean) / std in float32"""
    x = (np.arange(256, dtype=np.float64) * rescale_factor).astype(np.float32)
    return ((x - np.float32(mean)) / np.float32(std)).astype(np.float32)


def preprocess(frames: np.ndarray, size: int) -> np.ndarray:
    """uint8 [T, H, W, 3] -> float32 [T, 3, size, size]"""
    lut = normalise_lut()
    out = np.empty((frames.shape[0], 3, size, size), np.float32)
    for t in range(frames.shape[0]):
        r = pil_bicubic_resize(frames[t], size, size)
        out[t] = lut[r].transpose(2, 0, 1)
    return out
