#!/usr/bin/env python3
"""Golden vectors for the frame preprocessing, generated with the REAL third-party
libraries the reference calls: transformers' SiglipImageProcessor (/root/reference/eagle/model/multimodal_encoder/
clip_encoder.py:115, used at /root/reference/infer.py:55) and Pillow's BICUBIC resize underneath it.

Inputs are regenerated from seeds by the tests (numpy default_rng is a stable stream); outputs are committed:
  preprocess_small.npz : full float32 outputs of three small cases + the uint8 resize of each first frame
  preprocess_hd.npz    : 720x1280 -> 384 and 1080x1920 -> 448: CRC32 of the uint8 resize, sum and 64 sampled float outputs

    python tests/golden/make_preprocess_golden.py        (this container only; needs transformers + Pillow)
"""
import os
import zlib
import warnings

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from videoitg_amd.synth import synth_video_frames as synth_frames  # noqa: E402  (inputs are regenerated from seeds by the tests)


SMALL = [("down", 11, 3, 90, 120, 56), ("up", 12, 2, 30, 40, 56), ("same", 13, 2, 56, 56, 56), ("odd", 14, 2, 75, 33, 42)]
HD = [("hd720", 21, 1, 720, 1280, 384), ("hd1080", 22, 1, 1080, 1920, 448)]


def main():
    warnings.filterwarnings("ignore")
    from transformers import SiglipImageProcessor
    small = {}
    for name, seed, T, H, W, S in SMALL:
        fr = synth_frames(seed, T, H, W)
        proc = SiglipImageProcessor(size={"height": S, "width": S})
        out = proc.preprocess(fr, return_tensors="pt")["pixel_values"].numpy()
        assert out.dtype == np.float32 and out.shape == (T, 3, S, S)
        small[f"{name}_spec"] = np.array([seed, T, H, W, S], np.int64)
        small[f"{name}_out"] = out
        small[f"{name}_u8"] = np.asarray(Image.fromarray(fr[0]).resize((S, S), resample=Image.BICUBIC))
    np.savez_compressed(os.path.join(HERE, "preprocess_small.npz"), **small)
    hd = {}
    for name, seed, T, H, W, S in HD:
        fr = synth_frames(seed, T, H, W)
        proc = SiglipImageProcessor(size={"height": S, "width": S})
        out = proc.preprocess(fr, return_tensors="pt")["pixel_values"].numpy()
        u8 = np.asarray(Image.fromarray(fr[0]).resize((S, S), resample=Image.BICUBIC))
        idx = np.random.default_rng(seed + 100).integers(0, out.size, 64)
        hd[f"{name}_spec"] = np.array([seed, T, H, W, S], np.int64)
        hd[f"{name}_crc"] = np.array([zlib.crc32(np.ascontiguousarray(u8).tobytes())], np.uint32)
        hd[f"{name}_sum"] = np.array([out.astype(np.float64).sum()], np.float64)
        hd[f"{name}_idx"] = idx
        hd[f"{name}_val"] = out.reshape(-1)[idx]
    np.savez_compressed(os.path.join(HERE, "preprocess_hd.npz"), **hd)
    print("wrote preprocess_small.npz, preprocess_hd.npz")


if __name__ == "__main__":
    main()
