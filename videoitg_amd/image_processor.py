"""SigLIP image preprocessing twin (row A3 of SURVEY.md §8a).

The reference obtains ``image_processor`` from the tower ([3P] transformers ``SiglipImageProcessor``,
/root/reference/eagle/model/multimodal_encoder/clip_encoder.py:115) and calls
``image_processor.preprocess(frames, return_tensors="pt")["pixel_values"]`` (/root/reference/infer.py:55,
lmms_eval/models/videoitg.py:286).  Published algorithm (preprocessor_config.json of
google/siglip-so400m-patch14-384): PIL BICUBIC resize to (S,S) on uint8, rescale 1/255, normalise with
mean = std = 0.5, HWC -> CHW.  CPU work, outside the timed region.
"""
from __future__ import annotations

from typing import Sequence, Union

import numpy as np
import torch
from PIL import Image


class BatchFeature(dict):
    """dict with attribute access, like transformers.BatchFeature"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class SiglipImageProcessor:
    def __init__(self, size=384, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5), rescale_factor=1 / 255,
                 resample=Image.BICUBIC):
        if isinstance(size, dict):
            size = size["height"]
        self.size = {"height": int(size), "width": int(size)}
        self.image_mean = list(image_mean)
        self.image_std = list(image_std)
        self.rescale_factor = rescale_factor
        self.resample = resample
        self.crop_size = dict(self.size)

    def _one(self, img) -> np.ndarray:
        if isinstance(img, torch.Tensor):
            img = img.cpu().numpy()
        if isinstance(img, np.ndarray):
            if img.dtype != np.uint8:
                img = np.clip(np.rint(img * 255.0 if img.max() <= 1.0 else img), 0, 255).astype(np.uint8)
            if img.ndim == 3 and img.shape[0] in (1, 3) and img.shape[-1] not in (1, 3):
                img = np.transpose(img, (1, 2, 0))
            img = Image.fromarray(img)
        img = img.convert("RGB")
        s = self.size["height"]
        if img.size != (s, s):
            img = img.resize((s, s), resample=self.resample)
        # transformers image_transforms.rescale: float64 product, THEN the cast to float32 (a float32 product differs in the last
        # bit for some byte values); normalize: (x - mean) / std in float32
        a = (np.asarray(img).astype(np.float64) * self.rescale_factor).astype(np.float32)
        a = (a - np.asarray(self.image_mean, np.float32)) / np.asarray(self.image_std, np.float32)
        return np.ascontiguousarray(a.transpose(2, 0, 1))

    def preprocess(self, images: Union[np.ndarray, Sequence, Image.Image], return_tensors=None, **_):
        # GPU front-end: decoded frames already in HBM as uint8 [T,H,W,3] are resized / normalised by the HIP
        # kernel (bit-identical to the host path below) and stay on the device
        if isinstance(images, torch.Tensor) and images.is_cuda and images.dtype == torch.uint8 and images.dim() == 4:
            if return_tensors != "pt":
                raise ValueError("device frames are returned as torch tensors: pass return_tensors='pt'")
            if (self.image_mean, self.image_std) != ([0.5, 0.5, 0.5], [0.5, 0.5, 0.5]) or self.rescale_factor != 1 / 255 \
                    or self.resample != Image.BICUBIC:
                raise ValueError("the device path implements the SigLIP preprocessor_config (bicubic, 1/255, mean = std = 0.5)")
            from .torch_ops import K as ops
            return BatchFeature(pixel_values=ops.preprocess_u8(images, self.size["height"], torch.float32))
        if isinstance(images, Image.Image):
            images = [images]
        elif isinstance(images, (np.ndarray, torch.Tensor)) and images.ndim == 3:
            images = [images]
        out = np.stack([self._one(im) for im in images])
        if return_tensors == "pt":
            return BatchFeature(pixel_values=torch.from_numpy(out))
        if return_tensors in (None, "np"):
            return BatchFeature(pixel_values=out if return_tensors == "np" else list(out))
        raise ValueError(f"Unsupported tensor type: {return_tensors}")

    __call__ = preprocess
