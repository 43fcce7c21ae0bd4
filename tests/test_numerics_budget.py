"""CPU: the rounding model of the product (oracle/quant_model.py) degenerates to the oracle without rounding, and the
product's rounding sites cost no more error than the reference's own all-fp16 arithmetic."""
import json
import os

import numpy as np
import torch

from oracle import quant_model as QM
from oracle import videoitg_oracle as O
from videoitg_amd import synth
from videoitg_amd.config import ITGConfig


def _case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"{name}.npz"))
    cfg = ITGConfig.from_dict(json.loads(str(z["cfg"])))
    T = z["T"].tolist()
    images = [synth.synth_pixel_values(t, cfg.vision.image_size, seed=1234 + i) for i, t in enumerate(T)]
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    return z, cfg, sd, vsd, torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]), images


def test_no_rounding_equals_oracle(golden_dir):
    z, cfg, sd, vsd, ids, mask, images = _case(golden_dir, "batch2_pad")
    a = QM.forward(cfg, sd, vsd, ids, mask, images, dtype=None)
    for b, lg in enumerate(a):
        assert (lg - torch.from_numpy(z[f"logits{b}"])).abs().max() < 2e-5


def test_product_rounding_not_worse_than_reference_fp16(golden_dir):
    z, cfg, sd, vsd, ids, mask, images = _case(golden_dir, "pool_post")
    ref = O.forward(cfg, sd, vsd, ids, mask, images)[0]
    ours = QM.forward(cfg, sd, vsd, ids, mask, images, dtype=torch.float16, mode="product")[0]
    theirs = QM.forward(cfg, sd, vsd, ids, mask, images, dtype=torch.float16, mode="reference16")[0]
    e_ours, e_theirs = (ours - ref).abs().max().item(), (theirs - ref).abs().max().item()
    assert e_ours < 1e-2 and e_ours <= 1.25 * e_theirs, (e_ours, e_theirs)
