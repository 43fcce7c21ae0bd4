#!/usr/bin/env python3
"""Top-K frame selection entry point — twin of /root/reference/infer.py on the MI355X path.

    python infer.py [--model nvidia/VideoITG-8B | <checkpoint dir> | synthetic:tiny|full] [--video PATH | synthetic]
                    [--prompt TEXT] [--topk 32] [--dtype fp16|bf16]

Same flow as the reference (infer.py:53-79): decode up to 512 frames at 2 fps -> SigLIP preprocessing -> "<image>" +
prompt + "\\n" -> model(input_ids, attention_mask, images=[video]) -> sigmoid -> sort descending -> first K original
frame indices, ascending.  `--video synthetic` replaces the decoder (decord/PyAV are not in this image) with seeded
random frames so the whole entry point can run offline.
"""
import argparse

import numpy as np
import torch

from videoitg_amd import ops
from videoitg_amd.constants import DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX
from videoitg_amd.mm_utils import (get_frame_indices, get_model_name_from_path, pad_sequence, read_video_decord,
                                   tokenizer_image_token)
from videoitg_amd.model.builder import load_pretrained_model

DEVICE = torch.device("cuda:0")


def read_video(path, num_frm=512, target_fps=2):
    if path != "synthetic":
        return read_video_decord(path, num_frm=num_frm, target_fps=target_fps)
    fps = 29.97
    total = int(num_frm * fps / target_fps * 1.17)              # a clip a bit longer than num_frm samples at target_fps (512 -> ~10 min)
    idx = get_frame_indices(total, fps, target_fps, num_frm)
    rng = np.random.default_rng(1234)
    frames = rng.integers(0, 256, (len(idx), 120, 160, 3), dtype=np.uint8)
    return frames, idx, round(total / fps)


def topk_selection(model, tokenizer, image_processor, prompt, video_path, num_topk, dtype, num_frm=512):
    video, frame_idx, _ = read_video(video_path, num_frm=num_frm, target_fps=2)                   # infer.py:54
    video_tensor = image_processor.preprocess(video, return_tensors="pt")["pixel_values"].to(dtype).to(DEVICE)
    prompt = DEFAULT_IMAGE_TOKEN + prompt + "\n"                                                   # infer.py:60
    input_ids = tokenizer_image_token(prompt, tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt")
    pad_id = tokenizer.pad_token_id if tokenizer.pad_token_id is not None else tokenizer.eos_token_id
    input_ids = pad_sequence(tokenizer, [input_ids], batch_first=True, padding_value=pad_id).to(DEVICE)
    with torch.inference_mode():
        response = model(input_ids, attention_mask=input_ids.ne(pad_id), images=[video_tensor])
        scores, order = ops.rank(response.logits[0].view(-1).float())     # sigmoid + stable descending sort on the GPU
    selected = sorted(frame_idx[i] for i in order[:num_topk].tolist())                            # infer.py:77-78
    return selected, scores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="nvidia/VideoITG-8B")
    ap.add_argument("--video", default="./assets/imax.mp4")
    ap.add_argument("--prompt", default="Which IMAX movie isn't in the video? A. The Hunger Games: B. Catching Fire; C. The Dark "
                                        "Knight; D. Oppenheimer; E. Dune \n Please respond with only the letter of the correct answer.")
    ap.add_argument("--topk", type=int, default=32)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--num-frames", type=int, default=0, help="frames to sample (reference: 512; default 32 for synthetic:tiny, whose "
                                                              "72-token visual budget pools 512 frames down to nothing)")
    a = ap.parse_args()
    num_frm = a.num_frames or (32 if a.model.startswith("synthetic:tiny") or a.model == "synthetic" else 512)
    dtype = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    tok, model, proc, _ = load_pretrained_model(a.model, None, get_model_name_from_path(a.model), device_map="cuda:0", torch_dtype=dtype)
    model.eval().to(DEVICE)
    selected, _ = topk_selection(model, tok, proc, a.prompt, a.video, min(a.topk, num_frm), dtype, num_frm)
    print(selected)


if __name__ == "__main__":
    main()
