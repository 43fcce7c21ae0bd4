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
import os

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


def frame_at(video_path, frame_idx):
    """one decoded frame (uint8 HWC) by its ORIGINAL frame index - what infer.py:88-89 does with decord's vr[frame_idx]"""
    if video_path == "synthetic":
        return np.random.default_rng(1234 + int(frame_idx)).integers(0, 256, (120, 160, 3), dtype=np.uint8)
    from videoitg_amd.mm_utils import open_video_reader   # decord, or the built-in reader for .avi (MJPEG) / .y4m files
    return open_video_reader(video_path, num_threads=4)[frame_idx].asnumpy()


def save_selected_frames(video_path, selected_indices, save_dir="./vis"):
    """infer.py:81-94: the selected frames as JPEGs named frame_<rank>_idx<original index>.jpg"""
    from PIL import Image
    os.makedirs(save_dir, exist_ok=True)
    for i, frame_idx in enumerate(selected_indices):
        Image.fromarray(frame_at(video_path, frame_idx)).save(os.path.join(save_dir, f"frame_{i:03d}_idx{frame_idx}.jpg"), "JPEG")
    print(f"saved {len(selected_indices)} frames to {save_dir}")


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
    ap.add_argument("--operand-split", action="store_true",
                    help="this build's precision tier: every MFMA operand as an fp16 (hi, lo) pair - scores within ~1e-5 of the reference's fp32 CPU path on any "
                         "16-bit checkpoint, 2.4x the time (fp16 only; profiles/r06_operand_split.md)")
    ap.add_argument("--save-dir", default="./vis", help="where the selected frames are written as JPEGs (infer.py:81-94); '' = do not save")
    ap.add_argument("--num-frames", type=int, default=0, help="frames to sample (reference: 512; default 32 for synthetic:tiny, whose "
                                                              "72-token visual budget pools 512 frames down to nothing)")
    a = ap.parse_args()
    num_frm = a.num_frames or (32 if a.model.startswith("synthetic:tiny") or a.model == "synthetic" else 512)
    dtype = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    tok, model, proc, _ = load_pretrained_model(a.model, None, get_model_name_from_path(a.model), device_map="cuda:0", torch_dtype=dtype,
                                                operand_split=True if a.operand_split else None)
    model.eval().to(DEVICE)
    selected, _ = topk_selection(model, tok, proc, a.prompt, a.video, min(a.topk, num_frm), dtype, num_frm)
    print(selected)
    if a.save_dir:
        save_selected_frames(a.video, selected, save_dir=a.save_dir)


if __name__ == "__main__":
    main()
