"""Host-side helpers of the frame-selector path, same names / argument meaning / error behaviour as
/root/reference/eagle/mm_utils.py (rows A1, A2, A4 of SURVEY.md §8(a)) plus the lmms-eval sampler
(row A1', /root/reference/lmms_eval/models/videoitg.py:82-93).

Pure Python integer logic: no device code here.  Video decoding needs ``decord`` / ``av``, neither of
which exists in this image; the readers import them lazily and raise ImportError like the reference
would at import time.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from .constants import IMAGE_TOKEN_INDEX


def get_frame_indices(total_frames: int, original_fps: float, target_fps: float, num_frm: int) -> List[int]:
    """Row A1 (eagle/mm_utils.py:33-41; identical copy infer.py:34-42).

    Stride = max(1, round(fps/target)) with Python's half-to-even ``round``; if the strided list has at
    least ``num_frm`` entries, pick entry ``round((i+1)*len/num_frm - 1)`` for i in [0, num_frm)."""
    step = max(1, round(original_fps / target_fps))
    strided = range(0, total_frames, step)
    n = len(strided)
    if n < num_frm:
        return list(strided)
    scale = 1.0 * n / num_frm
    return [strided[round((i + 1) * scale - 1)] for i in range(num_frm)]


def get_seq_frames(total_frames: int, original_fps: float, target_fps: float, num_frm: int,
                   multiple: int = 1) -> List[int]:
    """Row A1' (lmms_eval/models/videoitg.py:82-93): no ``max(1, .)`` on the stride (a stride of 0 raises
    ValueError from ``range`` exactly as in the reference), truncating pick ``int(i*len/num_frm)``, short
    lists padded with index 0 up to a multiple of ``multiple``."""
    step = round(original_fps / target_fps)
    strided = list(range(0, total_frames, step))
    if len(strided) < num_frm:
        while len(strided) % multiple != 0:
            strided.append(0)
        return strided
    scale = len(strided) / num_frm
    return [strided[int(i * scale)] for i in range(num_frm)]


def tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX,
                          return_tensors: Optional[str] = None):
    """Row A4 (eagle/mm_utils.py:279-298): tokenise the text between ``<image>`` markers and join the pieces
    with ``image_token_index``.  If the first piece starts with the tokenizer's BOS id, BOS is kept once and
    stripped from every later piece."""
    pieces = [tokenizer(chunk).input_ids for chunk in prompt.split("<image>")]
    has_bos = bool(pieces) and len(pieces[0]) > 0 and pieces[0][0] == tokenizer.bos_token_id
    skip = 1 if has_bos else 0
    ids: List[int] = [pieces[0][0]] if has_bos else []
    for n, piece in enumerate(pieces):
        if n > 0:
            ids.append(image_token_index)
        ids.extend(piece[skip:])
    if return_tensors is not None:
        if return_tensors == "pt":
            return torch.tensor(ids, dtype=torch.long)
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return ids


def get_model_name_from_path(model_path: str) -> str:
    """eagle/mm_utils.py:301-307."""
    parts = model_path.strip("/").split("/")
    if parts[-1].startswith("checkpoint-"):
        return parts[-2] + "_" + parts[-1]
    return parts[-1]


def pad_sequence(tokenizer, input_ids: Sequence[torch.Tensor], batch_first: bool, padding_value: int) -> torch.Tensor:
    """Row A5 (infer.py:26-32; lmms_eval/models/videoitg.py:349-355): left- or right-pad per
    ``tokenizer.padding_side``."""
    left = getattr(tokenizer, "padding_side", "right") == "left"
    seqs = [torch.flip(x, [0]) for x in input_ids] if left else list(input_ids)
    out = torch.nn.utils.rnn.pad_sequence(seqs, batch_first=batch_first, padding_value=padding_value)
    return torch.flip(out, [1]) if left else out


def open_video_reader(video_path: str, num_threads: int = 4):
    """decord.VideoReader(path, ctx=cpu(0)) as the reference opens it (infer.py:45; lmms_eval/models/videoitg.py:98) - or, when decord is
    not installed AND the file is in a container the built-in reader parses (videoitg_amd.video_io: AVI, YUV4MPEG2, animated images, ISO base
    media / QuickTime) with frames that need no codec library, that reader with the same call surface.  Anything else without decord raises
    the ImportError the reference raises."""
    try:
        from decord import VideoReader, cpu
    except ImportError:
        from . import video_io
        # .mp4 / .mov / .m4v: the built-in reader parses the container; a track of individually coded frames (Motion-JPEG, PNG, uncompressed) is
        # decoded, a track that needs a codec raises video_io.CodecNeeded - an ImportError like the one above, naming the codec
        if not str(video_path).lower().endswith(video_io.SUFFIXES + video_io.BMFF_SUFFIXES):
            raise
        return video_io.VideoReader(video_path, ctx=None, num_threads=num_threads)
    return VideoReader(video_path, ctx=cpu(0), num_threads=num_threads)


def read_video_decord(video_path: str, num_frm: int = 16, target_fps: float = 2):
    """Row A2 (infer.py:44-51): uint8 [T,H,W,3] frames, their indices, duration in seconds."""
    vr = open_video_reader(video_path, num_threads=4)
    total = len(vr)
    fps = vr.get_avg_fps()
    idx = get_frame_indices(total, fps, target_fps, num_frm)
    frames = vr.get_batch(idx)
    vr.seek(0)
    return frames.asnumpy(), idx, round(total / fps)


def read_video_pyav(video_path: str, num_frm: int = 16, target_fps: float = 2, threads: int = 4) -> np.ndarray:
    """eagle/mm_utils.py:43-69 (PyAV fallback): decode, then sample with get_frame_indices."""
    import av  # not in this image

    container = av.open(video_path)
    stream = container.streams.video[0]
    stream.thread_type = "AUTO"
    stream.codec_context.thread_count = threads
    fps = stream.average_rate
    total = stream.frames
    frames = None
    if "webm" not in video_path and "mkv" not in video_path:
        try:
            idx = get_frame_indices(total, fps, target_fps, num_frm)
            wanted, last = set(idx), idx[-1]
            frames = []
            for i, fr in enumerate(container.decode(video=0)):
                if i > last:
                    break
                if i in wanted:
                    frames.append(fr)
        except Exception:
            frames = None
    if frames is None:
        every = [fr for packet in container.demux(video=0) for fr in packet.decode()]
        idx = get_frame_indices(len(every), fps, target_fps, num_frm)
        frames = [every[i] for i in idx]
    return np.stack([f.to_ndarray(format="rgb24") for f in frames])
