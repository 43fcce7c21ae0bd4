"""``videoitg`` lmms-eval model twin (/root/reference/lmms_eval/models/videoitg.py:132-357).

Second caller of the selector path: per document it samples frames (``get_seq_frames``, row A1'), preprocesses them,
runs the same forward as ``infer.py`` and appends one JSON object per line to ``<output_dir>/results.jsonl``

    {"index": [original frame ids, descending score], "logits": [scores rounded to 2 dp], "num_frames": <#visuals>,
     "contexts": ..., "video_path": ..., "doc_id": ...}

which downstream answerers read as ``index[:K]`` then sort ascending (internvl2.py:104-124, eagle2_5.py:99-128).
Differences from the reference, both deliberate: ties in the score order are broken by lower frame position
(``vitg_rank``; the reference's torch.sort is unstable), and the append holds an ``flock`` (the reference lets 8 ranks
append to one file unsynchronised, videoitg.py:270,318-319).  ``num_frames`` keeps the reference's meaning (number of
visuals, not frames, videoitg.py:312).
"""
from __future__ import annotations

import fcntl
import json
import os
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from ...constants import DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX
from ...mm_utils import get_model_name_from_path, get_seq_frames, pad_sequence, tokenizer_image_token
from ...model.builder import load_pretrained_model

try:                                   # the harness is not installed in this image; the contract is duck-typed
    from lmms_eval.api.model import lmms as _Base
    from lmms_eval.api.registry import register_model
except Exception:                      # pragma: no cover
    class _Base:                        # minimal stand-in for lmms_eval.api.model.lmms
        def __init__(self):
            self._rank, self._world_size = 0, 1

    def register_model(*names):
        def deco(cls):
            return cls
        return deco


def load_video(video_path: str, num_frames: int = 64, target_fps: float = 1):
    """videoitg.py:95-130: decord first, PyAV fallback on ANY failure of the first reader (the reference's bare `except:` - an AVI that carries a
    real codec makes the built-in reader raise ValueError, a broken file makes decord raise DECORDError; PyAV gets its chance either way);
    returns (list of PIL images, sampled frame ids).  If PyAV is not installed either, the FIRST reader's error is the one raised."""
    from PIL import Image
    first_error = None
    try:
        from ...mm_utils import open_video_reader                # decord; without it, the built-in reader for .avi (MJPEG) / .y4m
        vr = open_video_reader(video_path, num_threads=1)
        idx = get_seq_frames(int(len(vr)), vr.get_avg_fps(), target_fps, num_frames)
        frames = vr.get_batch(idx).asnumpy()
        return [Image.fromarray(f) for f in frames], idx
    except Exception as e:                                        # noqa: BLE001 - as the reference: whatever the first reader raised
        first_error = e
    try:
        import av
    except ImportError:
        raise first_error
    container = av.open(video_path)
    stream = container.streams.video[0]
    fps = float(stream.average_rate)
    total = stream.frames
    if not total or "webm" in video_path or "mkv" in video_path:
        total = sum(1 for p in container.demux(video=0) for _ in p.decode())
        container.seek(0)
    idx = get_seq_frames(int(total), fps, target_fps, num_frames)
    want = set(idx)
    frames = [fr.to_ndarray(format="rgb24") for i, fr in enumerate(container.decode(video=0)) if i in want]
    container.close()
    return [Image.fromarray(f) for f in frames], idx


@register_model("videoitg")
class VideoITG(_Base):
    def __init__(self, pretrained: str = "nvidia/VideoITG-8B", truncation: Optional[bool] = True, device: Optional[str] = "cuda",
                 dtype: Optional[Union[str, torch.dtype]] = "auto", batch_size: Optional[Union[int, str]] = 1,
                 use_flash_attn=True, device_map="", conv_template="qwen_1_5", use_cache=True, truncate_context=False,
                 num_frames: int = 512, target_fps: int = 1, multiple: int = 1,
                 output_dir="./longvideobench_val_v_grounding_patchv1", video_loader: Optional[Callable] = None, **kwargs) -> None:
        super().__init__()
        # process group / rank handling as in the reference (videoitg.py:154-214): an accelerate Accelerator owns the process group,
        # one model per GPU, every rank waits before loading, and with several processes the model goes through
        # accelerator.prepare_model(evaluation_mode=True) (no DDP wrapper in evaluation mode; the model object survives it,
        # tests/test_gpu_graphs.py::test_model_survives_accelerate_prepare_model)
        from datetime import timedelta
        from accelerate import Accelerator, InitProcessGroupKwargs
        accelerator = Accelerator(kwargs_handlers=[InitProcessGroupKwargs(timeout=timedelta(weeks=52))])
        if accelerator.num_processes > 1 and device_map == "":   # videoitg.py:156-158
            self._device = torch.device(f"cuda:{accelerator.local_process_index}")
            self.device_map = f"cuda:{accelerator.local_process_index}"
        else:
            self._device = torch.device(device)
            self.device_map = device_map or str(self._device)
        self.pretrained = self.model_path = pretrained
        self.model_name = get_model_name_from_path(pretrained)
        self.num_frames, self.target_fps, self.multiple = num_frames, target_fps, multiple
        self.output_dir = output_dir
        accelerator.wait_for_everyone()                           # videoitg.py:170
        self._tokenizer, self._model, self.image_processor, self._max_length = load_pretrained_model(
            self.model_path, None, self.model_name, device_map=self.device_map, use_flash_attn=use_flash_attn)
        self._config = self._model.config
        self.model.eval()
        self.model.tie_weights()
        if accelerator.num_processes > 1:                         # videoitg.py:189-208 (DDP branch; FSDP / DeepSpeed are training set-ups)
            self._model = accelerator.prepare_model(self.model, evaluation_mode=True)
            self._rank, self._world_size = accelerator.local_process_index, accelerator.num_processes
        else:
            self._rank, self._world_size = 0, 1
        self.accelerator = accelerator
        self.truncation, self.batch_size_per_gpu = truncation, int(batch_size)
        self.conv_template, self.use_cache, self.truncate_context = conv_template, use_cache, truncate_context
        self._video_loader = video_loader or load_video
        self.task_dict = {}

    # ---- properties of the lmms contract (lmms_eval/api/model.py) ---------------------------------------
    config = property(lambda self: self._config)
    tokenizer = property(lambda self: self._tokenizer)
    model = property(lambda self: self._model)
    eot_token_id = property(lambda self: self.tokenizer.eos_token_id)
    max_length = property(lambda self: self._max_length)
    batch_size = property(lambda self: self.batch_size_per_gpu)
    device = property(lambda self: self._device)
    rank = property(lambda self: self._rank)
    world_size = property(lambda self: self._world_size)

    @staticmethod
    def flatten(items):
        out = []
        for i in items:
            out.extend(i) if isinstance(i, (list, tuple)) else out.append(i)
        return out

    def score(self, contexts: str, visuals):
        """One document: -> (frame ids in descending score order, scores rounded to 2 dp, #visuals).

        Every visual of the document is encoded and the whole list is handed to the model as `images`, exactly as
        /root/reference/lmms_eval/models/videoitg.py:276-301 does.  The prompt holds ONE `<image>`, so the model scores
        `images[0]` (eagle_archv1.py consumes one entry per `<image>`; grounding_qwen2.py:153 reshapes by `images[0].shape[0]`),
        while the frame ids come from the LAST video loaded (the reference's `sampled_frames` is the loop variable that survives
        the loop).  For the one-video documents of every real task the two coincide; for a multi-visual document this reproduces
        the reference's result, including its IndexError when the first visual has more frames than the last video sampled."""
        from PIL import Image
        from ...torch_ops import K as ops
        if not isinstance(visuals, (list, tuple)):
            visuals = [visuals]
        videos, sampled = [], None
        for visual in visuals:
            if isinstance(visual, Image.Image):
                frames = [visual]
                if sampled is None:
                    sampled = [0]                       # (the reference has no frame list for a bare image: a one-image document maps to id 0)
            else:
                frames, sampled = self._video_loader(visual, self.num_frames, self.target_fps)
            videos.append(self.image_processor.preprocess(frames, return_tensors="pt")["pixel_values"].half().to(self.device))
        prompt = DEFAULT_IMAGE_TOKEN + contexts + "\n"                                    # videoitg.py:289
        ids = [tokenizer_image_token(prompt, self.tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt")]
        pad_id = self.tokenizer.pad_token_id if self.tokenizer.pad_token_id is not None else self.tokenizer.eos_token_id
        ids = pad_sequence(self.tokenizer, ids, batch_first=True, padding_value=pad_id).to(self.device)
        with torch.inference_mode():
            resp = self.model(ids, attention_mask=ids.ne(pad_id), images=videos)
            scores, order = ops.rank(resp.logits[0].view(-1).float())
        order = order.cpu().tolist()
        # the reference's model is .half(): its logit leaves out_proj as fp16 and the sigmoid is evaluated on that fp16 value and rounded to
        # fp16 again before .tolist() (videoitg.py:301-307) - the 2-dp figures in results.jsonl are roundings of THOSE values.  The ORDER
        # stays the stable fp32 one (ties of the fp16 scores are broken by the fp32 score, then by frame position).
        lg16 = resp.logits[0].view(-1).float().cpu().half()
        s16 = torch.sigmoid(lg16.float()).half().float()[order].tolist()
        return [int(sampled[i]) for i in order], [round(v, 2) for v in s16], len(videos)

    def generate_until(self, requests) -> List[dict]:
        os.makedirs(self.output_dir, exist_ok=True)
        path = os.path.join(self.output_dir, "results.jsonl")
        res = []
        for contexts, gen_kwargs, doc_to_visual, doc_id, task, split in [r.args for r in requests]:
            visuals = self.flatten([doc_to_visual(self.task_dict[task][split][doc_id])])
            index, logits, n_vis = self.score(contexts, visuals)
            visual = visuals[-1]                                         # videoitg.py:315: the loop variable after the loop
            out = {"index": index, "logits": logits, "num_frames": n_vis, "contexts": contexts,
                   "video_path": visual if isinstance(visual, str) else None, "doc_id": doc_id}
            with open(path, "a", encoding="utf-8") as f:               # all ranks share one file: serialise the append
                fcntl.flock(f, fcntl.LOCK_EX)
                f.write(json.dumps(out, ensure_ascii=False) + "\n")
                f.flush()
                fcntl.flock(f, fcntl.LOCK_UN)
            res.append(out)
        return res

    def loglikelihood(self, requests):
        raise NotImplementedError("VideoITG scores frames; it has no token likelihoods (videoitg.py:330-331)")

    def generate_until_multi_round(self, requests):
        raise NotImplementedError("multi-round generation is not implemented for the frame selector (videoitg.py:356-357)")
