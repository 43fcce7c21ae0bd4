"""``EagleQwenG`` twin: the model object of the operator surface (SURVEY.md §8b "Model object").

Same call contract as /root/reference/eagle/model/language_model/grounding_qwen2.py:69-181:

    out = model(input_ids, attention_mask=attention_mask, images=[Tensor[T,3,H,W], ...])
    out.logits            # python list of B tensors [T_b, 1] (raw logits; caller applies sigmoid)

and it tolerates the chaining the reference callers do (``.half().eval().to(device)``, infer.py:21;
``accelerator.prepare_model(model, evaluation_mode=True)``, lmms_eval/models/videoitg.py:205).  It is an
``nn.Module`` only so that those wrappers accept it; it holds no parameters — the arithmetic runs in
libvideoitg_hip.so through ``SelectorEngine`` and there is no fallback when that library is absent.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from ..config import ITGConfig
from ..engine import PackedWeights, SelectorEngine
from ..image_processor import SiglipImageProcessor

EagleQwenGConfig = ITGConfig


@dataclass
class SelectorOutput:
    """Field-compatible with the CausalLMOutputWithPast the reference returns (grounding_qwen2.py:175-181)."""
    logits: List[torch.Tensor]
    loss: Optional[torch.Tensor] = None
    past_key_values: Any = None
    hidden_states: Any = None
    attentions: Any = None

    def __getitem__(self, k):
        return getattr(self, k) if isinstance(k, str) else (self.loss, self.logits)[k]


class _VisionTowerHandle:
    """What callers touch on ``model.get_vision_tower()`` (builder.py:163-168, videoitg.py:179)."""

    def __init__(self, cfg: ITGConfig):
        self.config = cfg.vision
        self.vision_tower_name = cfg.mm_vision_tower
        self.image_processor = SiglipImageProcessor(size=cfg.vision.image_size)
        self.is_loaded = True
        self.select_layer = cfg.mm_vision_select_layer
        self.select_feature = cfg.mm_vision_select_feature

    def load_model(self, device_map=None):
        return None

    def to(self, *a, **k):
        return self

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return self.config.num_patches

    @property
    def num_patches_per_side(self):
        return self.config.grid

    @property
    def image_size(self):
        return self.config.image_size


class EagleQwenG(nn.Module):
    config_class = EagleQwenGConfig

    def __init__(self, config: ITGConfig, state_dict: Dict[str, torch.Tensor], vision_state_dict: Dict[str, torch.Tensor],
                 torch_dtype: torch.dtype = torch.float16, device="cuda"):
        super().__init__()
        if config.mm_projector_type != "seq_mlp":
            raise ValueError(f"Unknown projector type: {config.mm_projector_type}")       # multimodal_projector/builder.py:69
        if config.mm_use_4_vision_tokens:
            raise ValueError("mm_use_4_vision_tokens=True (2x2 token merge) is not part of the VideoITG selector path; "
                             "the released config carries False (train_itg.py:1358)")
        if config.mm_vision_select_feature != "patch":
            raise ValueError(f"Unexpected select feature: {config.mm_vision_select_feature}")  # clip_encoder.py:128
        self.config = config
        self._src = (state_dict, vision_state_dict)
        self._dtype = torch_dtype
        self._device = torch.device(device)
        self._engine: Optional[SelectorEngine] = None
        self._tower = _VisionTowerHandle(config)
        # opt-in: replay single-video forwards from captured HIP graphs keyed by (frames, resolution, prompt bucket)
        # (engine.GraphCache; BASELINE.json configs[3]).  VITG_GRAPHS=1 or model.use_graphs = True.
        self.use_graphs = os.environ.get("VITG_GRAPHS", "0") not in ("", "0")
        # opt-in fp8 weight path (BASELINE.json configs[4]): tower / decoder GEMM weights as e4m3fn + per-channel scales, per-token
        # fp8 activations into those GEMMs.  VITG_FP8=1 or model.fp8 = True before the first forward.
        self.fp8 = os.environ.get("VITG_FP8", "0") not in ("", "0")
        # opt-in operand-split precision tier (round 6, csrc/split.hip): every MFMA operand as an fp16 (hi, lo) pair - the tier that holds north_star's
        # 1e-3 against the fp32 CPU reference on checkpoints with massive-activation channels, at ~2.5x the step time.  VITG_OPERAND_SPLIT=1,
        # load_pretrained_model(..., operand_split=True) or model.operand_split = True before the first forward.
        self.operand_split = os.environ.get("VITG_OPERAND_SPLIT", "0") not in ("", "0")
        self._graphs = None

    # ---- chaining the callers do ---------------------------------------------------------------
    def _set(self, dtype=None, device=None):
        if dtype is not None and dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f"the selector kernels take fp16 or bf16 MFMA inputs, not {dtype}")
        if device is not None:
            device = torch.device(device)
            if device.type == "cuda" and device.index is None:      # "cuda" names the current device: keep the index we were given
                device = torch.device("cuda", self._device.index if self._device.type == "cuda" and self._device.index is not None else 0)
        changed = (dtype is not None and dtype != self._dtype) or (device is not None and device != self._device)
        if dtype is not None:
            self._dtype = dtype
        if device is not None:
            self._device = device
        if changed and self._engine is not None:
            if self._src is None:
                raise RuntimeError("weights were already packed and the source state dict was released; "
                                   "set dtype/device before the first forward or load with keep_source=True")
            self._engine = None
        return self

    def half(self):
        return self._set(dtype=torch.float16)

    def bfloat16(self):
        return self._set(dtype=torch.bfloat16)

    def to(self, *args, **kwargs):
        dtype = kwargs.get("dtype")
        device = kwargs.get("device")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif isinstance(a, (str, torch.device)):
                device = a
        return self._set(dtype=dtype, device=device)

    def cuda(self, device=None):
        return self._set(device="cuda" if device is None else device)

    def eval(self):
        return super().eval()

    def tie_weights(self):
        return None

    def requires_grad_(self, requires_grad: bool = False):
        return self

    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        return self._dtype

    def get_model(self):
        return self

    def get_vision_tower(self):
        return self._tower

    # ---- engine ----------------------------------------------------------------------------------
    def engine(self, keep_source: bool = False) -> SelectorEngine:
        if self._engine is None:
            if self._device.type != "cuda":
                raise RuntimeError("the VideoITG selector forward runs on an MI355X (device 'cuda:k' under ROCm); "
                                   "there is no CPU path")
            sd, vsd = self._src
            with torch.cuda.device(self._device):
                pw = PackedWeights.from_state_dicts(self.config, sd, vsd, self._dtype, self._device, fp8=self.fp8,
                                                    vit_q_prescaled=False if self.operand_split else None)
                self._engine = SelectorEngine(pw, operand_split=self.operand_split)
            if not keep_source:
                self._src = None
        return self._engine

    @torch.no_grad()
    def forward(self, input_ids: torch.LongTensor = None, attention_mask: Optional[torch.Tensor] = None,
                position_ids=None, past_key_values=None, inputs_embeds=None, labels=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, images=None, image_sizes=None, return_dict=None,
                modalities=None, dpo_forward=False, cache_position=None, grounding_labels=None, token_types=None,
                **_ignored) -> SelectorOutput:
        if images is None or input_ids is None:
            raise ValueError("the selector forward needs input_ids and images (grounding_qwen2.py:107-108)")
        if dpo_forward or inputs_embeds is not None:
            raise NotImplementedError("dpo / inputs_embeds branches are outside the inference hot path")
        eng = self.engine()
        with torch.cuda.device(self._device):
            images = list(images)
            single = len(images) == 1 and (input_ids.dim() == 1 or input_ids.shape[0] == 1) and \
                (attention_mask is None or bool(attention_mask.all()))
            # (never in training mode: the captured forward pools with the eval rule and would not consume the reference's random-grid draws)
            if self.use_graphs and single and grounding_labels is None and not self.training and not self.operand_split:
                if self._graphs is None or self._graphs.eng is not eng:
                    from ..engine import GraphCache
                    self._graphs = GraphCache(eng)
                im = images[0] if images[0].dim() == 4 else images[0].unsqueeze(0)
                ids_host = input_ids.reshape(-1).cpu()                # ONE device-to-host copy per call; the cache works on the host copy
                # a document the capture cannot serve - no / several <image> tokens, or a prompt whose 32-token BUCKET passes
                # tokenizer_model_max_length although the prompt itself may fit - is scored by the eager forward, which splices and
                # truncates like the reference (eagle_archv1.py:341-345), instead of aborting the run
                if self._graphs.serves(ids_host, im):
                    lg = self._graphs(ids_host, im.to(self._device, self._dtype))
                    return SelectorOutput(logits=[lg.reshape(-1, 1)])
            grids = None
            if self.training:
                # mlp_proj.py:51-52: in training mode the projector pools every video to a RANDOM grid, drawn with torch's global generator in
                # list order - the same calls here give the same grids under the same seed (forward values only: there is no backward)
                import math
                cfg = self.config
                G0 = (images[0].shape[-1] if images else 0) // cfg.vision.patch_size
                grids = []
                for im in images:
                    T_i = im.shape[0] if im.dim() == 4 else 1
                    hw = math.floor((cfg.vision_token_num / T_i) ** 0.5)
                    grids.append(min(torch.randint(cfg.vision_min_num, hw + 1, (1,)).item(), G0))
            logits = eng.forward(input_ids, attention_mask, images, pooled_grids=grids)
            if grounding_labels is None:
                return SelectorOutput(logits=logits)
            # labelled evaluation (grounding_qwen2.py:159-181): the loss VALUE (no backward here), and - as in the reference -
            # `logits` becomes the flat fp32 concatenation over the batch instead of the per-video list
            from ..torch_ops import K as ops
            flat = torch.cat([l.float().view(-1) for l in logits])
            lab = torch.cat([torch.as_tensor(g).view(-1) for g in grounding_labels]).to(flat.device, torch.float32)
            loss, _ = ops.bce_loss(flat, lab)
        return SelectorOutput(logits=flat, loss=loss)
