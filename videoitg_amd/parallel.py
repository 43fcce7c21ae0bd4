"""One video over N GPUs: frame-sharded tower/projector + context-parallel bidirectional LLM.

The reference has no intra-video parallelism (one model per GPU, whole videos per rank:
/root/reference/lmms_eval/models/videoitg.py:154-177, lmms_eval/utils.py:689-695).  Sharding the *model call* by
frames would change the scores (every visual token attends to every other one and to the prompt, and the pooled
grid depends on the global frame count, mlp_proj.py:48), so the split that preserves results is:

* tower + pool + projector: rank r owns frames [f0_r, f1_r); the pooled grid uses the GLOBAL T; no collective;
* LLM: rank r owns the token rows of its frames (+ the text before <image> on rank 0, the text after it on the
  last rank); weights replicated; per layer ONE exchange — an all-gather of the local K|V rows (GQA: 2*4*128
  values per token) — then local queries attend to all keys (non-causal => perfectly balanced, no zig-zag);
  RoPE uses global positions;
* score head: local frames -> all-gather of the T logits -> every rank ranks them.

The operand-split precision tier (engine.llm_forward_split) shards the same way; its exchange carries (hi, lo) PAIRS of the K|V rows through the blocking
form of the gather.

``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests) is the only
communication layer.  The planning and gather code is device-agnostic so the world_size-2 gloo tests exercise the
same code that runs on the GPUs.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.distributed as dist

from .constants import IMAGE_TOKEN_INDEX


@dataclass
class ShardPlan:
    world: int
    frames: List[range]          # frames owned by each rank
    n_pre: int                   # text tokens before <image> (rank 0)
    n_post: int                  # text tokens after <image> (last rank)
    tokens_per_frame: int
    n_local: List[int]           # LLM rows per rank
    offset: List[int]            # global position of each rank's first row
    total: int

    def vis_start(self, rank: int) -> int:
        """local row index of the first visual token on `rank`"""
        return self.n_pre if rank == 0 else 0


def make_plan(world: int, T: int, tokens_per_frame: int, n_pre: int, n_post: int) -> ShardPlan:
    base, extra = divmod(T, world)
    frames, f = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        frames.append(range(f, f + n))
        f += n
    n_local = [len(fr) * tokens_per_frame for fr in frames]
    n_local[0] += n_pre
    n_local[-1] += n_post
    offset, acc = [], 0
    for n in n_local:
        offset.append(acc)
        acc += n
    return ShardPlan(world, frames, n_pre, n_post, tokens_per_frame, n_local, offset, acc)


class RowGatherer:
    """all-gather of a ragged row-sharded matrix: rank r contributes rows [offset_r, offset_r + n_r).

    Two ways to use it.  ``gather(local)`` copies `local` into the send slot and gathers (what the CPU tests and round 2 used).  The
    engine's context-parallel layers use the split form so that the exchange runs UNDER compute (SURVEY.md 8e "overlap it with the
    Q-projection"): ``rows = local_rows(c, dtype, device)`` is this rank's slot INSIDE the receive buffer - the K|V projection writes its
    output there (no staging copy; the collective is the in-place form send == recv[rank]) - then ``w = start()`` launches the
    all-gather (RCCL: asynchronous, ordered after the work already queued on the current stream) and ``finish(w)`` makes the current
    stream wait for it and returns all rows (ragged shards: ONE gather-rows launch compacts the padded receive buffer)."""

    launched = 0          # exchanges started in this process (bench.py: a failure AFTER the first one cannot be agreed upon with collectives any more)

    def __init__(self, plan: ShardPlan, rank: int, group=None):
        self.plan, self.rank, self.group = plan, rank, group
        self.n_max = max(plan.n_local)
        self.even = len(set(plan.n_local)) == 1
        self._recv = self._out = self._idx = None
        self._async = None

    def _ensure(self, c: int, dtype, device) -> None:
        p = self.plan
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:           # an index-less "cuda" never equals a tensor's "cuda:0" (that
            device = torch.device("cuda", torch.cuda.current_device())   # re-allocated the buffers and dropped the compaction index every layer)
        if self._recv is None or self._recv.shape[1] != c or self._recv.dtype != dtype or self._recv.device != device:
            self._recv = torch.zeros((p.world * self.n_max, c), dtype=dtype, device=device)
            self._out = self._recv if self.even else torch.empty((p.total, c), dtype=dtype, device=device)
            self._idx = None

    def local_rows(self, c: int, dtype, device) -> torch.Tensor:
        """[n_local, c] view of this rank's slot in the receive buffer: write the local rows here, then start()"""
        self._ensure(c, dtype, device)
        return self._recv[self.rank * self.n_max:self.rank * self.n_max + self.plan.n_local[self.rank]]

    def _slot(self) -> torch.Tensor:
        return self._recv[self.rank * self.n_max:(self.rank + 1) * self.n_max]

    def start(self):
        """launch the exchange; returns a handle for finish().  RCCL ("nccl") runs it asynchronously on its own stream, ordered behind
        the current stream; other backends (gloo in the CPU / one-GPU tests) exchange in finish()."""
        if self._async is None:
            self._async = dist.get_backend(self.group) == "nccl"
        RowGatherer.launched += 1
        if self._async:
            return dist.all_gather_into_tensor(self._recv, self._slot(), group=self.group, async_op=True)
        return None

    def finish(self, work, producer_stream=None) -> torch.Tensor:
        if work is not None:
            work.wait()                                     # the CURRENT stream waits for the collective; the host does not block
        else:
            if producer_stream is not None and self._recv.is_cuda:
                torch.cuda.current_stream(self._recv.device).wait_stream(producer_stream)
            dist.all_gather_into_tensor(self._recv, self._slot(), group=self.group)
        return self._compact()

    def _compact(self) -> torch.Tensor:
        p = self.plan
        if self.even:
            return self._recv
        c = self._recv.shape[1]
        if self._recv.is_cuda:
            # ragged shards (the prompt tokens sit on the first / last rank): ONE gather-rows launch compacts the padded
            # receive buffer instead of `world` slice copies
            from .torch_ops import K as ops
            if self._idx is None:
                idx = torch.cat([torch.arange(p.n_local[r], dtype=torch.int32) + r * self.n_max for r in range(p.world)])
                self._idx = idx.to(self._recv.device)
            ops.gather_rows(self._recv, self._idx, 0, 1, self._out, c)
            return self._out
        for r in range(p.world):
            self._out[p.offset[r]:p.offset[r] + p.n_local[r]].copy_(self._recv[r * self.n_max:r * self.n_max + p.n_local[r]])
        return self._out

    def __call__(self, local: torch.Tensor) -> torch.Tensor:
        n, c = local.shape
        assert n == self.plan.n_local[self.rank]
        self.local_rows(c, local.dtype, local.device).copy_(local)
        return self.finish(self.start())


def gather_logits(local: torch.Tensor, plan: ShardPlan, rank: int, group=None) -> torch.Tensor:
    """per-rank frame logits -> all T logits on every rank, in frame order"""
    fmax = max(len(f) for f in plan.frames)
    send = torch.zeros(fmax, dtype=local.dtype, device=local.device)
    send[:local.numel()].copy_(local.reshape(-1))
    recv = torch.empty(plan.world * fmax, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    return torch.cat([recv[r * fmax:r * fmax + len(plan.frames[r])] for r in range(plan.world)])


class ContextParallel:
    def __init__(self, rank: int, world: int, group=None):
        self.rank, self.world, self.group = rank, world, group
        self._gatherers = {}             # plan key -> RowGatherer: send / receive buffers and the compaction index are built ONCE

    def local_frames(self, total_frames: int) -> range:
        """the frames this rank encodes (what a caller that shards the pixels itself has to hand to forward)"""
        base, extra = divmod(total_frames, self.world)
        start = self.rank * base + min(self.rank, extra)
        return range(start, start + base + (1 if self.rank < extra else 0))

    def forward(self, eng, input_ids: torch.Tensor, pix: torch.Tensor, total_frames: Optional[int] = None) -> torch.Tensor:
        """input_ids [1, L] (one <image>) -> logits fp32 [T] on every rank.  pix: either the whole video [T,3,R,R] (only the local
        frames are read) or, with total_frames = T, ONLY this rank's frames [len(local_frames(T)),3,R,R] - the pixels of the other
        ranks never need to be resident here."""
        from .torch_ops import K as ops
        cfg = eng.cfg
        ids = input_ids.reshape(-1).cpu()
        pos = (ids == IMAGE_TOKEN_INDEX).nonzero().flatten().tolist()
        if len(pos) != 1:
            raise ValueError("exactly one <image> token per sample")
        pre, post = ids[:pos[0]], ids[pos[0] + 1:]
        T = pix.shape[0] if total_frames is None else total_frames
        if T < self.world:
            # a rank without frames would have no rows to exchange (and, in the middle of the group, no text either): not a mode the path has - the
            # reference's only parallelism is one process per video.  Fail on every rank, before any collective.
            raise ValueError(f"{T} frames over {self.world} ranks: the context-parallel forward needs at least one frame per rank (use <= {T} ranks)")
        G = pix.shape[-1] // cfg.vision.patch_size
        hw = cfg.pooled_grid(T, G)                           # GLOBAL frame count (mlp_proj.py:48)
        plan = make_plan(self.world, T, hw * hw, pre.numel(), post.numel())
        fr = plan.frames[self.rank]
        if total_frames is None:
            local_pix = pix[fr.start:fr.stop]
        else:
            if pix.shape[0] != len(fr):
                raise ValueError(f"rank {self.rank} owns {len(fr)} of {T} frames, got {pix.shape[0]}")
            local_pix = pix
        feats, _, cmap, pc = eng.vit_forward(local_pix, hw)
        n = plan.n_local[self.rank]
        x = eng._buf("llm_x", (n, cfg.hidden_size), torch.float32)
        vs = plan.vis_start(self.rank)
        nv = len(fr) * hw * hw
        if self.rank == 0 and pre.numel():
            ops.embed_gather(pre.contiguous().to(eng.device), eng.w.embed, x[:pre.numel()])
        eng.project_into(feats, len(fr), G, hw, x[vs:vs + nv], cmap, pc)
        if self.rank == self.world - 1 and post.numel():
            ops.embed_gather(post.contiguous().to(eng.device), eng.w.embed, x[vs + nv:])
        key = (T, hw, pre.numel(), post.numel())
        gather = self._gatherers.get(key)
        if gather is None:
            gather = self._gatherers[key] = RowGatherer(plan, self.rank, self.group)
        hfin = eng.llm_forward(x, pos0=plan.offset[self.rank], total=plan.total, kv_gather=gather)
        local = torch.empty(len(fr), dtype=torch.float32, device=eng.device)
        ops.pool_score(hfin[vs:], len(fr), hw * hw, cfg.hidden_size, eng.w.head_w, eng.w.head_b, local)
        return gather_logits(local, plan, self.rank, self.group)
