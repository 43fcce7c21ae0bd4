#!/usr/bin/env python3
"""Context-parallel forward with WORLD > 1 on a single GPU box: `world` processes share cuda:0 and talk over gloo (RCCL refuses
two ranks on one device), so the real kernels run with the multi-rank plan (frame shards, global positions, ragged K|V
all-gather + compaction, logits gather) and the result is compared with the single-rank forward.
    python tools/cp_two_ranks_one_gpu.py [world] [frames]"""
import os
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(fp8=False, split=False):
    from videoitg_amd import synth
    from videoitg_amd.config import tiny_config
    from videoitg_amd.engine import PackedWeights, SelectorEngine
    cfg = tiny_config(vision_token_num=72)
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    eng = SelectorEngine(PackedWeights.from_state_dicts(cfg, sd, vsd, torch.float16, "cuda:0", fp8=fp8, vit_q_prescaled=False if split else None),
                         operand_split=split)
    return cfg, eng


def worker(rank, world, frames, port, q, fp8=False, split=False, prompt=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from videoitg_amd import synth
        from videoitg_amd.parallel import ContextParallel
        cfg, eng = build(fp8, split)
        pix = synth.synth_pixel_values(frames, cfg.vision.image_size, seed=1234).cuda()
        ids = torch.tensor([prompt or PROMPT])
        cp = ContextParallel(rank, world)
        fr = cp.local_frames(frames)
        local = pix[fr.start:fr.stop].clone()                 # every rank hands over ONLY its own frames
        first = cp.forward(eng, ids, local, total_frames=frames)
        out = cp.forward(eng, ids, local, total_frames=frames)  # second call: cached gatherer (buffers + compaction index reused)
        assert torch.equal(first, out) and len(cp._gatherers) == 1
        whole = cp.forward(eng, ids, pix)                      # the whole video resident: same result
        assert torch.equal(whole, out)
        torch.cuda.synchronize()
        if rank == 0:
            q.put(out.float().cpu().tolist())      # by value: a tensor would travel as a shared-memory fd that dies with this process
        dist.barrier()
    finally:
        dist.destroy_process_group()


PROMPT = [11, 12, -200, 5, 6, 7]          # text on both sides of <image>: rank 0 owns the rows before it, the last rank the rows after it


def run(world=2, frames=9, port=29533, fp8=False, split=False, prompt=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, frames, port, q, fp8, split, prompt)) for r in range(world)]
    for p in procs:
        p.start()
    got = torch.tensor(q.get(timeout=600), dtype=torch.float32)
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, p.exitcode
    from videoitg_amd import synth
    cfg, eng = build(fp8, split)
    pix = synth.synth_pixel_values(frames, cfg.vision.image_size, seed=1234).cuda()
    ids = torch.tensor([prompt or PROMPT])
    want = eng.forward(ids, None, [pix])[0].view(-1).float().cpu()
    return got, want



# ---- several cases through ONE set of rank processes (each spawn re-imports torch and rebuilds a model: 4-15 s per case on a cold box) ---------------
def worker_many(rank, world, cases, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from videoitg_amd import synth
        from videoitg_amd.parallel import ContextParallel
        engines, outs = {}, []
        for c in cases:
            key = (bool(c.get("fp8")), bool(c.get("split")))
            if key not in engines:
                engines[key] = build(*key)
            cfg, eng = engines[key]
            frames = c["frames"]
            pix = synth.synth_pixel_values(frames, cfg.vision.image_size, seed=1234).cuda()
            ids = torch.tensor([c.get("prompt") or PROMPT])
            cp = ContextParallel(rank, world)
            fr = cp.local_frames(frames)
            local = pix[fr.start:fr.stop].clone()
            first = cp.forward(eng, ids, local, total_frames=frames)
            out = cp.forward(eng, ids, local, total_frames=frames)          # second call: cached gatherer
            assert torch.equal(first, out) and len(cp._gatherers) == 1
            assert torch.equal(cp.forward(eng, ids, pix), out)              # the whole video resident: same result
            torch.cuda.synchronize()
            outs.append(out.float().cpu().tolist())
        if rank == 0:
            q.put(outs)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def run_many(world, cases, port=29570):
    """cases: [{"frames": T, "prompt": [...] | None, "fp8": bool, "split": bool}, ...] -> [(got, want), ...]: every case through the same `world` processes,
    `want` = the single-rank forward of the same engine in this process"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker_many, args=(r, world, cases, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = q.get(timeout=900)
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, p.exitcode
    from videoitg_amd import synth
    engines, res = {}, []
    for c, got in zip(cases, outs):
        key = (bool(c.get("fp8")), bool(c.get("split")))
        if key not in engines:
            engines[key] = build(*key)
        cfg, eng = engines[key]
        pix = synth.synth_pixel_values(c["frames"], cfg.vision.image_size, seed=1234).cuda()
        ids = torch.tensor([c.get("prompt") or PROMPT])
        want = eng.forward(ids, None, [pix])[0].view(-1).float().cpu()
        res.append((torch.tensor(got, dtype=torch.float32), want))
    return res

if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    got, want = run(world, frames, fp8=len(sys.argv) > 3 and sys.argv[3] == "fp8", split=len(sys.argv) > 3 and sys.argv[3] == "split")
    d = (got - want).abs().max().item()
    print(f"world={world} frames={frames}: max|dlogit| vs single rank = {d:.3e}", flush=True)
    assert d < 5e-3, d
