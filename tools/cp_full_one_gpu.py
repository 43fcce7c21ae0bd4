#!/usr/bin/env python3
"""BASELINE.json configs[2] at ITS OWN workload on a one-GPU box: `world` processes share cuda:0 and exchange over gloo (RCCL refuses
two ranks on one device), each builds the FULL model (26 SigLIP + 28 Qwen2 layers, real widths, counter-generated weights) and runs
`ContextParallel.forward` on its frame shard of a complete-pass oracle fixture (tests/golden/full_<case>.json, default the headline
512 x 448^2 video).  What meets the oracle here for the first time are the kernels only a rank of an N-GPU job launches: M = 1600 / 1660
row GEMMs on the 256-wide tiles with ragged last row tiles, split-K o / down projections, the K|V-first / Q-only projections, 1 6xx
queries against 12 860 gathered keys, 64-frame tower shards pooled with the GLOBAL grid.

    python tools/cp_full_one_gpu.py [world=8] [case=c2_512x448] [fp8 policy, e.g. llm.gu,llm.down]

Memory: world x (15.2 GB of weights + that rank's workspace) - 8 ranks ~ 140 GB of the 288 GB.  Timings of such a run mean nothing (the
ranks time-share one GPU); `tools/cp_rank_time.py` measures one rank's compute."""
import json
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_case(case: str) -> dict:
    with open(os.path.join(ROOT, "tests", "golden", f"full_{case}.json")) as f:
        return json.load(f)


def parse_fp8(fp8):
    if fp8 in (None, "", False, "0", "none"):
        return False
    if fp8 in (True, "1", "all"):
        return True
    return fp8


def worker(rank: int, world: int, port: int, case: str, fp8, q, split=False) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from videoitg_amd import synth
        from videoitg_amd.config import ITGConfig
        from videoitg_amd.engine import SelectorEngine
        from videoitg_amd.parallel import ContextParallel
        from videoitg_amd.synth_device import packed_random_weights
        rec = load_case(case)
        cfg = ITGConfig()
        dev = torch.device("cuda:0")
        # split: the operand-split precision tier (SelectorEngine(operand_split=True); K|V travel as (hi, lo) pairs) - 45 GB of weights per rank
        eng = SelectorEngine(packed_random_weights(cfg, torch.float16, dev, seed=rec["weight_seed"], fp8=parse_fp8(fp8), plan=synth.fixture_plan(rec),
                                                   vit_q_prescaled=False if split else None), operand_split=bool(split))
        ids, pix = synth.fixture_inputs(cfg, rec)
        T = rec["frames"]
        cp = ContextParallel(rank, world)
        fr = cp.local_frames(T)
        local = pix[fr.start:fr.stop].to(torch.float16).to(dev)          # every rank holds ONLY its own frames in HBM
        del pix
        out = cp.forward(eng, ids, local, total_frames=T)
        torch.cuda.synchronize()
        if rank == 0:
            # which launch forms this rank took (the point of the test is that the few-tile / split forms ran): plans cached by the engine
            forms = {"splitk": sorted({(k, v[1]) for k, v in eng._splitk.items() if v[0] == "split" and v[1] > 1}),
                     "n_local": cp._gatherers[next(iter(cp._gatherers))].plan.n_local}
            q.put((out.float().cpu().tolist(), forms))     # by value: a tensor would travel as a shared-memory fd that dies with this process
        dist.barrier()
    finally:
        dist.destroy_process_group()


def run(world: int = 8, case: str = "c2_512x448", fp8=None, port: int = 29561, timeout: int = 900, split: bool = False):
    """-> (logits fp32 [T] gathered on rank 0, {"splitk": [((M, N, K), splits), ...], "n_local": rows per rank})"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, case, fp8, q, split)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got, forms = q.get(timeout=timeout)
        got = torch.tensor(got, dtype=torch.float32)
    finally:
        for p in procs:
            p.join(timeout)
            if p.is_alive():
                p.kill()
    for p in procs:
        assert p.exitcode == 0, p.exitcode
    return got, forms


def single_gpu_logits(case: str = "c2_512x448", fp8=None) -> torch.Tensor:
    from videoitg_amd import synth
    from videoitg_amd.config import ITGConfig
    from videoitg_amd.engine import SelectorEngine
    from videoitg_amd.synth_device import packed_random_weights
    rec = load_case(case)
    cfg = ITGConfig()
    dev = torch.device("cuda:0")
    eng = SelectorEngine(packed_random_weights(cfg, torch.float16, dev, seed=rec["weight_seed"], fp8=parse_fp8(fp8), plan=synth.fixture_plan(rec)))
    ids, pix = synth.fixture_inputs(cfg, rec)
    out = eng.forward(ids, None, [pix.to(torch.float16).to(dev)])[0].view(-1).float().cpu()
    torch.cuda.synchronize()
    eng.release_workspace()
    del eng
    torch.cuda.empty_cache()
    return out


def report(case: str, got: torch.Tensor, single=None) -> dict:
    rec = load_case(case)
    want = torch.tensor(rec["logits"])
    order = torch.sort(torch.sigmoid(got), descending=True, stable=True).indices.tolist()
    k = rec["k"]
    out = {"case": case, "max_abs_dscore_vs_oracle": (torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item(),
           "max_abs_dlogit_vs_oracle": (got - want).abs().max().item(), f"top{k}_set_match": sorted(order[:k]) == rec["topk"],
           f"top{k}_kept": len(set(order[:k]) & set(rec["topk"]))}
    if single is not None:
        out["max_abs_dlogit_vs_single_gpu"] = (got - single).abs().max().item()
        out["max_abs_dscore_vs_single_gpu"] = (torch.sigmoid(got) - torch.sigmoid(single)).abs().max().item()
    return out


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    case = sys.argv[2] if len(sys.argv) > 2 else "c2_512x448"
    fp8 = sys.argv[3] if len(sys.argv) > 3 else None
    got, forms = run(world, case, fp8)
    single = single_gpu_logits(case, fp8)
    rep = report(case, got, single)
    rep.update(world=world, fp8=fp8 or False, rows_per_rank=forms["n_local"], splitk_shapes=[list(s[0]) + [s[1]] for s in forms["splitk"]])
    print(json.dumps(rep), flush=True)
