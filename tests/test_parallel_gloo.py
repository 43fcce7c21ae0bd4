"""CPU, world_size 2, gloo: the context-parallel plan / K-V gather / logits gather of videoitg_amd.parallel (the code
that runs over RCCL on the GPUs) reproduces the single-process oracle when each rank computes only its own frames and
token rows.  The per-rank arithmetic is done with oracle functions (test-side stand-in for the HIP kernels)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from oracle import videoitg_oracle as O
from videoitg_amd import synth
from videoitg_amd.config import tiny_config
from videoitg_amd.parallel import RowGatherer, gather_logits, make_plan


def test_plan_partitions_rows_and_positions():
    for world, T, tpf, pre, post in [(2, 8, 9, 0, 5), (8, 512, 25, 0, 60), (3, 7, 4, 2, 3), (4, 5, 1, 1, 0)]:
        p = make_plan(world, T, tpf, pre, post)
        assert sum(len(f) for f in p.frames) == T and p.frames[0].start == 0 and p.frames[-1].stop == T
        assert p.total == T * tpf + pre + post == sum(p.n_local)
        assert p.offset[0] == 0 and all(p.offset[r + 1] == p.offset[r] + p.n_local[r] for r in range(world - 1))
        assert max(len(f) for f in p.frames) - min(len(f) for f in p.frames) <= 1
        # global position of the first visual token of rank r
        for r in range(world):
            assert p.offset[r] + p.vis_start(r) == pre + p.frames[r].start * tpf


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ids, T, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = tiny_config(vision_token_num=72)
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    pix = synth.synth_pixel_values(T, cfg.vision.image_size, seed=1234)
    v = cfg.vision
    G = v.grid
    hw = cfg.pooled_grid(T, G)                                        # GLOBAL T
    pos = (ids == -200).nonzero().flatten().tolist()[0]
    pre, post = ids[:pos], ids[pos + 1:]
    plan = make_plan(world, T, hw * hw, pre.numel(), post.numel())
    fr = plan.frames[rank]
    from videoitg_amd.parallel import ContextParallel
    assert ContextParallel(rank, world).local_frames(T) == fr          # what a caller that shards the pixels itself hands over
    feats = O.siglip_tower(pix[fr.start:fr.stop], vsd, patch=v.patch_size, num_heads=v.num_attention_heads,
                           num_layers=v.num_hidden_layers, select_layer=cfg.mm_vision_select_layer)
    pooled = O.bilinear_pool(feats, hw)
    pw = "model.mm_projector.out_projection."
    vis = F.linear(F.gelu(F.linear(pooled, sd[pw + "0.weight"], sd[pw + "0.bias"])), sd[pw + "2.weight"], sd[pw + "2.bias"]).flatten(0, 1)
    parts = []
    if rank == 0 and pre.numel():
        parts.append(sd["model.embed_tokens.weight"][pre])
    parts.append(vis)
    if rank == world - 1 and post.numel():
        parts.append(sd["model.embed_tokens.weight"][post])
    x = torch.cat(parts)
    assert x.shape[0] == plan.n_local[rank]
    nq, nkv, hd, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.hidden_size
    cos, sin = O.rope_tables(plan.total, hd, cfg.rope_theta)
    sl = slice(plan.offset[rank], plan.offset[rank] + x.shape[0])
    cos, sin = cos[sl], sin[sl]
    gather = RowGatherer(plan, rank)
    n = x.shape[0]
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        h = O.rms_norm(x, sd[p + "input_layernorm.weight"], cfg.rms_norm_eps)
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]).view(n, nq, hd)
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"]).view(n, nkv, hd)
        vv = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"]).view(n, nkv, hd)
        q = q * cos[:, None] + O.rotate_half(q) * sin[:, None]
        k = k * cos[:, None] + O.rotate_half(k) * sin[:, None]
        kv = gather(torch.cat([k.reshape(n, -1), vv.reshape(n, -1)], 1).contiguous())        # [total, 2*nkv*hd]
        ka = kv[:, :nkv * hd].view(1, plan.total, nkv, hd).transpose(1, 2)
        va = kv[:, nkv * hd:].view(1, plan.total, nkv, hd).transpose(1, 2)
        o = O.attention_noncausal(q.unsqueeze(0).transpose(1, 2), ka, va, None).transpose(1, 2).reshape(n, D)
        x = x + F.linear(o, sd[p + "self_attn.o_proj.weight"])
        h = O.rms_norm(x, sd[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        x = x + F.linear(F.silu(F.linear(h, sd[p + "mlp.gate_proj.weight"])) * F.linear(h, sd[p + "mlp.up_proj.weight"]),
                         sd[p + "mlp.down_proj.weight"])
    hfin = O.rms_norm(x, sd["model.norm.weight"], cfg.rms_norm_eps)
    vs = plan.vis_start(rank)
    local = F.linear(hfin[vs:vs + len(fr) * hw * hw].reshape(len(fr), hw * hw, D).mean(1), sd["out_proj.weight"], sd["out_proj.bias"]).view(-1)
    logits = gather_logits(local, plan, rank)
    if rank == 0:
        torch.save(logits, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ids,T", [(2, [-200, 5, 6, 7, 8, 9], 8), (2, [11, 12, -200, 5, 6], 7), (3, [11, -200, 5, 6, 7], 8)])
def test_context_parallel_matches_single_process(tmp_path, world, ids, T):
    ids_t = torch.tensor(ids)
    out = str(tmp_path / "logits.pt")
    mp.spawn(_worker, args=(world, _free_port(), ids_t, T, out), nprocs=world, join=True)
    got = torch.load(out)
    cfg = tiny_config(vision_token_num=72)
    sd, vsd = synth.synth_llm_state_dict(cfg, 0), synth.synth_vision_state_dict(cfg, 0)
    pix = synth.synth_pixel_values(T, cfg.vision.image_size, seed=1234)
    want = O.forward(cfg, sd, vsd, ids_t.unsqueeze(0), None, [pix])[0].view(-1)
    assert got.shape == want.shape
    assert (got - want).abs().max().item() < 2e-4


def _split_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = make_plan(world, 7, 3, 2, 4)                       # ragged: 7 frames / world ranks, text on the first and last rank
    g = RowGatherer(plan, rank)
    c = 6
    res = []
    for it in range(2):                                        # second pass: cached buffers, in-place collective again
        rows = g.local_rows(c, torch.float32, "cpu")
        assert rows.shape == (plan.n_local[rank], c)
        base = plan.offset[rank]
        rows.copy_((torch.arange(plan.n_local[rank] * c, dtype=torch.float32).view(-1, c) + 1000 * (base + it)))
        allr = g.finish(g.start())
        res.append(allr.clone())
        # and the copying form gives the same rows
        assert torch.equal(g(rows.clone()), allr)
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_gatherer_split_api_in_place(tmp_path, world):
    """local_rows() / start() / finish(): the producer writes straight into its slot of the receive buffer and the collective runs in
    place (what the context-parallel layers do so that the exchange overlaps the Q projection)"""
    out = str(tmp_path / "rows.pt")
    mp.spawn(_split_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = torch.load(out)
    plan = make_plan(world, 7, 3, 2, 4)
    for it, allr in enumerate(res):
        assert allr.shape == (plan.total, 6)
        for r in range(world):
            want = torch.arange(plan.n_local[r] * 6, dtype=torch.float32).view(-1, 6) + 1000 * (plan.offset[r] + it)
            assert torch.equal(allr[plan.offset[r]:plan.offset[r] + plan.n_local[r]], want)


def test_fewer_frames_than_ranks_is_refused_before_any_collective():
    """a rank without frames has no rows to exchange: ContextParallel.forward raises on every rank - no engine, no process group touched"""
    from videoitg_amd.parallel import ContextParallel
    cfg = tiny_config(vision_token_num=72)
    eng = type("E", (), {"cfg": cfg})()
    pix = torch.zeros((3, 3, cfg.vision.image_size, cfg.vision.image_size))
    for rank in (0, 2, 3):
        with pytest.raises(ValueError, match="at least one frame per rank"):
            ContextParallel(rank, 4).forward(eng, torch.tensor([[-200, 5, 6]]), pix)
    with pytest.raises(ValueError, match="at least one frame per rank"):
        ContextParallel(1, 8).forward(eng, torch.tensor([[-200, 5]]), pix[:0], total_frames=5)
