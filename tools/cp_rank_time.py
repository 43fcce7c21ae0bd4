#!/usr/bin/env python3
"""Single-GPU estimate of one rank's compute time in the N-GPU context-parallel forward (no communication): the rank's
frame shard goes through the tower, its token rows through the LLM with a stand-in K/V gather that returns a full-length
K|V matrix.  python tools/cp_rank_time.py [world=8] [frames=512] [res=448]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops, synth
from videoitg_amd.config import ITGConfig
from videoitg_amd.engine import SelectorEngine
from videoitg_amd.parallel import make_plan
from videoitg_amd.synth_device import packed_random_weights

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
R = int(sys.argv[3]) if len(sys.argv) > 3 else 448
cfg = ITGConfig()
dev = torch.device("cuda:0")
eng = SelectorEngine(packed_random_weights(cfg, torch.float16, dev))
G = R // cfg.vision.patch_size
hw = cfg.pooled_grid(T, G)
plan = make_plan(world, T, hw * hw, 0, 60)
rank = int(os.environ.get("VITG_CP_RANK", world - 1))   # default: the rank that also owns the prompt tokens (the critical one)
fr = plan.frames[rank]
pix = (torch.rand((len(fr), 3, R, R), device=dev) * 2 - 1).half()
kvw = 2 * cfg.num_key_value_heads * cfg.head_dim
kv_all = torch.randn((plan.total, kvw), device=dev).half()

class NoCommGatherer:
    """the split API of parallel.RowGatherer without a collective: this rank's K|V rows are written in place into the full buffer
    (random rows stand for the other ranks'), start() / finish() cost nothing - what is measured is one rank's COMPUTE, including
    the K|V-first / Q split of the QKV projection that the overlap needs"""
    def local_rows(self, c, dtype, device):
        return kv_all[plan.offset[rank]:plan.offset[rank] + plan.n_local[rank]]
    def start(self):
        return None
    def finish(self, work, producer_stream=None):
        if producer_stream is not None:
            torch.cuda.current_stream().wait_stream(producer_stream)
        return kv_all

    def __call__(self, local):                                   # VITG_CP_OVERLAP=0: fused QKV projection, then this (round-2 form)
        kv_all[plan.offset[rank]:plan.offset[rank] + local.shape[0]].copy_(local)
        return kv_all


gather = NoCommGatherer()

def step():
    feats, _, cmap, pc = eng.vit_forward(pix, hw)
    n = plan.n_local[rank]
    x = eng._buf("llm_x", (n, cfg.hidden_size), torch.float32)
    eng.project_into(feats, len(fr), G, hw, x[:len(fr) * hw * hw], cmap, pc)
    h = eng.llm_forward(x, pos0=plan.offset[rank], total=plan.total, kv_gather=gather)
    return h

for _ in range(2): step()
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record()
feats = eng.vit_forward(pix, hw)
e[1].record()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"world={world}: rank-local step {dt * 1e3:.1f} ms (tower part {e[0].elapsed_time(e[1]):.1f} ms); "
      f"n_local={plan.n_local[rank]} of {plan.total} tokens, {len(fr)} frames")

# the same step replayed from a captured graph: what the launch gaps between the ~850 kernels of a step cost
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        step()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    dg = (time.perf_counter() - t0) / 3
    print(f"world={world}: graph replay {dg * 1e3:.1f} ms  (stream launches {dt * 1e3:.1f} ms)")
except Exception as ex:  # noqa: BLE001
    print("graph capture failed:", type(ex).__name__, str(ex)[:300])
