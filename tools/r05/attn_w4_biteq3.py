import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_w4_ab import run
from videoitg_amd import ops
B, H, HKV, D = 1, 4, 2, 128
for S in (2048, 2049, 2112, 2176, 2304, 2305, 2111):
    torch.manual_seed(1)
    qkv = torch.randn((B * S, (H + 2 * HKV) * D), device="cuda").half()
    outs = {}
    for tag, sp, nw in (("sp8", 0, 8), ("sp4", 0, 4), ("w4", 4, 0), ("pertile", 2, 0)):
        ops.set_option("attn_waves", nw)
        o = torch.zeros((B * S, H * D), dtype=torch.float16, device="cuda")
        run(qkv, o, B, H, HKV, S, D, sp)
        outs[tag] = o.float()
    ops.set_option("attn_waves", 0)
    print(f"S={S}: sp8 vs sp4 {int((outs['sp8'] != outs['sp4']).sum())}   sp8 vs w4 {int((outs['sp8'] != outs['w4']).sum())}   (rows of the diffs: {sorted(set((outs['sp8'] != outs['w4']).nonzero()[:, 0].tolist()))[:12]})", flush=True)
