import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_w4_ab import run
B, H, HKV, D, S = 1, 4, 2, 128, 4096
def count(qkv, tag):
    outs = []
    for sp in (0, 4, 0, 4):
        o = torch.zeros((B * S, H * D), dtype=torch.float16, device="cuda")
        run(qkv, o, B, H, HKV, S, D, sp)
        outs.append(o.float())
    print(f"{tag:28s}: sp vs w4 differ in {int((outs[0] != outs[1]).sum())}; sp vs sp {int((outs[0] != outs[2]).sum())}; w4 vs w4 {int((outs[1] != outs[3]).sum())} of {outs[0].numel()}", flush=True)
torch.manual_seed(0)
base = torch.randn((B * S, (H + 2 * HKV) * D), device="cuda").half()
count(base, "random")
x = base.clone(); x[:, H * D:(H + HKV) * D] = 0
count(x, "K = 0 (uniform P)")
x = base.clone(); x[:, :H * D] *= 0.01
count(x, "Q * 0.01 (flat scores)")
x = base.clone(); x[:, :H * D] *= 4
count(x, "Q * 4 (peaky, rescales)")
x = base.clone(); x[:, (H + HKV) * D:] = x[:, (H + HKV) * D:].round()
count(x, "V integers")
