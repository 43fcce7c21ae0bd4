import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd import ops
from attn_w4_ab import run, ref  # noqa
B, H, HKV, S, D = 1, 2, 1, int(sys.argv[1]) if len(sys.argv) > 1 else 192, 128
torch.manual_seed(S)
dt = torch.float16
qkv = torch.randn((B * S, (H + 2 * HKV) * D), device="cuda").to(dt)
r = ref(qkv, B, H, HKV, S, D)
o = torch.full((B * S, H * D), float("nan"), dtype=dt, device="cuda")
run(qkv, o, B, H, HKV, S, D, 4)
e = (o.double() - r).abs().view(S, H, D)
print("err by head:", e.amax((0, 2)).tolist())
eq = e.amax((1, 2))
print("bad query rows:", (eq > 1e-2).nonzero().flatten().tolist()[:64], "of", S)
ed = e.amax((0, 1))
print("bad d cols:", (ed > 1e-2).nonzero().flatten().tolist())
q = int(eq.argmax())
print("row", q, "got", o.view(S, H, D)[q, 0, :8].tolist(), "ref", r.view(S, H, D)[q, 0, :8].tolist())
# ratio test: is the output a rescaled version (normaliser wrong) or structurally wrong?
num = (o.double().view(S, H, D) * r.view(S, H, D)).sum(-1); den = (r.view(S, H, D) ** 2).sum(-1)
print("per-row scale (o.r / r.r), head 0, first 70 rows:", [round(float(x), 3) for x in (num / den)[:70, 0]])
