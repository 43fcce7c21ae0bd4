#!/usr/bin/env python3
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd.torch_ops import K
dev = "cuda:0"
torch.manual_seed(0)
def joined(t, w, d): return t[:, :d].double() + t[:, w:w + d].double()
def split_pair(x, width):
    out = torch.zeros((x.shape[0], 2 * width), dtype=torch.float16, device=dev)
    hi = x.half(); out[:, :x.shape[1]] = hi; out[:, width:width + x.shape[1]] = (x - hi.float()).half(); return out
for hd, H, KV, S in ((128, 1, 1, 64), (128, 1, 1, 128), (128, 1, 1, 192), (128, 1, 1, 523), (72, 1, 1, 128), (72, 1, 1, 1024)):
    wq, wk = H * hd, KV * hd
    q, k, v = torch.randn(S, wq, device=dev), torch.randn(S, wk, device=dev), torch.randn(S, wk, device=dev)
    q2, k2, v2 = split_pair(q, wq), split_pair(k, wk), split_pair(v, wk)
    op = K.pad64(wq)
    o2 = torch.zeros(S, 2 * op, dtype=torch.float16, device=dev)
    K.attention_split(q2, k2, v2, o2, lo=(wq, wk, wk, op), batch=1, heads=H, kv_heads=KV, seq=S, head_dim=hd, q_strides=(0, 2 * wq, hd),
                      k_strides=(0, 2 * wk, hd), v_strides=(0, 2 * wk, hd), o_strides=(0, 2 * op, hd), scale=hd ** -0.5)
    qe, ke, ve = joined(q2, wq, wq), joined(k2, wk, wk), joined(v2, wk, wk)
    sc = qe @ ke.t() * hd ** -0.5
    ref = torch.softmax(sc, -1) @ ve
    got = joined(o2, op, wq)
    e = (got - ref).abs()
    bad = (e > 2e-6).nonzero()
    rows = sorted(set(bad[:, 0].tolist()))
    print(f"hd {hd} S {S}: max {e.max().item():.2e} rms {e.pow(2).mean().sqrt().item():.2e}; elements > 2e-6: {bad.shape[0]} in rows {rows[:12]}{'...' if len(rows) > 12 else ''}")
    for r in rows[:3]:
        er = e[r]
        cols = (er > 2e-6).nonzero().flatten().tolist()
        # where is the row's max score, and how large is the gap to the runner-up?
        top = sc[r].topk(2)
        c0 = cols[0]
        hi_, lo_ = o2[r, c0].item(), o2[r, op + c0].item()
        rf = ref[r, c0].item()
        ehi = torch.tensor(rf).half().item()
        print(f"      got hi {hi_!r} lo {lo_!r}; ref {rf!r}; fp16(ref) {ehi!r}, ref - fp16(ref) {rf - ehi!r}; ulp(hi) {torch.tensor(abs(hi_)).half().nextafter(torch.tensor(65504.).half()).item() - abs(hi_)!r}")
        print(f"   row {r}: {len(cols)} bad cols {cols[:8]}; row max err {er.max().item():.2e}; argmax key {top.indices[0].item()} (tile {top.indices[0].item() // 64}), top scores {top.values.tolist()}, "
              f"hi-only err would be {(o2[r, :wq].double() - ref[r]).abs().max().item():.2e}")
