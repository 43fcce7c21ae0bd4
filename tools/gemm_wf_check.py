import torch, sys
import os; os.environ.setdefault('VITG_PROFILING_LIB', '1')   # fragment-packed weights live in the profiling library
sys.path.insert(0, '/root/repo')
from videoitg_amd import ops
torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float16):
    for (M, N, K, mode) in [(512, 512, 64, 0), (512, 512, 128, 0), (512, 512, 192, 0), (777, 520, 256, 0), (1024, 768, 1152, 0), (1000, 1152, 4352, 1), (1300, 1024, 512, 2), (2048, 3456, 1152, 0), (256, 256, 64, 1)]:
        a = (torch.rand((M, K), device='cuda') * 2 - 1).to(dt)
        w = ((torch.rand((N, K), device='cuda') * 2 - 1) * K ** -0.5).to(dt)
        bias = torch.randn(N, device='cuda')
        wf = ops.pack_weight_fragments(w)
        if mode == 1:
            res = torch.randn((M, N), device='cuda')
            o1 = ops.gemm(a, w, torch.empty((M, N), device='cuda'), mode=1, bias=bias, res=res)
            o2 = ops.gemm(a, w, torch.empty((M, N), device='cuda'), mode=1, bias=bias, res=res, wf=wf)
            ref = a.float() @ w.float().T + bias + res
        elif mode == 2:
            o1 = ops.gemm(a, w, torch.empty((M, N // 2), dtype=dt, device='cuda'), mode=2)
            o2 = ops.gemm(a, w, torch.empty((M, N // 2), dtype=dt, device='cuda'), mode=2, wf=wf)
            ref = None
        else:
            o1 = ops.gemm(a, w, torch.empty((M, N), dtype=dt, device='cuda'), mode=0, bias=bias)
            o2 = ops.gemm(a, w, torch.empty((M, N), dtype=dt, device='cuda'), mode=0, bias=bias, wf=wf)
            ref = a.float() @ w.float().T + bias
        d = (o1.float() - o2.float()).abs().max().item()
        e = (o2.float() - ref).abs().max().item() if ref is not None else float('nan')
        print(dt, M, N, K, mode, 'wf-vs-lds', d, 'wf-vs-fp32', e, 'equal' if torch.equal(o1, o2) else 'DIFF', flush=True)
