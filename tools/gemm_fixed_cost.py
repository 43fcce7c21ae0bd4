import os, sys, torch
os.environ.setdefault('VITG_PROFILING', '1'); os.environ.setdefault('VITG_PROFILING_LIB', '1')   # lets gemm_flags ablation bits through (they produce wrong results)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops
for _a in sys.argv[1:]:
    if '=' in _a:
        ops.set_option(_a.split('=')[0], int(_a.split('=')[1]))
dt = torch.float16
for (M,N,K,mode) in [(524288,3456,64,0),(524288,3456,128,0),(524288,3456,256,0),(524288,3456,1152,0),(524288,1152,64,1),(524288,1152,1152,1),(524288,1152,64,0)]:
    a = (torch.rand((M, K), device="cuda") * 2 - 1).to(dt)
    w = ((torch.rand((N, K), device="cuda") * 2 - 1) * K ** -0.5).to(dt)
    if mode == 1:
        out = torch.zeros((M, N), dtype=torch.float32, device="cuda"); kw = dict(res=out)
    else:
        out = torch.empty((M, N), dtype=dt, device="cuda"); kw = dict(bias=torch.zeros(N, device="cuda"))
    for _ in range(2): ops.gemm(a, w, out, mode=mode, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.gemm(a, w, out, mode=mode, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/5
    tiles = ((M+255)//256)*((N+255)//256)
    print(f"M={M} N={N} K={K} mode={mode}: {ms:.3f} ms, tiles={tiles}, per-tile-per-CU {ms*1e3/(tiles/256):.2f} us, out GB/s {M*N*(2 if mode==0 else 8)/ms/1e6:.0f}")
