#!/usr/bin/env python3
"""bench.py — frames scored / second of the VideoITG frame-selector forward on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one selector forward over one synthetic video (BASELINE.json configs[1]: 512 frames x 448^2,
VideoITG-8B architecture, random-init weights, 60-token prompt): pixel_values and input_ids already resident in
HBM -> SigLIP tower -> pool + projector -> splice -> 28 bidirectional Qwen2 layers -> score head -> sigmoid ->
stable descending sort -> index list on the host (rows A7-A13 of SURVEY.md §8a; timed region of §8d).

N > 1: one process per GPU; the video's frames are sharded over ranks for the tower/projector and the LLM runs
context-parallel (per-layer RCCL all-gather of K/V, final all-gather of the logits), so total work is fixed
("scaling": "strong").

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel = the MFMA GEMM; HIP
events around every GEMM launch of the timed steps, on the launch stream) and `cpu_baseline` (the oracle timed on
this host's cores on a bounded sample, rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense bf16/fp16, MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--res", type=int, default=448)
    ap.add_argument("--prompt-len", type=int, default=60)
    # bf16 = the dtype of BASELINE.json's bench configuration; fp16 = the reference's inference dtype (builder.py:59) and the dtype
    # the parity tests hold to 1e-3.  Same kernels, same rate class.
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"],
                    help="MFMA input type; bf16 = the configuration BASELINE.json names, fp16 = what the reference itself runs (model.half())")
    ap.add_argument("--config", default="full", choices=["full", "tiny"])
    ap.add_argument("--graph", action="store_true", help="replay the single-GPU step from a captured HIP graph (same kernels; no gain measured)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the from-uint8-frames (GPU preprocessing) measurement")
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------
def algorithmic_flops(cfg, T, R, n_text):
    """per-video algorithmic FLOPs (multiply-add = 2), needed ViT layers only."""
    v = cfg.vision
    G = R // v.patch_size
    P = G * G
    Dv, Fv, Lv = v.hidden_size, v.intermediate_size, cfg.num_vit_layers_needed()
    hw = cfg.pooled_grid(T, G)
    N = T * hw * hw + n_text
    D, Fl, Ll = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    kvw = cfg.num_key_value_heads * cfg.head_dim
    patch = 2.0 * T * P * (3 * v.patch_size ** 2) * Dv
    vit_gemm = T * Lv * 2.0 * P * (4 * Dv * Dv + 2 * Dv * Fv)
    vit_attn = T * Lv * 4.0 * P * P * Dv
    proj = 2.0 * T * hw * hw * (Dv * D + D * D)
    llm_gemm = Ll * 2.0 * N * (D * (D + 2 * kvw) + D * D + 3 * D * Fl)
    llm_attn = Ll * 4.0 * N * N * D
    head = 2.0 * T * D
    return dict(patch=patch, vit_gemm=vit_gemm, vit_attn=vit_attn, proj=proj, llm_gemm=llm_gemm, llm_attn=llm_attn,
                head=head, total=patch + vit_gemm + vit_attn + proj + llm_gemm + llm_attn + head,
                gemm=patch + vit_gemm + proj + llm_gemm, N=N, P=P, hw=hw)


class GemmTimer:
    """HIP events (torch.cuda.Event on the current stream = the stream every kernel is launched on) around each
    GEMM launch; resolved after the timed region."""

    def __init__(self):
        self.pairs = []
        self.enabled = False

    def wrap(self, fn):
        def timed(*a, **k):
            if not self.enabled:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            self.pairs.append((e0, e1))
            return r
        return timed

    def summary(self):
        ms = [a.elapsed_time(b) for a, b in self.pairs]
        return len(ms), sum(ms)


def cpu_baseline(cfg, T, R, n_text, threads):
    """Oracle (kind "port") on this host's cores, bounded sample of the same workload (~10-30 s):
    ViT: embed + 2 encoder layers on 16 frames at R;  LLM: 1 decoder layer on 4096 of the N tokens.  Extrapolated to the
    whole video: ViT x (T/16) frame groups x (26/2) layers (frames are independent, layers identical); LLM x 28 layers x
    the FLOP ratio of a layer at N vs at 4096 tokens (GEMMs scale with N, attention with N^2).  Projector/head < 0.1 %."""
    from oracle import videoitg_oracle as O          # checker/baseline only
    from videoitg_amd import synth
    torch.set_num_threads(threads)
    fl = algorithmic_flops(cfg, T, R, n_text)
    v = cfg.vision
    specs = {k: (s, m, sd) for k, s, m, sd in synth.vision_param_specs(cfg)}
    vsd = {}
    for k, (s, m, sd) in specs.items():
        if k.startswith("vision_model.embeddings") or ".layers.0." in k or ".layers.1." in k:
            vsd[k] = synth.synth_tensor(0, k, s, m, sd)
    nf = 16
    pix = synth.synth_pixel_values(nf, R, seed=1234)
    with torch.inference_mode():
        O.siglip_tower(pix[:2], vsd, patch=v.patch_size, num_heads=v.num_attention_heads, num_layers=2, select_layer=2)  # warm
        t0 = time.perf_counter()
        O.siglip_tower(pix, vsd, patch=v.patch_size, num_heads=v.num_attention_heads, num_layers=2, select_layer=2,
                       frame_chunk=nf)
        t_vit = time.perf_counter() - t0
    sd = {}
    for k, s, m, sdv in synth.llm_param_specs(cfg):
        if ".layers.0." in k:
            sd[k] = synth.synth_tensor(0, k, s, m, sdv)
    N = fl["N"]
    Ns = min(N, 4096)
    D, Fl, kvw = cfg.hidden_size, cfg.intermediate_size, cfg.num_key_value_heads * cfg.head_dim
    layer_flops = lambda n: 2.0 * n * (D * (D + 2 * kvw) + D * D + 3 * D * Fl) + 4.0 * n * n * D
    x = torch.randn(1, Ns, cfg.hidden_size)
    cos, sin = O.rope_tables(Ns, cfg.head_dim, cfg.rope_theta)
    with torch.inference_mode():
        t0 = time.perf_counter()
        O.qwen2_layer(x, sd, 0, num_heads=cfg.num_attention_heads, num_kv_heads=cfg.num_key_value_heads,
                      eps=cfg.rms_norm_eps, cos=cos, sin=sin, key_mask=None)
        t_llm = time.perf_counter() - t0
    t_llm_full = t_llm * layer_flops(N) / layer_flops(Ns)
    est = t_vit * (T / nf) * (cfg.num_vit_layers_needed() / 2.0) + t_llm_full * cfg.num_hidden_layers
    return dict(value=T / est, unit="frames/s", cores=threads, kind="port",
                sample=f"oracle (torch CPU fp32, {threads} threads): SigLIP embed+2 layers on {nf} frames@{R} in {t_vit:.2f}s; 1 Qwen2 "
                       f"layer on {Ns} of {N} tokens in {t_llm:.2f}s (x{layer_flops(N) / layer_flops(Ns):.2f} by FLOPs to N); extrapolated "
                       f"x{T // nf} frame groups x{cfg.num_vit_layers_needed()}/2 layers + x{cfg.num_hidden_layers} layers = {est:.0f}s per video",
                sample_seconds=round(t_vit + t_llm, 2))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU path for the product)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from videoitg_amd import ops, synth
    from videoitg_amd.config import ITGConfig, tiny_config
    from videoitg_amd.engine import SelectorEngine
    from videoitg_amd.parallel import ContextParallel
    from videoitg_amd.synth_device import packed_random_weights

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    cfg = ITGConfig() if args.config == "full" else tiny_config(vision_token_num=16384)
    T, R = args.frames, args.res
    if args.config == "tiny":
        R = cfg.vision.image_size
    weights = packed_random_weights(cfg, dtype, dev, seed=args.seed)
    eng = SelectorEngine(weights)
    cp = ContextParallel(rank, world) if world > 1 else None

    # synthetic inputs, identical on every rank; resident in HBM before the timed region
    g = torch.Generator(device="cpu").manual_seed(1234)
    ids = torch.cat([torch.tensor([-200]), synth.synth_prompt_ids(args.prompt_len, cfg.vocab_size)]).unsqueeze(0)
    pix = (torch.rand((T, 3, R, R), generator=g) * 2 - 1).to(dtype).to(dev)

    timer = GemmTimer()
    ops.gemm = timer.wrap(ops.gemm)

    graphed = None
    if args.graph and cp is None:
        from videoitg_amd.engine import GraphedForward
        graphed = GraphedForward(eng, ids, pix)

    def step():
        if graphed is not None:
            logits = graphed(ids, pix)
        elif cp is None:
            logits = eng.forward(ids, None, [pix])[0].view(-1)
        else:
            logits = cp.forward(eng, ids, pix)
        scores, order = ops.rank(logits)
        return order.cpu().tolist(), scores          # D2H of the index list = end of the timed region

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        order, scores = step()
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    n_launch, gemm_ms = timer.summary()
    fl = algorithmic_flops(cfg, T, R, ids.shape[1] - 1)
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        gemm_flops_step = fl["gemm"] / world                     # this rank's share (frames / tokens are sharded)
        per_launch_flops = gemm_flops_step * args.steps / max(n_launch, 1)
        avg_ms = gemm_ms / max(n_launch, 1)
        achieved = per_launch_flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "gemm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "frames scored/sec", "value": round(T / (dt / args.steps), 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic (random-init weights of the VideoITG-8B architecture, uniform pixels, "
                                         "60-token prompt)",
            "config": {"workload": f"{T} frames x {R}^2, SigLIP-so400m/14 ({cfg.num_vit_layers_needed()} layers) + seq_mlp "
                                   f"(pool {fl['hw']}x{fl['hw']}) + Qwen2-7B bidirectional ({cfg.num_hidden_layers} layers), "
                                   f"N={fl['N']} tokens; pixels+ids resident in HBM -> sorted index list on host",
                       "frames": T, "resolution": R, "llm_tokens": fl["N"],
                       "parallelism": "single GPU" if world == 1 else f"frame-sharded tower + context-parallel LLM x{world}",
                       "algorithmic_tflop_per_video": round(fl["total"] / 1e12, 1)},
            "end_to_end_tflops": round(fl["total"] / (dt / args.steps) / 1e12, 1),
            "roofline": {"kernel": "gemm256_kernel (all GEMM launches of the step)", "bound": "mfma", "achieved": round(achieved, 1),
                         "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                         "traffic": traffic, "launches": n_launch, "avg_launch_ms": round(avg_ms, 4),
                         "gemm_share_of_step": round(gemm_ms / (dt * 1e3), 3)},
            "top32": sorted(order[:32]),
        }
        if world == 1 and not args.no_secondary:
            # What the matrix pipe sustains on this box with NO memory traffic, on zero and on random operands: the shader clock under
            # MFMA load is power managed and data dependent (probe.hip; profiles/r01_gemm_shapes.md section 13), so `peak` (nominal)
            # is not reachable on real data by any kernel.  Reported for reading `frac`; `peak`/`frac` stay on the nominal figure.
            out["roofline"]["mfma_only_probe"] = {
                "zero_operands": round(ops.mfma_probe_tflops(dtype, False), 1),
                "random_operands": round(ops.mfma_probe_tflops(dtype, True), 1), "unit": "TFLOP/s",
                "what": "MFMA-only kernel, 8 waves/CU, no memory traffic; power-managed clock"}
            # SURVEY.md 8d "secondary": the same job starting from decoded uint8 NHWC frames resident in HBM (GPU
            # preprocessing front-end, row N2, in front of the selector forward).  Reported beside `value`, never as `value`.
            frames = torch.randint(0, 256, (T, 720, 1280, 3), dtype=torch.uint8, device=dev)

            def step_u8():
                px = ops.preprocess_u8(frames, R, dtype)
                _, order_ = ops.rank(eng.forward(ids, None, [px])[0].view(-1))
                return order_.cpu().tolist()
            step_u8()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step_u8()
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.preprocess_u8(frames, R, dtype)
            e1.record()
            torch.cuda.synchronize()
            out["secondary"] = {"from_uint8_frames": {"input": f"{T} x 720x1280x3 uint8 NHWC in HBM", "value": round(T / d1, 2),
                                                      "unit": "frames/s", "ms_per_step": round(d1 * 1e3, 2),
                                                      "preprocess_ms": round(e0.elapsed_time(e1), 3)}}
            del frames
            # B0 = 512 frames at the reference-native 384^2 (27x27 grid, no position-embedding interpolation): SURVEY.md 8d
            g0 = torch.Generator().manual_seed(1234)
            pix0 = (torch.rand((T, 3, 384, 384), generator=g0) * 2 - 1).to(dtype).to(dev)

            def step_b0():
                _, order_ = ops.rank(eng.forward(ids, None, [pix0])[0].view(-1))
                return order_.cpu().tolist()
            step_b0()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            step_b0()
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t2
            fl0 = algorithmic_flops(cfg, T, 384, ids.shape[1] - 1)
            out["secondary"]["b0_384"] = {"input": f"{T} x 384^2 (reference-native resolution), N={fl0['N']}", "value": round(T / d2, 2),
                                          "unit": "frames/s", "ms_per_step": round(d2 * 1e3, 2),
                                          "algorithmic_tflop_per_video": round(fl0["total"] / 1e12, 1)}
            del pix0
            # the other 16-bit type on the same workload: bf16 is what BASELINE.json's configuration names, fp16 what the reference
            # itself runs (model.half()); both go through the same kernels
            other = torch.float16 if dtype == torch.bfloat16 else torch.bfloat16
            eng.release_workspace()
            eng_o = SelectorEngine(packed_random_weights(cfg, other, dev, seed=args.seed))
            pix_o = pix.to(other)

            def step_o():
                _, order_ = ops.rank(eng_o.forward(ids, None, [pix_o])[0].view(-1))
                return order_.cpu().tolist()
            step_o()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            step_o()
            step_o()
            torch.cuda.synchronize()
            d3 = (time.perf_counter() - t3) / 2
            out["secondary"]["fp16" if other == torch.float16 else "bf16"] = {"value": round(T / d3, 2), "unit": "frames/s",
                                                                               "ms_per_step": round(d3 * 1e3, 2)}
            eng_o.release_workspace()
            del eng_o, pix_o
        if world == 1 and not args.no_cpu_baseline:
            eng.release_workspace()
            try:
                import psutil
                cores = psutil.cpu_count(logical=False) or os.cpu_count() or 1
            except Exception:
                cores = os.cpu_count() or 1
            out["cpu_baseline"] = cpu_baseline(cfg, T, R, ids.shape[1] - 1, cores)
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
