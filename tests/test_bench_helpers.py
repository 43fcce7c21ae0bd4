"""CPU: the pure-Python parts of bench.py - FLOP accounting (SURVEY.md 8d figures), executed-FLOP correction, parity block."""
import importlib.util
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_algorithmic_flops_match_the_survey():
    from videoitg_amd.config import ITGConfig
    cfg = ITGConfig()
    fl = bench.algorithmic_flops(cfg, 512, 448, 60)
    assert fl["N"] == 12860 and fl["hw"] == 5 and fl["P"] == 1024
    assert abs(fl["total"] / 1e12 - 714.8) < 0.5                      # R = 448 -> 714.8 TFLOP per video
    fl0 = bench.algorithmic_flops(cfg, 512, 384, 60)
    assert abs(fl0["total"] / 1e12 - 563.3) < 0.5                     # B0
    fl1 = bench.algorithmic_flops(cfg, 64, 384, 60)
    assert fl1["hw"] == 16 and abs(fl1["total"] / 1e12 - 364.8) < 0.5  # C1
    assert bench.algorithmic_flops(cfg, 16, 448, 60)["hw"] == 32      # T <= 22 at a 32 x 32 grid: no pooling (actual-grid clamp)


def test_executed_gemm_flops_subtract_only_the_dead_rows():
    from videoitg_amd.config import ITGConfig
    cfg = ITGConfig()
    fl = bench.algorithmic_flops(cfg, 512, 448, 60)
    ex = bench.executed_gemm_flops(cfg, 512, 448, 60, True)
    v = cfg.vision
    skipped = 512 * 2.0 * (1024 - 100) * (2 * v.hidden_size ** 2 + 2 * v.hidden_size * v.intermediate_size)   # 100 of 1024 tokens feed the 5x5 pool
    assert abs((fl["gemm"] - ex) - skipped) < 1e6 and 0.015 < skipped / fl["gemm"] < 0.03
    assert bench.executed_gemm_flops(cfg, 512, 448, 60, False) == fl["gemm"]
    assert bench.executed_gemm_flops(cfg, 16, 448, 60, True) == bench.algorithmic_flops(cfg, 16, 448, 60)["gemm"]   # nothing pooled, nothing skipped


def test_parity_block_against_the_committed_fixture():
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "full_c2_512x448.json")))
    lg = torch.tensor(rec["logits"])
    blk = bench.parity_block(512, 448, 60, 0, lg, rec["order"])
    assert blk["fixture"].endswith("full_c2_512x448.json") and blk["pass"] and blk["max_abs_dscore"] == 0.0 and blk["top32_set_match"]
    off = lg + 0.02                                                    # a 2e-2 logit shift moves scores by up to 5e-3: fails the 1e-3 gate
    assert not bench.parity_block(512, 448, 60, 0, off, rec["order"])["pass"]
    assert bench.parity_block(512, 448, 60, 0, off, rec["order"], tol=0.15)["pass"]
    assert bench.parity_block(100, 448, 60, 0, lg[:100], list(range(100))) is None      # no fixture for this workload
    # a looser precision tier (fp8) never passes on its score bound alone: losing the oracle's top frame / Top-32 set is reported and fails it
    wrong = rec["order"][40:] + rec["order"][:40]
    blk = bench.parity_block(512, 448, 60, 0, off, wrong, tol=0.15)
    assert blk["within_score_tolerance"] and not blk["selects_same_frames"] and not blk["top1_kept"] and not blk["pass"]


def test_plain_gpus_n_command_becomes_its_own_launcher(monkeypatch):
    """`python bench.py --gpus 8` with no launcher around it (WORLD_SIZE unset) must start its ranks itself: it re-executes its own command
    line under torch's elastic launcher, one process per GPU, rendezvous on 127.0.0.1 (the reference's counterpart is
    `accelerate launch --num_processes=8`, scripts/eval_lmms_eval/videomme_grounding.sh:3-5)."""
    seen = {}
    monkeypatch.setattr(bench.os, "execv", lambda exe, cmd: seen.update(exe=exe, cmd=cmd))
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VITG_DIST_BACKEND", "VITG_BENCH_SINGLE_DEVICE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)   # an 8-GPU node as the launcher's pre-flight sees it
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    bench.main()                                                       # returns right after the (mocked) exec: no GPU is touched
    cmd = seen["cmd"]
    assert seen["exe"] == bench.sys.executable and cmd[:3] == [bench.sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]


def test_gpus_n_fails_fast_and_says_why(monkeypatch):
    """more ranks than visible GPUs, or a backend that is not RCCL, ends `bench.py --gpus N` with a reason BEFORE any rank starts
    (nothing is exec'ed, nothing hangs); the one-GPU dry run (VITG_BENCH_SINGLE_DEVICE=1 + gloo) stays possible."""
    import pytest
    assert bench.preflight_ranks(8, 8, "nccl", False) is None and bench.preflight_ranks(1, 1, "nccl", False) is None
    assert "4 GPU(s) visible" in bench.preflight_ranks(8, 4, "nccl", False)
    assert "RCCL" in bench.preflight_ranks(2, 2, "gloo", False)
    assert bench.preflight_ranks(2, 1, "gloo", True) is None
    assert bench.preflight_ranks(0, 8, "nccl", False)
    execd = []
    monkeypatch.setattr(bench.os, "execv", lambda exe, cmd: execd.append(cmd))
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "8"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VITG_DIST_BACKEND", "VITG_BENCH_SINGLE_DEVICE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 2)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "8 ranks asked for, 2 GPU(s) visible" in str(e.value) and not execd
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setenv("VITG_DIST_BACKEND", "gloo")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "RCCL" in str(e.value) and not execd


def test_watchdog_names_the_phase_that_hung():
    """a rank that makes no progress ends ITSELF with exit code 124 and the name of the phase, instead of hanging the launcher"""
    import time
    said, codes = [], []
    dog = bench.Watchdog(0.05, rank=3, _exit=codes.append, _err=said.append)
    dog.phase("rendezvous")
    dog.phase("warm-up")                       # re-arming cancels the first timer
    time.sleep(0.3)
    assert codes == [124] and dog.fired == "warm-up" and "rank 3" in said[0] and "'warm-up'" in said[0]
    quiet = bench.Watchdog(0.05, _exit=codes.append, _err=said.append)
    quiet.phase("timed steps")
    quiet.stop()
    time.sleep(0.15)
    assert codes == [124]                      # stopped in time: nothing fired
    off = bench.Watchdog(0.0, _exit=codes.append, _err=said.append)      # single-GPU runs: no watchdog
    off.phase("x")
    time.sleep(0.05)
    assert codes == [124]


def test_first_step_falls_back_to_the_blocking_exchange_on_every_rank():
    """the asynchronous K|V exchange has met RCCL on one rank only: if the first context-parallel step raises on ANY rank, all ranks switch to the
    blocking exchange (eng.cp_overlap = False = VITG_CP_OVERLAP=0), the step is repeated, and the reason goes into the JSON line"""
    import pytest

    class Eng:
        cp_overlap = True
    calls = []

    def step_ok():
        calls.append("ok")
    eng = Eng()
    assert bench.first_step_with_overlap_fallback(step_ok, eng, lambda mine: mine) is None and eng.cp_overlap and calls == ["ok"]
    # this rank is fine, another one failed: fall back anyway (the agreement is an all-reduce MAX)
    eng = Eng()
    why = bench.first_step_with_overlap_fallback(step_ok, eng, lambda mine: True)
    assert "another rank" in why and eng.cp_overlap is False and calls == ["ok", "ok", "ok"]

    def step_async_breaks():
        if eng.cp_overlap:
            raise RuntimeError("NCCL error: in-place all_gather on a side stream")
        calls.append("blocking")
    eng = Eng()
    why = bench.first_step_with_overlap_fallback(step_async_breaks, eng, lambda mine: mine)
    assert why.startswith("RuntimeError: NCCL error") and eng.cp_overlap is False and calls[-1] == "blocking"
    eng = Eng()
    eng.cp_overlap = False                     # already blocking and still failing: loud

    def always():
        raise ValueError("boom")
    with pytest.raises(RuntimeError, match="blocking exchange too"):
        bench.first_step_with_overlap_fallback(always, eng, lambda mine: mine)
    # a failure AFTER this rank started an exchange cannot be agreed upon (the peers sit in a collective this rank left): no all-reduce, no
    # retry - the rank ends the job with a message naming the way out
    eng = Eng()
    said = []
    agreed = []
    out = bench.first_step_with_overlap_fallback(step_async_breaks, eng, lambda mine: agreed.append(mine) or mine, collectives_started=lambda: 3,
                                                 _abort=lambda msg: said.append(msg) or "aborted")
    assert out == "aborted" and agreed == [] and eng.cp_overlap is True
    assert "AFTER 3 K|V exchange" in said[0] and "VITG_CP_OVERLAP=0" in said[0]


def test_algorithmic_gemm_bytes_of_the_headline_workload():
    """what the step's 220 GEMM launches have to move: ~2.8 GB per launch on the 16-bit tower stream (623 GB / 220), more on the fp32 one"""
    from videoitg_amd.config import ITGConfig
    cfg = ITGConfig()
    by, n = bench.algorithmic_gemm_bytes(cfg, 512, 448, 60, 2)
    assert n == 1 + 4 * 26 + 2 + 4 * 28 == 219
    assert 2.6e9 < by / n < 3.1e9
    by4, _ = bench.algorithmic_gemm_bytes(cfg, 512, 448, 60, 4)
    assert by4 - by == (26 * 4 + 1) * 512 * 1024 * 1152 * 2                       # out-proj + fc2 read and write the stream, the patch embedding writes it


def test_hwmon_sampler_is_silent_without_a_gpu():
    """`roofline.power_clock` comes from the amdgpu hwmon files of the GPU the bench runs on; where they do not exist (this CPU container, another
    driver) the sampler starts nothing and reports None - the JSON line then simply has no such block."""
    s = bench.HwmonSampler()
    if s.power is None:                       # no matching hwmon directory: the normal case off the GPU box
        s.start()
        assert s.stop() is None
        assert s.reason and "hwmon" in s.reason        # ... and says why, so the JSON line can carry `power_clock_unavailable`
    else:                                     # a GPU box: a short sample carries the three readings
        import time
        s.start(); time.sleep(0.2)
        r = s.stop()
        assert r is None or (r["power_w"] > 0 and r["samples"] >= 1)


def test_printed_line_stays_under_four_kilobytes(tmp_path, monkeypatch, capsys):
    """r05's 21 KB line came back from the driver as `parsed: null`.  The printed line is the contract keys + roofline + cpu_baseline +
    parity, every string bounded, < 4096 bytes whatever the run measured; everything else goes to bench_full.json, which the line names."""
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_v6b_fp16_driver_style.json")))      # a real 21 KB record of the default run
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full, "bench_full.json")
    blob = json.dumps(line)
    assert len(blob) < bench.LINE_LIMIT == 4096
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["config"]["workload"].startswith("512 frames x 448^2") and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > r["traffic_algorithmic"] > 0
    assert r["launches"] and r["avg_launch_ms"] and "gemm16_kernel" in r["kernel"] and r["power_clock"]["sclk_ghz"]
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 32 and c["unit"] == "frames/s" and 0 < len(c["sample"]) <= 200
    assert line["parity"]["pass"] is True and line["parity"]["top32_set_match"] is True and line["parity"]["max_abs_dscore"] < 1e-3
    assert line["speedup_vs_cpu_baseline"] > 50 and line["full_record"] == "bench_full.json"
    assert set(line["roofline_other"]) >= {"attention_hd72", "attention_hd128", "layernorm"}
    # strings a future run could blow up are clipped, and an oversized extra block is dropped rather than breaking the contract
    fat = dict(full, data="x" * 5000, config=dict(full["config"], workload="w" * 5000))
    fat["cpu_baseline"] = dict(full["cpu_baseline"], sample="s" * 9000, sample_short=None)
    fat["secondary"] = {f"tier_{i:03d}_with_a_long_name": {"value": 1.0 * i} for i in range(200)}
    thin = bench.compact_line(fat, "bench_full.json")
    assert len(json.dumps(thin)) < 4096 and "secondary_frames_per_s" not in thin and "roofline" in thin and "cpu_baseline" in thin
    # emit(): full record on disk, ONE line on stdout
    monkeypatch.setenv("VITG_BENCH_FULL", str(tmp_path / "full.json"))
    bench.emit(full)
    out = capsys.readouterr().out
    assert out.count("\n") == 1 and len(out) < 4097
    assert json.loads(out)["value"] == full["value"] and json.load(open(tmp_path / "full.json"))["secondary"].keys() == full["secondary"].keys()
