#!/usr/bin/env python3
"""Socket power and shader clock while the step (and the MFMA-only probe) runs:  python tools/power_trace.py [seconds per phase]

A sampler thread reads the amdgpu hwmon files (power1_average / power1_input in uW, power1_cap, freq1_input = sclk in Hz) every 20 ms;
if the files are not there it falls back to one `rocm-smi --showpower --showclocks` snapshot per phase.  Phases: idle, the headline
fp16 step in a loop, the MFMA-only probe on zero and on random fp16 operands, a 8192^3 GEMM loop.  Printed per phase: mean / p95 power,
the cap, mean sclk - the evidence for "occupancy x clock is what the chip holds constant" (profiles/r02_gemm_timeline.md)."""
import glob, os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videoitg_amd import ops, synth
from videoitg_amd.config import ITGConfig
from videoitg_amd.engine import SelectorEngine
from videoitg_amd.synth_device import packed_random_weights

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0


def find_hwmon():
    """hwmon directory of the GPU this process computes on (cuda:0): matched by PCI address - /sys/class/drm lists every GPU and
    partition of the box, and card0 is not necessarily ours"""
    want = None
    try:
        pr = torch.cuda.get_device_properties(0)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    except Exception:
        pass
    cands = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        names = os.listdir(d)
        p = next((os.path.join(d, n) for n in ("power1_average", "power1_input") if n in names), None)
        if not p:
            continue
        pci = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))
        cands.append(dict(dir=d, pci=pci, power=p, cap=os.path.join(d, "power1_cap") if "power1_cap" in names else None,
                          sclk=os.path.join(d, "freq1_input") if "freq1_input" in names else None))
    for c in cands:
        if want and c["pci"].lower().startswith(want):
            c["matched"] = True
            return c
    if cands:
        cands[0]["matched"] = False                     # could not match: say so in the output
        return cands[0]
    return None


def read_num(path):
    try:
        with open(path) as f:
            return float(f.read().strip())
    except Exception:
        return None


class Sampler(threading.Thread):
    def __init__(self, hw):
        super().__init__(daemon=True)
        self.hw, self.stop_, self.p, self.f = hw, False, [], []

    def run(self):
        while not self.stop_:
            v = read_num(self.hw["power"])
            if v is not None: self.p.append(v / 1e6)
            if self.hw["sclk"]:
                v = read_num(self.hw["sclk"])
                if v is not None: self.f.append(v / 1e9)
            time.sleep(0.02)


def phase(name, fn, hw):
    torch.cuda.synchronize()
    s = Sampler(hw) if hw else None
    if s: s.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < SECS:
        fn(); torch.cuda.synchronize(); n += 1
    dt = time.perf_counter() - t0
    snap = ""
    if s:
        s.stop_ = True; s.join()
        p = sorted(s.p[len(s.p) // 5:])                    # drop the ramp-up fifth
        f = s.f[len(s.f) // 5:]
        if p:
            snap = f"power mean {sum(p) / len(p):7.1f} W  p95 {p[int(0.95 * (len(p) - 1))]:7.1f} W  ({len(p)} samples)"
        if f:
            snap += f"  sclk mean {sum(f) / len(f):.3f} GHz (min {min(f):.3f} max {max(f):.3f})"
    else:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
            snap = " | ".join(l.strip() for l in out.splitlines() if "Power" in l or "sclk" in l)
        except Exception as e:
            snap = f"no hwmon files and rocm-smi failed: {e}"
    print(f"{name:34s}: {n:4d} calls in {dt:5.1f} s ({dt / max(n, 1) * 1e3:8.2f} ms each)  {snap}", flush=True)


def main():
    hw = find_hwmon()
    print("hwmon:", hw, flush=True)
    if hw and hw["cap"]:
        print(f"power cap: {read_num(hw['cap']) / 1e6:.0f} W", flush=True)
    dev = torch.device("cuda:0")
    cfg = ITGConfig()
    eng = SelectorEngine(packed_random_weights(cfg, torch.float16, dev, seed=0))
    ids, pix = synth.workload_inputs(cfg, 512, 448, 60, 1234)
    pix = pix.to(torch.float16).to(dev)
    eng.forward(ids, None, [pix]); torch.cuda.synchronize()
    phase("idle (sleep)", lambda: time.sleep(0.2), hw)
    phase("fp16 step 512x448 (headline)", lambda: eng.forward(ids, None, [pix]), hw)
    phase("MFMA-only probe, zero operands", lambda: ops.mfma_probe_tflops(torch.float16, False, 40.0), hw)
    phase("MFMA-only probe, random operands", lambda: ops.mfma_probe_tflops(torch.float16, True, 40.0), hw)
    a = torch.randn((8192, 8192), device=dev).half(); w = (torch.randn((8192, 8192), device=dev) / 90).half(); o = torch.empty((8192, 8192), dtype=torch.float16, device=dev)
    def g():
        for _ in range(50): ops.gemm(a, w, o, mode=0)
    phase("8192^3 fp16 GEMM loop", g, hw)


if __name__ == "__main__":
    main()
