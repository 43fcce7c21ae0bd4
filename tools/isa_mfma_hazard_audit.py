#!/usr/bin/env python3
"""Audit a hipcc -S listing for the hazards hipcc does not pad around inline-assembly MFMAs (cdna_hip_programming.md 5.7 item 2):
  (1) a VALU / v_accvgpr_* write of a register an asm MFMA reads (A, B or C) within the 2 instructions in front of it
      (unless the MFMA statement itself opens with s_nop 1);
  (2) a non-MFMA read or write of an asm MFMA's destination within 12 issue slots behind it (8-pass shapes), s_nop N counting N + 1.
usage: isa_mfma_hazard_audit.py listing.s [kernel-name-substring]      exit code 1 if anything is found"""
import re, sys

def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m: return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"-?\|?([va])(\d+)\|?", tok)
    if m: return {(m.group(1), int(m.group(2)))}
    return set()

def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.startswith(".") or line.endswith(":"): return None
    parts = line.split(None, 1)
    op = parts[0]
    ops = [t for t in (parts[1].split(",") if len(parts) > 1 else [])]
    return op, [regs(t.split()[0]) if t.strip() else set() for t in ops]

def main():
    path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    kernel, insts, bad = None, [], 0
    in_asm = False
    for raw in open(path):
        l = raw.rstrip("\n")
        m = re.match(r"^(_Z\w+):", l)
        if m: kernel, insts = m.group(1), []
        if "#ASMSTART" in l: in_asm = True; continue
        if "#ASMEND" in l: in_asm = False; continue
        p = parse(l)
        if not p or kernel is None or want not in kernel: continue
        op, ops = p
        insts.append((op, ops, in_asm, l.strip()))
        if op.startswith("v_mfma") and in_asm:
            reads = set().union(*ops[1:]) if len(ops) > 1 else set()
            prev = [i for i in insts[-3:-1]]
            padded = len(insts) >= 2 and insts[-2][0] == "s_nop" and insts[-2][2]      # s_nop 1 inside the same statement
            if not padded:
                for pop, pops, pasm, ptxt in prev:
                    if pop.startswith("v_") and not pop.startswith("v_mfma") and pops and (pops[0] & reads):
                        print(f"[{kernel}] write->MFMA read: {ptxt}   ->   {l.strip()}"); bad += 1
    # pass 2: MFMA destination touched too early (per kernel, linear scan)
    kernel, win = None, []
    for raw in open(path):
        l = raw.rstrip("\n")
        m = re.match(r"^(_Z\w+):", l)
        if m: kernel, win = m.group(1), []
        if "#ASMSTART" in l: in_asm = True; continue
        if "#ASMEND" in l: in_asm = False; continue
        p = parse(l)
        if not p or kernel is None or want not in kernel: continue
        op, ops = p
        slots = 1
        if op == "s_nop":
            try: slots = int(l.split()[1]) + 1
            except Exception: slots = 1
        if op in ("s_barrier", "s_waitcnt") : slots = 1
        touched = set().union(*ops) if ops else set()
        if not op.startswith("v_mfma"):
            for dst, age, txt in win:
                if age < 12 and (touched & dst) and (op.startswith("v_") or op.startswith("ds_") or op.startswith("global_") or op.startswith("scratch_") or op.startswith("buffer_")):
                    print(f"[{kernel}] MFMA D touched after {age} slots: {txt}   ->   {l.strip()}"); bad += 1
        win = [(d, a + slots, t) for d, a, t in win if a + slots < 16]
        if op.startswith("v_mfma") and in_asm and ops: win.append((ops[0], 0, l.strip()))
    print(f"{path}: {bad} hazard(s)")
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
