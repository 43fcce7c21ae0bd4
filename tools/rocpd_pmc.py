#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite, ROCm 7.2).

    python tools/rocpd_pmc.py <fetch.db> <write.db> > profiles/r01_pmc_traffic.md

FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports HALF the bytes
of a wide coalesced stream read (128-B requests tallied at 64 B) -> the "fetch x2" column; WRITE_SIZE is uncalibrated
and shown raw."""
import collections
import json
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(.*$", "", n).replace("void ", "")
    return n[:70]


def load(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, grid_size, value from counters_collection where counter_name=?", (counter,)).fetchall()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, g, v in rows:
        a = agg[(short(n), g)]
        a[0] += 1
        a[1] += v
    return agg


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    print("# HBM traffic per dispatch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)\n")
    print("| kernel | grid | calls | fetch MiB (raw) | fetch MiB (x2, gfx950) | write MiB (raw) |")
    print("|---|---|---|---|---|---|")
    gem = [0, 0.0, 0.0]
    for k in sorted(f, key=lambda k: -f[k][1]):
        if not ("vitg" in k[0]):
            continue
        nf, sf = f[k]
        nw, sw = w.get(k, (1, 0.0))
        print(f"| `{k[0]}` | {k[1]} | {nf} | {sf / nf / 1024:.1f} | {2 * sf / nf / 1024:.1f} | {sw / max(nw, 1) / 1024:.1f} |")
        if "gemm" in k[0]:
            gem[0] += nf
            gem[1] += 2 * sf
            gem[2] += sw
    if gem[0]:
        per = (gem[1] + gem[2]) * 1024 / gem[0]
        print(f"\nGEMM kernels: {gem[0]} launches, HBM bytes per launch = {per / 1e6:.1f} MB "
              f"(fetch x2 {gem[1] * 1024 / gem[0] / 1e6:.1f} MB + write {gem[2] * 1024 / gem[0] / 1e6:.1f} MB)")
        json.dump({"hbm_bytes_per_launch": per, "launches": gem[0], "source": "rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, separate passes"},
                  open("profiles/gemm_traffic.json", "w"))


if __name__ == "__main__":
    main()
