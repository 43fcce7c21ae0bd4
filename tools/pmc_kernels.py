#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes (rocpd sqlite) per kernel: python tools/pmc_kernels.py db1 [db2 ...]"""
import collections, re, sqlite3, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(list)
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    for n, g, cn, v, d in c.execute("select kernel_name, grid_size, counter_name, value, duration from counters_collection"):
        if "vitg" not in n: continue
        k = (re.sub(r"\(.*$", "", n).replace("void ", "")[:64], g)
        agg[k][cn] += v; cnt[k][cn] += 1
for k in sorted(agg):
    print(k)
    for cn in sorted(agg[k]):
        print(f"    {cn:28s} {agg[k][cn] / cnt[k][cn]:16.0f}  (avg over {cnt[k][cn]} dispatches)")
