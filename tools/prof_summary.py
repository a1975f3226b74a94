#!/usr/bin/env python3
"""Condense rocprofv3 output (rocpd sqlite: kernel trace + PMC passes) into a short text summary."""
import glob
import os
import sqlite3
import sys

out = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "%"
for f in sorted(glob.glob(os.path.join(out, "trace", "*.db"))):
    c = sqlite3.connect(f)
    print("== kernel trace:", os.path.relpath(f, out))
    q = ("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
         "group by name order by 6 desc limit 14")
    tot = c.execute("select sum(end-start) from kernels").fetchone()[0] or 1
    for name, n, avg, mn, mx, sm in c.execute(q):
        short = name if len(name) < 100 else name[:97] + "..."
        print(f"  {short:100s} calls={n:5d} avg_ns={avg:12.0f} min={mn:10d} max={mx:10d} pct={100 * sm / tot:5.1f}")
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*.db")):
        c = sqlite3.connect(f)
        print("== PMC pass", os.path.basename(d))
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
             "where kernel_name like ? group by kernel_name, counter_name")
        last = None
        for k, cn, n, v in c.execute(q, (pat,)):
            if k != last:
                print("  kernel:", k[:100])
                last = k
            print(f"    {cn:32s} n={n:3d} mean={v:18.1f}")
