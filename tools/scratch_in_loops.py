#!/usr/bin/env python3
"""Where a kernel's scratch traffic sits: per function of an assembly listing (hipcc -S --cuda-device-only), the scratch loads / stores
inside each loop (backward branch).  A spill outside the loops is a few cycles per workgroup; a reload inside a DMA-pipelined loop is a
vmcnt(0) behind the requests in flight.      python tools/scratch_in_loops.py file.s [name-filter]"""
import re, sys

lines = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for k, s in enumerate(starts):
    name = lines[s].split(":")[0]
    if flt not in name:
        continue
    e = starts[k + 1] if k + 1 < len(starts) else len(lines)
    body = lines[s:e]
    for i, l in enumerate(body):
        if l.startswith(".Lfunc_end"):
            body = body[:i]
            break
    scr = [i for i, l in enumerate(body) if re.search(r"\bscratch_(load|store)", l)]
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch\w+\s+(\.LBB\S+)", l) or re.search(r"s_branch\s+(\.LBB\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            lo = labels[m.group(1)]
            loops.append((lo, i, sum(1 for x in scr if lo <= x <= i)))
    print(name, "| scratch ops:", len(scr), "| loops (start, end, scratch ops inside):", loops)
