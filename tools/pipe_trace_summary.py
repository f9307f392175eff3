#!/usr/bin/env python3
"""Summary of a step trace of the pipelined record sweeps (k_lusgs_pipe / k_lusgs_pipe4).

A library built with -DAGX_PIPE_TRACE and run with AGX_PIPE_TRACE=<file> appends, per launch,
five timestamps (wall_clock64, 10 ns ticks) per step of the workgroup with the middle ticket:
step start, predecessor seen, cells done, stores acknowledged, barrier passed.
usage: pipe_trace_summary.py <file> [launches from the end, default 2]"""
import sys
import numpy as np

segs, rows, hdr = [], [], ""
for ln in open(sys.argv[1]):
    if ln.startswith("#"):
        if rows:
            segs.append((hdr, np.array(rows)))
        hdr, rows = ln.strip(), []
    else:
        v = [int(x) for x in ln.split()]
        if v[0]:
            rows.append(v)
if rows:
    segs.append((hdr, np.array(rows)))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for hdr, a in segs[-last:]:
    n = len(a)
    print(f"{hdr} steps {n} total us {(a[-1, 4] - a[0, 0]) * 10 / 1e3:.1f}")
    cuts = [0, n // 8, 3 * n // 8, 5 * n // 8, 7 * n // 8, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        seg = a[lo:hi]
        if len(seg) < 2:
            continue
        wait = (seg[:, 1] - seg[:, 0]).mean() * 10
        work = (seg[:, 2] - seg[:, 1]).mean() * 10
        ack = (seg[:, 3] - seg[:, 2]).mean() * 10
        bar = (seg[:, 4] - seg[:, 3]).mean() * 10
        step = np.diff(seg[:, 0]).mean() * 10
        print(f"  steps {lo}-{hi}: wait for the plane below {wait:.0f}  cells {work:.0f}  "
              f"store acknowledgement {ack:.0f}  barrier {bar:.0f}  step {step:.0f} ns")
