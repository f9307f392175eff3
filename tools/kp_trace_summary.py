"""Summarise the step timestamps a -DAGX_KP_TRACE build of the library writes
(AGX_KP_TRACE=<file>): per phase of a step of the middle k-plane, in ns."""
import sys
import numpy as np
lines = open(sys.argv[1]).read().split('\n')
blocks, cur = [], None
for l in lines:
    if l.startswith('#'):
        cur = []
        blocks.append((l, cur))
    elif l.strip():
        cur.append([int(v) for v in l.split()])
for hdr, rows in blocks[:int(sys.argv[2]) if len(sys.argv) > 2 else 2]:
    a = np.array(rows, dtype=np.int64)
    a = a[a[:, 0] > 0]
    d = np.diff(a, axis=1) * 10      # 100 MHz ticks -> ns
    step = np.diff(a[:, 0]) * 10
    print(hdr, "steps", len(a), "total us", (a[-1, 5] - a[0, 0]) * 10 / 1000)
    for lo, hi in [(0, 50), (50, 150), (200, 300), (400, 500)]:
        seg = d[lo:hi]
        if len(seg) == 0:
            continue
        # stamps (k_lusgs_kp): 0 top, 1 requests issued, 2 in-plane terms done, 3 the
        # k-neighbours' x has landed, 4 k-term + stores + record done, 5 barrier + rotation
        print(f"  steps {lo}-{hi}: issue {seg[:,0].mean():.0f} in-plane {seg[:,1].mean():.0f} "
              f"poll {seg[:,2].mean():.0f} k-term+store+record {seg[:,3].mean():.0f} "
              f"barrier+rotate {seg[:,4].mean():.0f}  step {step[lo:hi].mean():.0f} ns")
