#!/usr/bin/env python3
"""Per-unit timeline of one op of the persistent step (diagnostic build -DPS_DBG_UNITS=<op>): streaming waves 0 / 4 / 8 of every CU stamp
[0] op start, [1] before the first predecode, one stamp after each predecoded unit, 'quads seen', then the start of every consumed / streamed unit and the end of the run,
into three extra stamp areas behind the ops'.   usage: tools/pstep_unit_timeline.py <stamps.bin> <layers> <op>"""
import sys
import numpy as np
path, nl, op = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
a = np.fromfile(path, dtype=np.uint64)
nops = 4 * nl + 1
ncu = a.size // ((nops + 3) * 32)
a = a.reshape(nops + 3, ncu, 32).astype(np.int64)
for g in range(3):
    t = a[nops + g] / 100.0                       # [ncu][32] us
    n = (a[nops + g] > 0).sum(axis=1)
    cnt = int(np.median(n))
    sel = n == cnt
    t = t[sel][:, :cnt]
    t = t - t[:, :1]
    med = np.median(t, axis=0)
    print(f"wave {4 * g}: {cnt} stamps on {int(sel.sum())} CUs; median us since op start:", " ".join(f"{x:.2f}" for x in med))
    print(f"          deltas:", " ".join(f"{x:.2f}" for x in np.diff(med)))
