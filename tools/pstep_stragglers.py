#!/usr/bin/env python3
"""Which workgroups are late?  Phase stamps of the persistent step -> per op type, the finish time (slot 7) of every CU relative to the op's median, averaged over layers:
systematic stragglers (same CU late in every layer) point at an uneven plan, random ones at arrival noise.   usage: tools/pstep_stragglers.py <stamps.bin> <layers>"""
import sys
import numpy as np
path, nl = sys.argv[1], int(sys.argv[2])
a = np.fromfile(path, dtype=np.uint64)
nops = 4 * nl + 1
ncu = a.size // (nops * 32)
a = a.reshape(nops, ncu, 32).astype(np.int64) / 100.0
names = ["qkv", "o", "gate_up", "down"]
for k in range(4):
    ops = [4 * l + k for l in range(2, nl)]
    for slot, lab in ((5, "quads published"), (7, "done")):
        t = np.stack([a[o, :, slot] - np.median(a[o, :, slot]) for o in ops])       # [layers][ncu]
        mean = t.mean(axis=0); sd = t.std(axis=0)
        order = np.argsort(-mean)[:8]
        print(f"{names[k]:8s} {lab:16s}: spread of per-CU means {mean.max() - mean.min():.2f} us (p95-p5 {np.percentile(mean, 95) - np.percentile(mean, 5):.2f}); per-layer sd median {np.median(sd):.2f};"
              f" latest CUs {[(int(c), round(float(mean[c]), 2)) for c in order]}")
    # by XCD (cu % 8) and by position inside the XCD
    t = np.stack([a[o, :, 7] - np.median(a[o, :, 7]) for o in ops]).mean(axis=0)
    print(f"          done by XCD (cu % 8): {[round(float(t[x::8].mean()), 2) for x in range(8)]}")
# ---- where do the stragglers lose the time?  intervals of the service wave 0 (slots 3 op start, 4 edge passed, 8 inputs in registers, 9 row sums, 5 quads published, 6 streamers done, 10, 11, 7 done)
late = None
for k in (2,):
    ops = [4 * l + k for l in range(2, nl)]
    t7 = np.stack([a[o, :, 7] - np.median(a[o, :, 7]) for o in ops]).mean(axis=0)
    late = np.argsort(-t7)[:6]; early = np.argsort(t7)[:6]
    segs = [(3, 4), (4, 8), (8, 9), (9, 5), (5, 6), (6, 10), (10, 11), (11, 7)]
    med = {sg: np.median(np.stack([a[o, :, sg[1]] - a[o, :, sg[0]] for o in ops])) for sg in segs}
    print("gate_up intervals (median over CUs):", {f"{i}->{j}": round(float(v), 2) for (i, j), v in med.items()})
    for grp, lab in ((late, "latest"), (early, "earliest")):
        for c in grp:
            d = {f"{i}->{j}": round(float(np.mean([a[o, c, j] - a[o, c, i] for o in ops])), 2) for (i, j) in segs}
            st = float(np.mean([a[o, c, 3] - np.median(a[o, :, 3]) for o in ops]))
            print(f"  {lab} CU {int(c):3d} (XCD {int(c) % 8}, slot {int(c) // 8}): op start {st:+.2f} vs median;", d)
