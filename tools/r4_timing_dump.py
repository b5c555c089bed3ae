#!/usr/bin/env python3
"""Offline reader of the harness's H_DUMP_TIMING file (4 MiB from 48 MiB of the workspace = 8 slots x 8192 workgroups x 8 uint64): per slot (= one of the
step's last 8 GEMV launches) the per-workgroup phases of tools/gemv_timeline.py, measured INSIDE the replayed hipGraph (cold weights, real neighbours)."""
import sys, json
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(8, 8192, 8)
for sl in range(8):
    t = raw[sl].astype(np.int64)
    t = t[(t[:, 0] > 0) & (t[:, 5] >= t[:, 0]) & (t[:, 5] - t[:, 0] < 1000000)]
    if not len(t): continue
    # the slot may hold stale rows of an earlier, larger launch: keep the newest cluster (within 200 us of the newest entry)
    t = t[t[:, 0] > t[:, 0].max() - 20000]
    xcc = t[:, 6] & 15
    rel = np.zeros((len(t), 6))
    for x in range(16):
        s_ = xcc == x
        if s_.any(): rel[s_] = (t[s_, :6] - t[s_, 0].min()) * 0.01
    d = np.diff(t[:, :6].astype(float), axis=1) * 0.01
    pct = lambda v: [round(float(np.percentile(v, q)), 2) for q in (0, 50, 90, 100)]
    life = (t[:, 5] - t[:, 0]) * 0.01
    row = {"slot": sl, "workgroups": int(len(t)), "entry": pct(rel[:, 0]), "loads_issued": pct(rel[:, 1]), "first_prep": pct(rel[:, 2]), "stream_done": pct(rel[:, 3]),
           "slab_written": pct(rel[:, 5]), "phase_to_first_prep": pct(d[:, 1]), "phase_stream": pct(d[:, 2]), "phase_out": pct(d[:, 4]),
           "clock_GHz": round(float(np.median(t[:, 7][life > 0] / life[life > 0])) * 1e-3, 3)}
    print(json.dumps(row))
