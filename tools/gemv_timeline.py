#!/usr/bin/env python3
"""Phase timeline of the gen-2 GEMV workgroups (diagnostics build with -DG2_TIMING):
   EXL3_HIP_LIB=build/libexl3_hip_timing.so python tools/gemv_timeline.py
Every workgroup's wave 0 records 100 MHz timestamps: entry, ring loads issued, first prep done, stream done, partials in LDS,
slab written.  Printed per call type of the Llama-3.1-8B decode step: percentiles relative to the earliest workgroup entry."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exllamav3_amd import ext, _lib
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama

dev = torch.device("cuda:0")
model = SyntheticEXL3Llama(SHAPES["llama-3.1-8b"], K=4, cb=2, device=dev, kv_bits=4, layers=2)
model.alloc_state(int(os.environ.get("BSZ", "1")))
pipeline = os.environ.get("PIPELINE", "glue")       # glue | resid | fx
{"resid": model.decode_step_resid, "fx": model.decode_step_fx}.get(pipeline, model.decode_step_fused)(); torch.cuda.synchronize()
if os.environ.get("SPLITS"):          # "qkv,o,gu,down" forced split-k factors (0 = the library's choice)
    q_, o_, g_, d_ = (int(v) for v in os.environ["SPLITS"].split(","))
    model.split = {"qkv": q_, "o": o_, "gu": g_, "down": d_}
calls = model.gemv_calls(pipeline)
names = ["qkv", "o", "gate_up", "down"]
grids = {"qkv": 48, "o": 32, "gate_up": 224, "down": 32, "lm_head": 1002}
buf = torch.zeros(8 * 8192, dtype=torch.int64, device=dev)
out = {}
prev_newest = 0
for i, nm in enumerate(names):
    c = calls[4 + i] if i < 4 else calls[-1]          # second layer's instances
    for rep in range(3):
        c(); torch.cuda.synchronize()
        if rep == 1:
            _lib.lib().exl3_debug_copy_workspace(buf.data_ptr(), 48 << 20, buf.numel() * 8, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            tt = buf.cpu().numpy().reshape(-1, 8)
            tt = tt[(tt[:, 0] > 0) & (tt[:, 5] >= tt[:, 0]) & (tt[:, 5] - tt[:, 0] < 1000000)]   # uninitialised rows are garbage
            prev_newest = int(tt[:, 5].max())           # everything up to the second repetition is "old"
    _lib.lib().exl3_debug_copy_workspace(buf.data_ptr(), 48 << 20, buf.numel() * 8, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(-1, 8)
    t = t[(t[:, 0] > 0) & (t[:, 5] >= t[:, 0]) & (t[:, 5] - t[:, 0] < 1000000)]
    # the number of workgroups of this launch: rows whose entry stamp is within 1 ms of the newest
    t = t[t[:, 0] > prev_newest]                        # stamps of this call type only (the workspace keeps older ones)
    prev_newest = int(t[:, 5].max())
    # s_memrealtime is per XCD (offsets of a few us between XCDs were observed), so phases are taken relative to the earliest
    # workgroup entry on the SAME XCD
    xcc = t[:, 6].astype(int) & 15
    rel = np.zeros((len(t), 6))
    for x in range(16):
        sel = xcc == x
        if sel.any(): rel[sel] = (t[sel, :6] - t[sel, 0].min()) * 0.01
    row_off = [round(float((t[xcc == x, 0].min() - t[:, 0].min()) * 0.01), 2) for x in range(8) if (xcc == x).any()]
    row = {"workgroups": int(len(t)), "xcd_first_entry_offsets_us": row_off}
    for j, ph in enumerate(["entry", "loads_issued", "first_prep", "stream_done", "partials", "slab_written"]):
        v = rel[:, j]
        row[ph] = [round(float(np.percentile(v, q)), 2) for q in (0, 50, 100)]
    # per-workgroup phase durations (stamp j - stamp j-1): what one workgroup spends where, independent of when it started
    d = np.diff(t[:, :6].astype(float), axis=1) * 0.01
    row["per_wg_phase_us_p50_p90"] = {ph: [round(float(np.percentile(d[:, j], 50)), 2), round(float(np.percentile(d[:, j], 90)), 2)]
                                      for j, ph in enumerate(["to_loads_issued", "to_first_prep", "to_stream_done", "to_partials", "to_slab_written"])}
    life = (t[:, 5] - t[:, 0]).astype(float) * 0.01
    if os.environ.get("BSZ", "1") not in ("1", "2", "3", "4"):   # generation 3 stores shader cycles of the workgroup in column 7
        row["shader_clock_GHz_median"] = round(float(np.median(t[:, 7][life > 0] / life[life > 0])) * 1e-3, 3)
    row["xcc_counts"] = np.bincount(t[:, 6].astype(int) & 15, minlength=8).tolist()
    if (t[:, 6] >> 8).any():
        # generation-3 timing builds also record HW_ID: workgroups per CU (cu_id [11:8], sh_id [12], se_id [15:13] of HW_REG_HW_ID) and the stream phase by co-residency
        hw = (t[:, 6] >> 8).astype(np.int64)
        cu_key = (t[:, 6].astype(np.int64) & 15) * 4096 + ((hw >> 8) & 0xff)
        uniq, inv, cnt = np.unique(cu_key, return_inverse=True, return_counts=True)
        row["cus_used"] = int(len(uniq)); row["workgroups_per_cu_histogram"] = {int(k): int(v) for k, v in zip(*np.unique(cnt, return_counts=True))}
        per = cnt[inv]
        # residency census: the largest number of workgroups whose [entry, last stamp] intervals overlap on one CU
        mx = []
        for u in range(len(uniq)):
            ev = sorted([(int(a), 1) for a in t[inv == u, 0]] + [(int(b), -1) for b in t[inv == u, 5]], key=lambda e: (e[0], e[1]))
            c = m_ = 0
            for _, dlt in ev:
                c += dlt; m_ = max(m_, c)
            mx.append(m_)
        row["max_resident_workgroups_per_cu_histogram"] = {int(k): int(v) for k, v in zip(*np.unique(mx, return_counts=True))}
        row["stream_us_p50_by_workgroups_on_cu"] = {int(k): round(float(np.median(d[per == k, 2])), 2) for k in np.unique(per)}
    out[nm] = row
    print(nm, json.dumps(row), flush=True)
