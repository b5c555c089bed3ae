#!/usr/bin/env python3
"""Phase stamps of the persistent decode step (exl3_pstep_stamps: [op][cu][32] x u64, 100 MHz) -> per-op-type medians in microseconds.
Slots -- streaming wave 0: 0 op start, 1 activation quads seen, 2 its units done; service wave 0: 3 op start, 4 edge passed, 8 inputs in registers
(R / slab lines, tags valid), 9 row sum of squares complete (RMSNorm ops), 5 quads published, 6 all streaming waves done, 10 partial rows summed,
11 outputs issued, 12 atomics acknowledged (ops that add into R), 7 arrived / done.
usage: tools/pstep_stamps.py <stamps.bin> <layers> [json out]"""
import sys, json
import numpy as np

def main():
    path, nl = sys.argv[1], int(sys.argv[2])
    a = np.fromfile(path, dtype=np.uint64)
    nops = 4 * nl + 1
    ncu = a.size // (nops * 32)
    a = a.reshape(nops, ncu, 32).astype(np.int64)
    a0 = a.copy()                                    # raw ticks (slots 13 / 14 are shader-clock counts, not 100 MHz stamps)
    a = (a - a[0, :, 3].min()) / 100.0
    names = ["qkv", "o", "gate_up", "down"]
    out = {"total_us": float(a[-1, :, 7].max()), "workgroups": int(ncu), "ops": {}}
    print(f"total {out['total_us']:.1f} us, {ncu} CUs")
    chain = {  # per op type: ordered (label, from slot, to slot)
        "norm_in": [("edge wait", 3, 4), ("R loads", 4, 8), ("row sum sync", 8, 9), ("scale+rotate->quads", 9, 5)],
        "slab_in": [("op start", 3, 4), ("tagged slab lines (incl. waiting for the producers)", 4, 8), ("finish+rotate->quads", 8, 5)],
        "common": [("wait streamers", 5, 6), ("sum partials", 6, 10), ("outputs issued", 10, 11)],
        "atomic_out": [("atomics acknowledged", 11, 12), ("arrive", 12, 7)],
        "other_out": [("done", 11, 7)],
    }
    for k in range(4):
        ops = [4 * l + k for l in range(1, nl)]
        if not ops: continue
        segs = (chain["norm_in"] if k in (0, 2) else chain["slab_in"]) + chain["common"] + (chain["atomic_out"] if k in (1, 3) else chain["other_out"])
        rec = {}
        line = []
        for lab, i, j in segs:
            d = a[ops][:, :, j] - a[ops][:, :, i]
            rec[lab] = {"median_us": float(np.median(d)), "p90_us": float(np.percentile(d, 90))}
            line.append(f"{lab} {np.median(d):.2f}")
        per = float(np.median([np.median(a[o + 1, :, 3] - a[o, :, 3]) for o in ops]))
        rec["period_us"] = per
        rec["streaming_wave0"] = {"decode_ahead_and_wait_us": float(np.median(a[ops][:, :, 1] - a[ops][:, :, 0])), "stream_us": float(np.median(a[ops][:, :, 2] - a[ops][:, :, 1]))}
        dt = (a0[ops][:, :, 2] - a0[ops][:, :, 1]).astype(np.float64); dc = (a0[ops][:, :, 14] - a0[ops][:, :, 13]).astype(np.float64)
        okc = (dt > 50) & (a0[ops][:, :, 13] > 0)
        if okc.any():
            rec["shader_clock_mhz_during_streaming"] = float(np.median(dc[okc] / dt[okc] * 100.0))
            print(f"           shader clock during the streaming phase (s_memtime / s_memrealtime, wave 0): {rec['shader_clock_mhz_during_streaming']:.0f} MHz")
        rec["last_arrival_to_next_op_edge_passed_us"] = float(np.median([np.median(a[o + 1, :, 4]) - a[o, :, 7].max() for o in ops]))
        rec["finish_spread_max_minus_median_us"] = float(np.median([a[o, :, 7].max() - np.median(a[o, :, 7]) for o in ops]))
        # per streaming wave: done (slot 16 + w) relative to the moment the LAST service wave published its quads (max of slots 28..31); per service wave: publish time
        # relative to the first
        qp = a[ops][:, :, 28:32]; fin = a[ops][:, :, 16:28]
        have = (fin > 0).all(axis=2) & (qp > 0).all(axis=2)
        if have.any():
            rel = (fin - qp.max(axis=2, keepdims=True))[have]
            rec["streaming_waves_done_after_quads_us"] = [round(float(np.median(rel[:, w])), 2) for w in range(12)]
            rec["service_waves_publish_spread_us"] = [round(float(np.median((qp - qp.min(axis=2, keepdims=True))[have][:, s_])), 2) for s_ in range(4)]
            print(f"           streaming waves done after the quads: {rec['streaming_waves_done_after_quads_us']}; service publish spread {rec['service_waves_publish_spread_us']}")
        out["ops"][names[k]] = rec
        print(f"  {names[k]:8s} period {per:5.2f} | " + " | ".join(line))
        print(f"           streamer w0: ahead+wait {rec['streaming_wave0']['decode_ahead_and_wait_us']:.2f}, stream {rec['streaming_wave0']['stream_us']:.2f}; last arrival -> next edge passed {rec['last_arrival_to_next_op_edge_passed_us']:.2f}; finish spread {rec['finish_spread_max_minus_median_us']:.2f}")
    o = nops - 1
    out["ops"]["head"] = {"edge_wait_us": float(np.median(a[o, :, 4] - a[o, :, 3])), "prep_us": float(np.median(a[o, :, 5] - a[o, :, 4])),
                          "wait_streamers_us": float(np.median(a[o, :, 6] - a[o, :, 5])), "finish_us": float(np.median(a[o, :, 7] - a[o, :, 6]))}
    dt = float(np.median(a0[o, :, 2] - a0[o, :, 1])); dc = float(np.median(a0[o, :, 14] - a0[o, :, 13]))
    if dt > 50 and dc > 0: out["ops"]["head"]["shader_clock_mhz_during_streaming"] = dc / dt * 100.0
    print("  head    ", out["ops"]["head"])
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)

if __name__ == "__main__":
    main()
