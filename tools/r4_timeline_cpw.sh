for c in 1 2 4; do for sp in "0,0,0,0" "0,0,2,0" "0,0,4,0"; do echo "== CPW $c SPLITS $sp"; EXL3_HIP_GEMM3_CPW=$c SPLITS=$sp EXL3_HIP_LIB=build/lite_g3timing/libexl3_hip.so BSZ=16 timeout 300 python tools/gemv_timeline.py 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/r4_timeline_bs16_c.txt 2>&1
python - <<PY
import json
for l in open("gpurun_out/r4_timeline_bs16_c.txt"):
    if l.startswith("=="): print(l.strip()); continue
    try:
        nm, js = l.split(" ", 1); d = json.loads(js)
    except Exception: print(l.strip()[:300]); continue
    print(" ", nm, d["workgroups"], "loads_issued", d["loads_issued"], "first_prep", d["first_prep"], "stream_done", d["stream_done"], "slab", d["slab_written"], "phase p50/p90", d["per_wg_phase_us_p50_p90"]["to_first_prep"], d["per_wg_phase_us_p50_p90"]["to_stream_done"], "maxres", d.get("max_resident_workgroups_per_cu_histogram"))
PY
