#!/usr/bin/env python3
"""A/B copies of the library that differ only in the persistent-step unit: tools/ps_alt.py <tag> [-DMACRO ...] -> build/alt_<tag>/libexl3_hip.so
(all other objects are taken from build/obj; run the harness with LD_LIBRARY_PATH=build/alt_<tag>)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
tag, defs = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "build", "alt_" + tag); os.makedirs(out, exist_ok=True)
obj = os.path.join(out, "exl3_pstep.o")
cmd = [g.HIPCC] + g.CFLAGS + defs + ["-c", os.path.join(g.CSRC, "exl3_pstep.hip"), "-o", obj]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode: sys.exit(r.stderr)
objs = [os.path.join(g.BUILD, u[2]) for u in g._units() if u[2] != "exl3_pstep.o"] + [obj]
r = subprocess.run([g.HIPCC, "-shared", "-fPIC", f"--offload-arch={g.ARCH}", "-fno-gpu-rdc", "-o", os.path.join(out, "libexl3_hip.so")] + objs + ["-L/opt/rocm/lib", "-lhipblaslt"], capture_output=True, text=True)
if r.returncode: sys.exit(r.stderr)
print(os.path.join(out, "libexl3_hip.so"))
