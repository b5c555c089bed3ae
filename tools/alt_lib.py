#!/usr/bin/env python3
"""A/B copies of the library that differ in ONE translation unit: tools/alt_lib.py <tag> <object name, e.g. exl3_gemm3_k4.o> [-DMACRO ...]
-> build/alt_<tag>/libexl3_hip.so (all other objects from build/obj; run with EXL3_HIP_LIB=build/alt_<tag>/libexl3_hip.so or LD_LIBRARY_PATH)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
tag, objname, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
unit = [u for u in g._units() if u[2] == objname]
if not unit: sys.exit(f"no unit produces {objname}: {[u[2] for u in g._units()]}")
src, extra, _ = unit[0]
out = os.path.join(ROOT, "build", "alt_" + tag); os.makedirs(out, exist_ok=True)
obj = os.path.join(out, objname)
r = subprocess.run([g.HIPCC] + g.CFLAGS + extra + defs + ["-c", os.path.join(g.CSRC, src), "-o", obj], capture_output=True, text=True)
if r.returncode: sys.exit(r.stderr)
objs = [os.path.join(g.BUILD, u[2]) for u in g._units() if u[2] != objname] + [obj]
r = subprocess.run([g.HIPCC, "-shared", "-fPIC", f"--offload-arch={g.ARCH}", "-fno-gpu-rdc", "-o", os.path.join(out, "libexl3_hip.so")] + objs + ["-L/opt/rocm/lib", "-lhipblaslt"], capture_output=True, text=True)
if r.returncode: sys.exit(r.stderr)
print(os.path.join(out, "libexl3_hip.so"))
