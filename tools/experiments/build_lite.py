"""Small variant libraries for kernel A/B runs with the C++ harnesses (decode_step_harness.hip): a full libexl3_hip.so is 26 MB because every
*.kspec.hip is compiled for K = 1..8; a variant library for the 4.0 bpw harness needs K = 4 only, so it is ~7 MB, builds in well under a minute and
adds little to the snapshot that travels to the GPU box.

    python tools/experiments/build_lite.py TAG [-DMACRO ...]     ->  build/lite_TAG/libexl3_hip.so
    LD_LIBRARY_PATH=build/lite_TAG tools/bin/decode_step_harness 32 2 0 8 0 0 1        (the harness's RUNPATH gives way to LD_LIBRARY_PATH)

The macros apply to EVERY translation unit (headers such as exl3_lane_decode.cuh are shared); the other bit widths are stubs that abort.
"""
from __future__ import annotations
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

STUB = r'''
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include "exl3_gemv_args.h"
#include <stdio.h>
#include <stdlib.h>
static void lite_abort(int K) { fprintf(stderr, "libexl3_hip (lite variant build): only K = 4 kernels are in this library, K = %d requested\n", K); abort(); }
#define STUBS(KK) \
    void exl3_gemv2_launch_k##KK(int, int, int, int, dim3, size_t, hipStream_t, const GemvArgs&) { lite_abort(KK); } \
    void exl3_gemm3_launch_k##KK(int, int, int, int, dim3, size_t, hipStream_t, const GemvArgs&) { lite_abort(KK); } \
    int exl3_gemm3_max_waves_k##KK(int, int) { return 4; } \
    void exl3_gemv4_launch_k##KK(int, int, int, int, dim3, size_t, hipStream_t, const GemvArgs&) { lite_abort(KK); }
STUBS(1) STUBS(2) STUBS(3) STUBS(5) STUBS(6) STUBS(7) STUBS(8)
'''


def build_lite(tag: str, defines: list[str]) -> str:
    out_dir = os.path.join(ROOT, "build", "lite_" + tag)
    obj_dir = os.path.join(out_dir, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    stub_src = os.path.join(out_dir, "lite_stubs.hip")
    with open(stub_src, "w") as f:
        f.write(STUB)
    cflags = g.CFLAGS + ["-I" + g.CSRC] + defines
    units = []
    for src in g._sources():
        if src.endswith(".kspec.hip"):
            units.append((os.path.join(g.CSRC, src), ["-DG2_K=4"], src.replace(".kspec.hip", "_k4.o")))
        else:
            units.append((os.path.join(g.CSRC, src), [], src.replace(".hip", ".o")))
    units.append((stub_src, [], "lite_stubs.o"))

    def cc(u):
        src, extra, name = u
        obj = os.path.join(obj_dir, name)
        r = subprocess.run([g.HIPCC] + cflags + extra + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, units))
    lib = os.path.join(out_dir, "libexl3_hip.so")
    r = subprocess.run([g.HIPCC, "-shared", "-fPIC", f"--offload-arch={g.ARCH}", "-fno-gpu-rdc", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lhipblaslt"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    for o in objs:
        os.remove(o)
    os.rmdir(obj_dir)
    return lib


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    lib = build_lite(sys.argv[1], sys.argv[2:])
    print(lib, f"{os.path.getsize(lib) / 1e6:.1f} MB")
