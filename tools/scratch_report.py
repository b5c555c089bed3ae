#!/usr/bin/env python3
"""Private-segment (scratch) bytes of every kernel in build/obj/*.o.  Any scratch use costs a launch 1.5-3 us on MI355X
(profiles/r01_launch_chain_microbench.json), so the decode-path kernels must report 0.  usage: scratch_report.py [--all]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={co}"], check=True, capture_output=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, name = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m: name = m.group(1)
        m = re.match(r"\s*\.private_segment_fixed_size:\s+(\d+)", line)
        if m and name: out.append((name, int(m.group(1)))); name = None
    return out


def report(pattern="*.o"):
    res = {}
    for obj in sorted(glob.glob(os.path.join(ROOT, "build", "obj", pattern))):
        res[os.path.basename(obj)] = kernels_of(obj)
    return res


if __name__ == "__main__":
    bad = 0
    for obj, ks in report().items():
        off = [(n, b) for n, b in ks if b]
        print(f"{obj}: {len(ks)} kernels, {len(off)} with scratch")
        for n, b in off:
            bad += 1
            if "--all" in sys.argv or bad <= 40: print(f"    {b:5d} B  {n}")
    sys.exit(0)
