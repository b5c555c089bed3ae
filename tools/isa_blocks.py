"""Static instruction accounting of one kernel in a hipcc -S listing: basic blocks (label to label), instruction classes per block and the
backward branches (loops).  python tools/isa_blocks.py build/tmp/gemv2_k4.s '_Z17exl3_gemv2_kernelILi4ELi2ELi1ELi1ELi1EEv8GemvArgs'"""
import re, sys, json

def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_endpgm") or op.startswith("s_setpc"): return "branch"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"): return "vmem_rd"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"): return "vmem_wr"
    if op.startswith("global_atomic") or op.startswith("buffer_atomic") or op.startswith("flat_atomic"): return "atomic"
    if op.startswith("scratch_"): return "scratch"
    return "other"

def blocks(path, fn):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(fn + ":"))
    out, cur, order = {}, "entry", ["entry"]
    out[cur] = {"ops": [], "line": start}
    for i in range(start + 1, len(lines)):
        l = lines[i]
        if l.startswith(".Lfunc_end"): break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1); out[cur] = {"ops": [], "line": i}; order.append(cur); continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."): continue
        op = t.split()[0]
        out[cur]["ops"].append((op, t))
    return out, order

def main():
    path, fn = sys.argv[1], sys.argv[2]
    bl, order = blocks(path, fn)
    idx = {b: i for i, b in enumerate(order)}
    tot = {}
    for b in order:
        c = {}
        for op, t in bl[b]["ops"]:
            k = classify(op); c[k] = c.get(k, 0) + 1; tot[k] = tot.get(k, 0) + 1
        tgt = [t.split()[-1] for op, t in bl[b]["ops"] if classify(op) == "branch" and len(t.split()) > 1]
        back = [x for x in tgt if x in idx and idx[x] <= idx[b]]
        desc = " ".join(f"{k}={v}" for k, v in sorted(c.items()))
        print(f"{b:12s} line {bl[b]['line']:7d}  {desc}" + (f"   -> {','.join(tgt)}" if tgt else "") + (f"   LOOP back to {back}" if back else ""))
    print("TOTAL", tot)

if __name__ == "__main__":
    main()
