import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from exllamav3_amd import ext, _lib
from oracle import exl3_oracle as o
dev = torch.device("cuda:0")
ext.set_gemv_gen(2); ext.set_gemv_variant(1)
def ws(nfloats):
    buf = torch.empty(nfloats, dtype=torch.float32, device=dev)
    _lib.lib().exl3_debug_copy_workspace(buf.data_ptr(), 0, nfloats * 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); return buf
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
k, K, m = 384, 4, 8
mats = [o.synth_linear(k, n, K, seed=3 + i, realistic=True) for i, n in enumerate((512, 256, 256))]
Bs = [T(t[0]) for t in mats]; su = [T(t[1]) for t in mats]; sv = [T(t[2]) for t in mats]
rng = np.random.default_rng(1)
w = T((1 + 0.1 * rng.standard_normal(k)).astype(np.float16))
nAB = nBC = nxn = 0
for trial in range(30):
    r = T((rng.standard_normal((m, k)) * rng.uniform(0.2, 5)).astype(np.float16))
    xh = [torch.empty((m, k), dtype=torch.half, device=dev) for _ in range(3)]; xs = [torch.empty((m, k // 128), dtype=torch.float32, device=dev) for _ in range(3)]
    xn = torch.empty((m, k), dtype=torch.half, device=dev)
    ext.glue_norm(None, 0, None, None, r, w, 1e-5, su, xh, xs, m, xn_out=xn)
    for cb in (0, 2):
        _, S = ext.exl3_gemv_ex(None, xh, xs, Bs, None, None, None, m, False, cb == 2, 3); n = 8 * S * m * 128; a = ws(n).clone()
        _, S = ext.exl3_gemv_ex(xn, None, None, Bs, None, su, None, m, False, cb == 2, 2); b = ws(n).clone()
        ss = torch.empty((m, k // 128), dtype=torch.float32, device=dev)
        ext.glue_resid(None, 0, None, None, r, ss, m)
        _, S = ext.exl3_gemv_ex_norm(r, w, ss, 1e-5, Bs, None, su, None, m, False, cb == 2, 2); c = ws(n).clone()
        nAB += int(not torch.equal(a, b)); nBC += int(not torch.equal(b, c))
    # xn recomputed on the host from ss
    rmf = torch.rsqrt(ss.sum(1, keepdim=True) / k + 1e-5)
    xn2 = (r.float() * w.float() * rmf).half()
    nxn += int(not torch.equal(xn, xn2))
print("rotated vs raw differ:", nAB, "/60   raw vs in-norm differ:", nBC, "/60   xn host-recompute differs:", nxn, "/30")
