import torch, time
dev = torch.device("cuda:0")
n = 1 << 29  # halves: 1 GiB
a = torch.empty(n, dtype=torch.half, device=dev); b = torch.empty(n, dtype=torch.half, device=dev)
def t(fn, bytes_):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return bytes_ / ms / 1e9
print("fill  (write only)  %.2f TB/s" % t(lambda: a.fill_(1.0), n * 2))
print("copy  (read+write)  %.2f TB/s (sum of both directions)" % t(lambda: b.copy_(a), n * 4))
print("sum   (read only)   %.2f TB/s" % t(lambda: a.view(torch.int16).sum(), n * 2))
# 117 MB fill (one gate matrix)
c = torch.empty(4096 * 14336, dtype=torch.half, device=dev)
print("fill 117 MB         %.2f TB/s" % t(lambda: c.fill_(1.0), c.numel() * 2))
