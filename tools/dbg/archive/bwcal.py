"""What this box sustains on plain streaming kernels (calibration for the roofline discussion)."""
import torch
dev = torch.device("cuda:0")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (98, 393, 1572):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x); z = torch.randn(n, device=dev)
    bufs = [torch.randn(n, device=dev) for _ in range(3)]   # rotate to defeat the 256 MB infinity cache
    i = [0]
    def copy():
        i[0] = (i[0] + 1) % 3
        y.copy_(bufs[i[0]])
    def add():
        i[0] = (i[0] + 1) % 3
        torch.add(bufs[i[0]], z, out=y)
    def red():
        i[0] = (i[0] + 1) % 3
        bufs[i[0]].sum()
    def fill():
        y.fill_(1.0)
    for name, f, bytes_ in (("copy", copy, 2 * n * 4), ("add(2r+1w)", add, 3 * n * 4), ("sum(read)", red, n * 4), ("fill(write)", fill, n * 4)):
        us = timeit(f)
        print("%5d MB %-12s %7.1f us  %6.0f GB/s" % (mb, name, us, bytes_ / us / 1e3))
