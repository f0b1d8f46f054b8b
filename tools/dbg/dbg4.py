import sys, torch
sys.path.insert(0, '.')
import bench
from flamo_amd import ops
dev = torch.device("cuda:0")
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(32, 96000, 8, device=dev)
orig = ops._Irfft.backward
def dbg(ctx, gy):
    print("gy shape", tuple(gy.shape), "strides", gy.stride(), "contig", gy.is_contiguous(), "planar", ops._is_planar(gy))
    return orig(ctx, gy)
ops._Irfft.backward = staticmethod(dbg)
y = model(x)
print("y strides", y.stride())
loss = (y ** 2).mean()
loss.backward()
