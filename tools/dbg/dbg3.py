import sys, torch
sys.path.insert(0, '.')
from flamo_amd import ops
dev = torch.device("cuda:0")
for dt in (torch.float32, torch.complex64, torch.complex128):
    for (B, R, C) in [(2, 49, 4), (1, 49, 16), (3, 1000, 8), (2, 8, 1000), (1, 301, 169), (2, 96000, 8), (1, 5, 3), (1, 257, 1), (1,49,2)]:
        x = torch.randn(B, R, C, device=dev).to(dt) if not dt.is_complex else torch.randn(B, R, C, dtype=dt, device=dev)
        y = ops._transpose(x.contiguous(), B, R, C).view(B, C, R)
        ok = torch.equal(y, x.transpose(1, 2).contiguous())
        print(dt, (B, R, C), ok)
