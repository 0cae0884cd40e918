#!/usr/bin/env python
"""Known-byte-count kernel for calibrating FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md §HBM):
a 256 MiB fp32 device-to-device elementwise copy, 5 launches; each reads 256 MiB and writes 256 MiB
(larger than the 256 MiB Infinity Cache together with its destination, so it streams from HBM)."""
import torch

n = 64 * 1024 * 1024
x = torch.rand(n, device='cuda')
y = torch.empty_like(x)
torch.cuda.synchronize()
for _ in range(5):
    torch.add(x, 1.0, out=y)      # vectorized elementwise kernel: 4 B read + 4 B written per element
torch.cuda.synchronize()
print('calib bytes per launch: read %d written %d' % (4 * n, 4 * n))
