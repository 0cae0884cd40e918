#!/usr/bin/env python
"""Phase timing inside one conv_s4 workgroup (instrumented build), in shader clocks, under a full grid.

    make -C panoptic-forecasting_amd/csrc libpfhip_probe_s4.so
    PF_PROBE=1 PF_LIBPFHIP=$PWD/panoptic-forecasting_amd/csrc/libpfhip_probe_s4.so python tools/probe_s4.py cin cout h w b nt wide
Per round: [barrier] [issue the next stage's DMA] [fragments + MFMAs] [wait for the stage].
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import MiniNet, MiniSpec  # noqa: E402
from panoptic_forecasting_amd import hardnet_arch as arch  # noqa: E402
from panoptic_forecasting_amd import lib as pflib  # noqa: E402

cin, cout, h, w, b, nt, wide = [int(x) for x in sys.argv[1:8]]
L = pflib.load()
g = torch.Generator().manual_seed(0)
x = torch.randn(b, 8, h, w, generator=g).cuda()
spec = MiniSpec(8)
c0 = spec.conv('c0', [arch.Src(0, 0, 8)], cin, 1, 1)
c = spec.conv('c', [arch.Src(c0, 0, cin)], cout, 3, 1)
spec.conv('c2', [arch.Src(c, 0, cout)], 4, 1, 1)      # a reader: keeps `c` in the packed layout
net = MiniNet(spec, {'c0': (torch.randn(cin, 8, 1, 1, generator=g), torch.zeros(cin)),
                     'c': (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, torch.zeros(cout)),
                     'c2': (torch.randn(4, cout, 1, 1, generator=g), torch.zeros(4))})
L.pf_debug_force_conv(5, nt, wide, 0)
buf = (ctypes.c_longlong * 64)()
names = ['barrier', 'issue', 'mfma', 'wait']
for rep in range(4):
    net.run(x)
    rc = L.pf_debug_probe_read(buf)
    ts = list(buf)[:60]
    if rep >= 2:
        nr = min((((cin + 3) // 4) + 1) // 2, 14)
        print('rep %d: %d rounds took %d clocks, epilogue %d' % (rep, nr, ts[(nr - 1) * 4 + 3] - ts[0], ts[59] - ts[58]))
        for r in range(0, nr - 1):
            t = ts[r * 4:r * 4 + 5]
            print('  round %2d: ' % r + '  '.join('%s %5d' % (names[i], t[i + 1] - t[i]) for i in range(4)))
