#!/usr/bin/env python
"""Phase timing inside one conv_split workgroup (instrumented build), in shader clocks, under a full grid.

    make -C panoptic-forecasting_amd/csrc libpfhip_probe.so
    PF_PROBE=1 PF_LIBPFHIP=$PWD/panoptic-forecasting_amd/csrc/libpfhip_probe.so python tools/probe_split.py cin cout h w b nt wide
Per round: [issue loads] [fetch fragments + issue MFMAs] [barrier 1] [wait for the next round's pixels] [split + store] .
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
x = torch.randn(b, cin, h, w, generator=g).cuda()
spec = MiniSpec(cin)
spec.conv('c', [arch.Src(0, 0, cin)], cout, 3, 1)
net = MiniNet(spec, {'c': (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, torch.zeros(cout))})
L.pf_debug_force_conv(4, nt, wide, 0)
buf = (ctypes.c_longlong * 64)()
names = ['loads', 'mfma', 'bar1', 'wait', 'split', 'bar2']
for rep in range(4):
    net.run(x)
    rc = L.pf_debug_probe_read(buf)
    ts = list(buf)[:60]
    if rep >= 2:
        nr = min((cin + 7) // 8, 10)      # 60 stamp slots = 10 rounds
        print('rep %d: rounds 0..%d took %d clocks' % (rep, nr - 1, ts[(nr - 1) * 6 + 5] - ts[0]))
        for r in range(1, min(nr - 1, 9)):
            t = ts[r * 6:r * 6 + 6] + [ts[(r + 1) * 6] if (r + 1) * 6 < 60 else ts[r * 6 + 5]]
            print('  round %d: ' % r + '  '.join('%s %5d' % (names[i], t[i + 1] - t[i]) for i in range(6)))
