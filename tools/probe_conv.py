#!/usr/bin/env python
"""In-kernel phase timing of one conv_wave launch (instrumented build).

    make -C panoptic-forecasting_amd/csrc libpfhip_probe.so
    PF_PROBE=1 PF_LIBPFHIP=$PWD/panoptic-forecasting_amd/csrc/libpfhip_probe.so \
        python tools/probe_conv.py cin cout k h w  mh nt wk
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

cin, cout, k, h, w, mh, nt, wk = [int(x) for x in sys.argv[1:9]]
L = pflib.load()
g = torch.Generator().manual_seed(0)
x = torch.randn(1, cin, h, w, generator=g).cuda()
spec = MiniSpec(cin)
spec.conv('c', [arch.Src(0, 0, cin)], cout, k, 1)
net = MiniNet(spec, {'c': (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, torch.zeros(cout))})
L.pf_debug_force_conv(2, mh, nt, wk)
buf = (ctypes.c_longlong * 64)()
for rep in range(4):
    net.run(x)
    rc = L.pf_debug_probe_read(buf)
    ts = [t for t in buf if t]
    if rep >= 2 and ts:
        d = [ts[i + 1] - ts[i] for i in range(len(ts) - 1)]
        print('rep %d: total %d cycles; deltas: %s' % (rep, ts[-1] - ts[0], d))
