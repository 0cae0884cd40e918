#!/usr/bin/env python
"""Time single conv_s4 3x3 layers (packed-pair sources and destination) at a forced shape, in isolation.

    python tools/time_s4_layer.py H W B cin:cout:nt:wide [cin:cout:nt:wide ...]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import MiniNet, MiniSpec  # noqa: E402
from panoptic_forecasting_amd import hardnet_arch as arch  # noqa: E402
from panoptic_forecasting_amd import lib as pflib  # noqa: E402

h, w, b = [int(x) for x in sys.argv[1:4]]
L = pflib.load()
g = torch.Generator().manual_seed(0)
x = torch.randn(b, 8, h, w, generator=g).cuda()
for case in sys.argv[4:]:
    cin, cout, nt, wide = [int(v) for v in case.split(':')]
    spec = MiniSpec(8)
    c0 = spec.conv('c0', [arch.Src(0, 0, 8)], cin, 1, 1)
    c = spec.conv('c', [arch.Src(c0, 0, cin)], cout, 3, 1)
    spec.conv('c2', [arch.Src(c, 0, cout)], 4, 1, 1)      # a reader: keeps `c` in the packed layout
    net = MiniNet(spec, {'c0': (torch.randn(cin, 8, 1, 1, generator=g), torch.zeros(cin)),
                         'c': (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, torch.zeros(cout)),
                         'c2': (torch.randn(4, cout, 1, 1, generator=g), torch.zeros(4))})
    L.pf_debug_force_conv(5, nt, wide, 0)
    for _ in range(2):
        net.run(x)
    torch.cuda.synchronize()
    pflib.profile(True)
    for _ in range(5):
        net.run(x)
    torch.cuda.synchronize()
    recs = pflib.profile_results()
    pflib.profile(False)
    for r in recs:
        if 'conv_s4_kernel' in r['label']:
            print('%3d -> %3d  %dx%d B=%d  %-40s %8.1f us' % (cin, cout, h, w, b, r['label'].replace('void pf::', '').replace('(pf::ConvArgs)', ''),
                                                            r['ms'] / r['launches'] * 1e3))
    net.close()
L.pf_debug_force_conv(0, 0, 0, 0)
