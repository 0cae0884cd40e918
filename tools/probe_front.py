#!/usr/bin/env python
"""Phase timing inside one conv_front workgroup (instrumented build), in shader clocks, under a full grid.

    make -C panoptic-forecasting_amd/csrc libpfhip_probe_front.so
    PF_PROBE=1 PF_LIBPFHIP=$PWD/panoptic-forecasting_amd/csrc/libpfhip_probe_front.so python tools/probe_front.py [batch]
One step (2 output rows of a 31-column strip) of the workgroup in the middle of the grid, per wave.
"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panoptic_forecasting_amd import lib as pflib  # noqa: E402
from panoptic_forecasting_amd import synth  # noqa: E402
from panoptic_forecasting_amd.registry import build_model  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
h, w = 1024, 2048
params = {'task': 'bg', 'no_gpu': False, 'load_model': None, 'load_best_model': False,
          'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])], 'min_depth': 0.1, 'max_depth': 200},
          'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True, 'final_h': h, 'final_w': w, 'fuse_front': 1}}
m = build_model(params)
with open(os.path.join(ROOT, 'tests', 'golden', 'calib_seed1234.json')) as f:
    m.load_state_dict(synth.make_state_dict(seed=1234, calib=json.load(f)))
m.eval()
inp = synth.make_bg_inputs(b=b, h=h, w=w, seed=1)
cu = {k: v.cuda() for k, v in inp.items()}
cu['seg'] = cu['seg'].to(torch.uint8)
L = pflib.load()
buf = (ctypes.c_longlong * 64)()
names = ['wait+barrier', 'dma issue', 'stores', 'offsets', 'conv1', 'split', 'barrier', 'conv2', 'bias']
for rep in range(4):
    m.predict(cu, None)
    torch.cuda.synchronize()
    L.pf_debug_probe_read(buf)
    ts = list(buf)
    if rep < 2:
        continue
    t0 = min(ts[16 * wv] for wv in range(4))
    print('rep %d: the step took %d clocks' % (rep, max(ts[16 * wv + len(names)] for wv in range(4)) - t0))
    for wv in range(4):
        t = ts[16 * wv:16 * wv + 16]
        print('  wave %d (+%4d): ' % (wv, t[0] - t0) + '  '.join('%s %5d' % (names[i], t[i + 1] - t[i]) for i in range(len(names))))
