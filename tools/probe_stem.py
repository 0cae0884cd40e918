#!/usr/bin/env python
"""In-kernel phase timing of the stem (instrumented build): PF_PROBE=1 PF_LIBPFHIP=.../libpfhip_probe.so python tools/probe_stem.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panoptic_forecasting_amd import lib as pflib  # noqa: E402
from panoptic_forecasting_amd.registry import build_model  # noqa: E402

L = pflib.load()
model = build_model(bench.model_params())
model.load_state_dict(bench.calibrated_state_dict())
batch = bench.make_batch(1, 0, torch.device('cuda'))
buf = (ctypes.c_longlong * 64)()
for rep in range(3):
    model.predict(batch, None)
    torch.cuda.synchronize()
    L.pf_debug_probe_read(buf)
    ts = list(buf)[:5]
    a = list(buf)
    if a[63]:
        print('kernel span %.1f us, %d workgroups, mean workgroup time %.2f us => mean concurrency %.1f workgroups' % (
            (a[61] - a[60]) / 100.0, a[63], a[62] / a[63] / 100.0, a[62] / max(a[61] - a[60], 1)))
    print('wall (100 MHz ticks) of the same workgroup: init %d, stage1 %d, stage2 %d, store %d' % (a[9] - a[8], a[10] - a[9], a[11] - a[10], a[12] - a[11]))
    print('stem phases (cycles): init %d, stage1 %d, stage2 %d, store %d' % (ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3]))
