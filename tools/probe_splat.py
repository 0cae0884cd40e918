#!/usr/bin/env python
"""Phase timing of one raster_kernel workgroup (instrumented build, PF_PROBE=1 PF_LIBPFHIP=.../libpfhip_probe.so)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from panoptic_forecasting_amd import lib as pflib
from panoptic_forecasting_amd.pc_transform_model import WarpSplat
L = pflib.load()
inp = bench.make_batch(1, 0, torch.device('cuda'))
ws = WarpSplat()
buf = (ctypes.c_longlong * 64)()
for rep in range(3):
    ws(inp['depth'], inp['depth_mask'], inp['seg'], inp['intrinsics'], inp['extrinsics'], inp['target_T'],
       Kinv=inp['intrinsics_inv'], Einv=inp['extrinsics_inv'], per_frame=True, want_result2d=False)
    torch.cuda.synchronize()
    L.pf_debug_probe_read(buf)
    t = list(buf)
    print('   of which init + box scan %.2f us' % ((t[4] - t[0]) / 100.))
    print('raster workgroup: scan+rasterise %.2f us (%d source tiles hit), resolve %.2f us' % ((t[1] - t[0]) / 100., t[3], (t[2] - t[1]) / 100.))
