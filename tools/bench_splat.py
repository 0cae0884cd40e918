#!/usr/bin/env python
"""Per-kernel timing of the warp/splat pair on the bench workload (B frames x 3 input frames, 1024x2048, per-frame
z-buffers), hipEvents on the launch stream.  One JSON line: {kernel: avg us}, algorithmic GB/s of the pair.

    python tools/bench_splat.py [--batch 16] [--steps 10] [--term short|mid]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panoptic_forecasting_amd import lib as pflib  # noqa: E402
from panoptic_forecasting_amd.pc_transform_model import WarpSplat  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--term', default='short')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    inp = bench.make_batch(a.batch, 100, dev, a.term)
    sp = WarpSplat()
    run = lambda: sp(inp['depth'], inp['depth_mask'], inp['seg'], inp['intrinsics'], inp['extrinsics'], inp['target_T'],
                     Kinv=inp.get('intrinsics_inv'), Einv=inp.get('extrinsics_inv'), per_frame=True, want_result2d=False)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    pflib.profile(True)
    for _ in range(a.steps):
        run()
    torch.cuda.synchronize()
    recs = pflib.profile_results()
    pflib.profile(False)
    out = {r['label']: round(r['ms'] / r['launches'] * 1e3, 1) for r in recs}
    total_us = sum(out.values())
    pts = a.batch * 3 * bench.H * bench.W
    out['pair_us'] = round(total_us, 1)
    out['algorithmic_GBps'] = round(11.0 * pts / (total_us * 1e-6) / 1e9, 1)
    out['batch'] = a.batch
    print(json.dumps(out))


if __name__ == '__main__':
    main()
