#!/usr/bin/env python
"""Roofline lines for the kernels either side of the forecast path (scope rows f2/f3), one JSON line per kernel.

    python tools/bench_stages.py [--steps 20] [--batch 4] [--instances 60]

All are HBM-bound elementwise / gather kernels; `achieved` = algorithmic bytes (unique input + output bytes of the
launch) / hipEvent time on the launch stream.  `cpu_baseline` = the C oracle (one core) on one image.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panoptic_forecasting_amd import hop_io, lib as pflib, panoptic as pp  # noqa: E402

H, W = 1024, 2048
PEAK = 8000.0


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    pflib.profile(True)
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    recs = pflib.profile_results()
    pflib.profile(False)
    return recs


def line(rec, steps, extra):
    ms = rec['ms'] / rec['launches']
    gbs = rec['bytes'] / rec['launches'] / (ms * 1e-3) / 1e9
    out = {'kernel': rec['label'], 'avg_launch_us': ms * 1e3,
           'roofline': {'bound': 'hbm', 'achieved': gbs, 'peak': PEAK, 'unit': 'GB/s', 'frac': gbs / PEAK, 'traffic': None}}
    out.update(extra)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--instances', type=int, default=60)
    ap.add_argument('--no-cpu', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    B = a.batch
    g = torch.Generator().manual_seed(0)
    seg = torch.randint(0, 19, (B, H, W), generator=g, dtype=torch.uint8).to(dev)
    depth = (torch.rand(B, H, W, generator=g) * 100).to(dev)
    for r in timed(lambda: hop_io.device_export(seg, depth, hop_io.SEG_TRAINID_TO_ID), a.steps):
        line(r, a.steps, {'workload': 'hop export: %d x 1024x2048 label+depth maps' % B})
    _, q = hop_io.device_export(None, depth)
    for r in timed(lambda: hop_io.device_load_depth(q, 0.1, 200.0), a.steps):
        line(r, a.steps, {'workload': 'hop load: %d x 1024x2048 u16 depth codes' % B})

    n = a.instances
    counts = [n] * B
    masks = (torch.rand(B * n, 28, 28, generator=g) * 0.7 + 0.2).to(dev)
    boxes = torch.stack([torch.rand(B * n, generator=g) * W, 300 + torch.rand(B * n, generator=g) * 500,
                         20 + torch.rand(B * n, generator=g) ** 2 * 400, 20 + torch.rand(B * n, generator=g) ** 2 * 300], 1).to(dev)
    depths = (5 + torch.rand(B * n, generator=g) * 80).to(dev)
    classes = torch.randint(0, 8, (B * n,), generator=g).to(dev)
    mg = pp.PanopticMerger()
    out = {}

    def run_merge():
        out['seg'] = mg.merge(masks, boxes, depths, classes, counts, background=seg, out_dtype=torch.int32)
    cpu = None
    if not a.no_cpu:
        from oracle import panoptic as op
        t0 = time.perf_counter()
        want = op.merge([masks[:n].cpu()], [boxes[:n].cpu()], [depths[:n].cpu()], [classes[:n].cpu()], H, W,
                        background=seg[:1].cpu().long())
        dt = time.perf_counter() - t0
        run_merge()
        assert torch.equal(out['seg'][0].long().cpu(), want[0]), 'merge differs from the oracle'
        cpu = {'value': 1.0 / dt, 'unit': 'images/s', 'cores': 1, 'kind': 'port',
               'sample': '1 image, %d instances, C oracle (instance-after-instance like the reference), %.1f s' % (n, dt)}
    for r in timed(run_merge, a.steps):
        if 'merge_kernel' in r['label']:
            ms = r['ms'] / r['launches']
            line(r, a.steps, {'workload': 'panoptic merge: %d images x %d instances, 1024x2048' % (B, n),
                              'images_per_s': B / (ms * 1e-3), 'cpu_baseline': cpu, 'parity': 'bit-exact vs oracle' if cpu else None})
    for r in timed(lambda: pp.encode(out['seg'], convert=True), a.steps):
        line(r, a.steps, {'workload': 'panoptic encode: %d images' % B})


if __name__ == '__main__':
    main()
