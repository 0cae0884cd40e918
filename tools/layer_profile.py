#!/usr/bin/env python
"""Per-op timing of one bg_forecast step (hipEvents via the library's profiling hooks).

    python tools/layer_profile.py [--batch B] [--steps 5]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panoptic_forecasting_amd import lib as pflib  # noqa: E402
from panoptic_forecasting_amd.registry import build_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--force', default='', help='kind,p0,p1,p2 for pf_debug_force_conv')
ap.add_argument('--size', default='', help='HxW instead of 1024x2048 (the kernel tables have no rows for other sizes: the heuristic choice)')
args = ap.parse_args()
for kv in filter(None, os.environ.get('PF_OPTS', '').split(',')):
    k, v = kv.split('=')
    pflib.check(pflib.load().pf_set_option(k.encode(), int(v)), 'pf_set_option')
if args.force:
    pflib.check(pflib.load().pf_debug_force_conv(*[int(v) for v in args.force.split(',')]), 'pf_debug_force_conv')
if args.size:
    bench.H, bench.W = [int(v) for v in args.size.lower().split('x')]
model = build_model(bench.model_params(final_h=bench.H, final_w=bench.W))
model.load_state_dict(bench.calibrated_state_dict())
batch = bench.make_batch(args.batch, 0, torch.device('cuda'))
for _ in range(3):
    model.predict(batch, None)
torch.cuda.synchronize()
pflib.check(pflib.load().pf_set_option(b'profile_tag_ops', 1), 'pf_set_option')
pflib.profile(True)
for _ in range(args.steps):
    model.predict(batch, None)
torch.cuda.synchronize()
recs = pflib.profile_results()
pflib.profile(False)
tot = sum(r['ms'] for r in recs) / args.steps
print('total kernel time per step: %.3f ms  (B=%d)' % (tot, args.batch))
recs.sort(key=lambda r: r['label'].split('@')[-1])
for r in recs:
    ms = r['ms'] / args.steps
    k, _, tag = r['label'].partition(' @')
    k = k.replace('void pf::', '').replace('(pf::ConvArgs)', '').replace('pf::', '')
    print('%-44s %-52s %8.1f us %7.1f TF/s %8.0f GB/s' % (tag[:44], k[:52], ms * 1e3, r['flops'] / args.steps / max(ms, 1e-9) / 1e9,
                                                          r['bytes'] / args.steps / max(ms, 1e-9) / 1e6))
