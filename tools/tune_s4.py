#!/usr/bin/env python
"""Per-layer time of every conv_s4 shape (pf_debug_force_conv(5, nt, wide)) next to the automatic choice, in situ.
    python tools/tune_s4.py [--batch B] [--steps 3]
"""
import argparse
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panoptic_forecasting_amd import lib as pflib  # noqa: E402
from panoptic_forecasting_amd.registry import build_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--margin', type=float, default=0.03, help='a challenger shape replaces the shape the current table runs only if it is this much faster (run-to-run noise is 1-3 %%)')
ap.add_argument('--size', default='', help='HxW instead of 1024x2048: rows for a second image size (conv_select.cpp prefers rows measured at exactly the size asked for)')
ap.add_argument('--emit', default=None, help='append the conv_s4 winners as C table rows to this .inc file')
args = ap.parse_args()
L = pflib.load()
pflib.check(L.pf_set_option(b'profile_tag_ops', 1), 'pf_set_option')   # per-op labels in the profile records
if args.size:
    bench.H, bench.W = [int(v) for v in args.size.lower().split('x')]
model = build_model(bench.model_params(final_h=bench.H, final_w=bench.W))
model.load_state_dict(bench.calibrated_state_dict())
batch = bench.make_batch(args.batch, 0, torch.device('cuda'))
CONFIGS = [(0, 0, 0, 0)] + [(5, nt, wd, 0) for nt in (1, 2, 3, 4) for wd in (0, 1)] + [(5, 1, 2, 0), (5, 2, 2, 0), (5, 1, 3, 0), (5, 2, 3, 0), (5, 1, 4, 0), (5, 2, 4, 0)]   # wd 2: 16x32 tiles, 8 waves; wd 3 / 4: 8x32 tiles, K split over two / four wave groups


def run(cfg):
    L.pf_debug_force_conv(*cfg)
    for _ in range(2):
        model.predict(batch, None)
    torch.cuda.synchronize()
    pflib.profile(True)
    for _ in range(args.steps):
        model.predict(batch, None)
    torch.cuda.synchronize()
    recs = pflib.profile_results()
    pflib.profile(False)
    out = {}
    for r in recs:
        k, _, tag = r['label'].partition(' @')
        if tag and 'conv' in k:
            out[tag] = (r['ms'] / args.steps * 1e3, k.replace('void pf::', '').replace('(pf::ConvArgs)', ''))
    return out


table = {}
for cfg in CONFIGS:
    for tag, v in run(cfg).items():
        table.setdefault(tag, {})[cfg] = v
L.pf_debug_force_conv(0, 0, 0, 0)
tot_auto = tot_best = 0.0
emit_rows = []
for tag in sorted(table):
    d = table[tag]
    auto = d.get((0, 0, 0, 0), (0.0, '?'))
    s4 = {}
    for c, (us, kern) in d.items():
        if c[0] == 5 and 'conv_s4' in kern:
            nums = [int(x) for x in re.findall(r'-?\d+', kern.split('<', 1)[1].split('>')[0])]
            s4.setdefault(kern.split('>')[0] + '>', []).append(us)     # the shape that actually ran
    best = min(s4.items(), key=lambda kv: min(kv[1])) if s4 else ('-', [auto[0]])
    # hysteresis: the shape the current table already runs stays unless a challenger beats it by the margin
    inc = auto[1].split('>')[0] + '>' if 'conv_s4' in auto[1] else None
    if s4 and inc in s4 and min(best[1]) > (1.0 - args.margin) * min(min(s4[inc]), auto[0]):
        best = (inc, [min(min(s4[inc]), auto[0])])
    tot_auto += auto[0]
    tot_best += min(min(best[1]), auto[0])
    if s4:
        emit_rows.append((tag, auto, best[0], min(best[1])))
    print('%-44s auto %7.1f us %-34s | ' % (tag[:44], auto[0], auto[1][:34]) +
          '  '.join('%s %.1f' % (k.replace('conv_s4_', '').replace('kernel', 'k'), min(v)) for k, v in sorted(s4.items())))
print('conv total: auto %.1f us -> best of auto / s4 per layer %.1f us  (B=%d)' % (tot_auto, tot_best, args.batch))

if args.emit:
    with open(args.emit, 'a') as f:
        f.write('    // B=%d%s: tools/tune_s4.py on MI355X, every eligible layer on conv_s4 (auto %.0f us -> best shapes %.0f us)\n'
                % (args.batch, (' at %dx%d' % (bench.H, bench.W)) if args.size else '', tot_auto, tot_best))
        seen = set()
        for tag, auto, kern, us in emit_rows:
            m = re.match(r'\d+[ab]? (\S+) (\d+)->(\d+) (\d+)x(\d+)', tag)
            cin, cout, h, w = int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))
            nums = [int(x) for x in re.findall(r'-?\d+', kern.split('<', 1)[1])]
            ks = 1 if '1x1' in kern else 3
            key = (ks, cin, cout, h, w)
            if key in seen:
                continue
            seen.add(key)
            nt, wide = nums[0], (4 if ks == 3 and len(nums) > 3 and nums[3] == 4 else 3 if ks == 3 and len(nums) > 3 and nums[3] == 2 else 2 if ks == 3 and len(nums) > 2 and nums[2] == 16 else 1 if ks == 3 and nums[1] == 64 else 0)
            # keep = 0: the non-S4 kernel the table picks was faster in situ (mixed formats are decided by the plan's fixpoint)
            keep = 1 if ('conv_s4' in auto[1] or us < 0.98 * auto[0]) else 0
            f.write('    {%d, %d, %d, %d, %d, %d, {%d, %d, %d, 0}},   // %s: auto %.1f (%s) -> %.1f us\n'
                    % (ks, cin, cout, h, w, args.batch, 5 if keep else 0, nt, wide, m.group(1), auto[0],
                       auto[1].split('<')[0].replace('conv_', '').replace('_kernel', ''), us))
