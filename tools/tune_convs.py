#!/usr/bin/env python
"""Time every conv kernel shape on every layer of one bg_forecast step (in situ, per-op hipEvent timing) and print
the best shape per layer.   python tools/tune_convs.py [--batch B] [--steps 3] [--json out.json]
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
from panoptic_forecasting_amd.registry import build_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--json', default=None)
ap.add_argument('--emit', default=None, help='append the winners as C table rows to this .inc file')
ap.add_argument('--height', type=int, default=bench.H)
ap.add_argument('--width', type=int, default=bench.W)
ap.add_argument('--fp32', action='store_true', help='the strict-fp32 model (split_f16 = 0): only the fp32-MFMA kernel families compete')
args = ap.parse_args()
bench.H, bench.W = args.height, args.width
L = pflib.load()
pflib.check(L.pf_set_option(b'profile_tag_ops', 1), 'pf_set_option')   # per-op labels in the profile records
pflib.check(L.pf_set_option(b'use_tuned_table', 0), 'pf_set_option')   # 'auto' = the cost model alone
model = build_model(bench.model_params(**({'split_f16': 0} if args.fp32 else {})))
model.load_state_dict(bench.calibrated_state_dict())
batch = bench.make_batch(args.batch, 0, torch.device('cuda'))

CONFIGS = [(0, 0, 0, 0)] + [(1, wm, nt, 0) for wm in (4, 2, 1) for nt in (1, 2, 3, 4) if not (wm == 1 and nt > 2)] + \
          [(2, mh, nt, wk) for mh in (1, 2, 4) for nt in (1, 2) for wk in (2, 4, 8, 16)] + \
          ([(4, nt, wd, 0) for nt in (1, 2, 3, 4) for wd in (0, 1)] if os.environ.get('PF_TUNE_SPLIT', '1') != '0' and not args.fp32 else [])


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
            out[tag] = (r['ms'] / args.steps * 1e3, k.replace('void pf::', '').replace('(pf::ConvArgs)', ''), r['flops'] / args.steps)
    return out


table = {}
for cfg in CONFIGS:
    res = run(cfg)
    for tag, (us, kern, fl) in res.items():
        table.setdefault(tag, {})[cfg] = (us, kern, fl)
L.pf_debug_force_conv(0, 0, 0, 0)

tot_auto = tot_best = 0.0
rows = []
for tag in sorted(table):
    d = table[tag]
    if (0, 0, 0, 0) not in d:      # a launch that exists only under a forced shape (the fused front end splits into its two layers)
        continue
    auto = d[(0, 0, 0, 0)]
    # a forced shape that is not built falls back to the automatic choice: keep an entry only if the kernel that ran
    # (template arguments in its label) is the one that was asked for
    def ran_as_asked(c, kern):
        import re as _re
        nums = [int(x) for x in _re.findall(r'-?\d+', kern.split('<', 1)[1])] if '<' in kern else []
        if c[0] == 0:
            return True
        if c[0] == 1:     # conv_dma_kernel<KS, STRIDE, WM, WK, NT, EPI, RV>
            return 'conv_dma' in kern and len(nums) == 7 and nums[2] == c[1] and nums[4] <= c[2]
        if c[0] == 4 and 'conv_split1' in kern:     # conv_split1_kernel<NT, EPI> (1x1 layers)
            return len(nums) == 2 and nums[0] <= c[1] and c[2] == 0
        if c[0] == 4:     # conv_split_kernel<NT, TW>
            return 'conv_split' in kern and len(nums) == 2 and nums[0] <= c[1] and nums[1] == (64 if c[2] else 32)
        if c[0] == 3:     # conv_valu_kernel<CP, ROWS, KC>
            return 'conv_valu' in kern and len(nums) == 3 and nums[1] == c[1]
        # conv_wave_kernel<KS, MH, NT, WK, EPI>
        return 'conv_wave' in kern and len(nums) == 5 and nums[1] == c[1] and nums[2] <= c[2] and nums[3] == c[3]
    cands = {c: v for c, v in d.items() if ran_as_asked(c, v[1])}
    best = min(cands, key=lambda c: cands[c][0])
    tot_auto += auto[0]
    tot_best += cands[best][0]
    bw = min((c for c in cands if c[0] == 2), key=lambda c: cands[c][0], default=None)
    bd = min((c for c in cands if c[0] == 1), key=lambda c: cands[c][0], default=None)
    rows.append({'tag': tag, 'auto_us': auto[0], 'auto_kernel': auto[1], 'best': list(best), 'best_us': cands[best][0],
                 'best_wave': list(bw) if bw else None, 'best_wave_us': cands[bw][0] if bw else None,
                 'best_dma': list(bd) if bd else None, 'best_dma_us': cands[bd][0] if bd else None,
                 'gflop': auto[2] / 1e9,
                 'all': {'%d,%d,%d,%d' % c: round(v[0], 2) for c, v in sorted(cands.items())}})
    print('%-44s auto %7.1f us (%s) | best %-14s %7.1f us %6.1f TF/s | wave %-14s %7.1f | dma %-12s %7.1f' % (
        tag[:44], auto[0], auto[1][-22:], best, cands[best][0], auto[2] / cands[best][0] / 1e6,
        bw, cands[bw][0] if bw else -1, bd, cands[bd][0] if bd else -1))
print('conv total: auto %.1f us -> best-per-layer %.1f us  (B=%d)' % (tot_auto, tot_best, args.batch))
if args.emit:
    import re
    with open(args.emit, 'a') as f:
        f.write('    // B=%d, %dx%d input: tools/tune_convs.py on MI355X (best-per-layer %.0f us vs cost-model %.0f us)\n'
                % (args.batch, args.height, args.width, tot_best, tot_auto))
        seen = set()
        for r in rows:
            m = re.match(r'\d+[ab]? (\S+) (\d+)->(\d+) (\d+)x(\d+)', r['tag'])
            cin, cout, h, w = int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))
            ks = 1 if ('conv1x1' in r['tag'] or 'finalConv' in r['tag'] or re.match(r'\d+ base\.(5|8|11|14|17) ', r['tag'])) else 3
            key = (ks, cin, cout, h, w, args.batch)
            if key in seen:
                continue
            seen.add(key)
            b = r['best']
            if b[0] == 0 or r['best_us'] > 0.97 * r['auto_us']:     # within noise of the cost model: keep the model
                continue
            f.write('    {%d, %d, %d, %d, %d, %d, {%d, %d, %d, %d}},   // %s: %.1f -> %.1f us\n'
                    % (ks, cin, cout, h, w, args.batch, b[0], b[1], b[2], b[3], m.group(1), r['auto_us'], r['best_us']))
if args.json:
    with open(args.json, 'w') as f:
        json.dump({'batch': args.batch, 'h': args.height, 'w': args.width, 'rows': rows}, f, indent=1)
