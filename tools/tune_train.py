#!/usr/bin/env python3
"""Measure the workgroup shapes of the training step's convolutions (pf_train_autotune) at one batch configuration and write
them as rows of csrc/train_tuned.inc.

    python tools/tune_train.py [--batch 8] [--size 800] [--runs 3] [--emit panoptic-forecasting_amd/csrc/train_tuned.inc]

Every run builds a fresh trainer, lets one step measure every geometry and reads the choices back; a geometry gets a table row
only when a forced shape won (at least 3 % faster than the cost model's) in the majority of the runs, with the shape most runs picked.
"""
import argparse
import collections
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panoptic_forecasting_amd import lib as pflib, synth  # noqa: E402
from panoptic_forecasting_amd.bg_train import BGTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--size', type=int, default=800)
    ap.add_argument('--runs', type=int, default=3)
    ap.add_argument('--emit', default='')
    a = ap.parse_args()
    L = pflib.load()
    votes = collections.defaultdict(collections.Counter)
    for run in range(a.runs):
        params = {'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])]},
                  'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True},
                  'training': {'lr': 2e-3, 'mom': 0.9, 'wd': 1e-4, 'clip_grad_norm': 5.0, 'autotune': True}}
        tr = BGTrainer(params)
        tr.load_state_dict(synth.make_state_dict(seed=1234))
        inp = {k: v.cuda() for k, v in synth.make_bg_inputs(b=a.batch, h=a.size, w=a.size, seed=1).items()}
        inp['seg'] = inp['seg'].to(torch.uint8)
        lab = {'seg': torch.randint(0, 11, (a.batch, a.size, a.size), dtype=torch.uint8, device='cuda')}
        tr.train_step(inp, lab)
        torch.cuda.synchronize()
        n = ctypes.c_int()
        pflib.check(L.pf_train_tuned_shapes(tr._t, None, 0, ctypes.byref(n)), 'pf_train_tuned_shapes')
        rows = (ctypes.c_int * (10 * n.value))()
        pflib.check(L.pf_train_tuned_shapes(tr._t, rows, n.value, ctypes.byref(n)), 'pf_train_tuned_shapes')
        for i in range(n.value):
            r = list(rows[i * 10:i * 10 + 10])
            votes[tuple(r[:8])][tuple(r[8:])] += 1
        del tr
    out = []
    for key in sorted(votes):
        (pick, cnt), = votes[key].most_common(1)
        if pick != (0, 0) and cnt * 2 > a.runs:
            out.append('    {{%s}, %d, %d},' % (', '.join(str(v) for v in key), pick[0], pick[1]))
    print('%d geometries, %d rows' % (len(votes), len(out)))
    text = ('// tools/tune_train.py --batch %d --size %d --runs %d on MI355X: {ks, stride, Cin, Cout, Hin, Win, B, accumulate}, pixel waves, cout tiles\n'
            % (a.batch, a.size, a.runs)) + '\n'.join(out) + '\n'
    if a.emit:
        with open(a.emit, 'w') as f:
            f.write(text)
    else:
        print(text)


if __name__ == '__main__':
    main()
