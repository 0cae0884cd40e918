#!/usr/bin/env python
"""Diagnostic: how far is the TRAINING-mode forward pass (batch-statistics BatchNorm) from float64 - for the HIP step and for
torch fp32 on the CPU (the reference's arithmetic) - per block output, relative L2.  The gradient distances of
tests/test_gpu_train.py follow the forward's (ReLU masks flip where a pre-activation is closer to zero than the round-off).

    python tools/train_fwd_error.py [--size 800] [--batch 2] [--table-batch 8]
"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import hardnet_ref  # noqa: E402
from panoptic_forecasting_amd import lib as pflib, synth  # noqa: E402
from panoptic_forecasting_amd.bg_train import BGTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=800)
ap.add_argument('--width', type=int, default=0)
ap.add_argument('--batch', type=int, default=2)
ap.add_argument('--table-batch', type=int, default=8)
ap.add_argument('--use-table', type=int, default=1)
ap.add_argument('--blocked', type=int, default=1, help='option train_blocked_sum: per-round partial sums in the 3x3 convolutions (round 6)')
a = ap.parse_args()
for kv in filter(None, os.environ.get('PF_OPTS', '').split(',')):    # e.g. PF_OPTS=train_forward_s4=1
    k, v = kv.split('=')
    pflib.check(pflib.load().pf_set_option(k.encode(), int(v)), 'pf_set_option')
h, w, b = a.size, a.width or a.size, a.batch
with open(os.path.join(ROOT, 'tests', 'golden', 'calib_seed1234.json')) as f:
    sd = synth.make_state_dict(seed=1234, calib=json.load(f))
inputs = synth.make_bg_inputs(b=b, h=h, w=w, seed=31)
g = torch.Generator().manual_seed(4007)
lab = torch.randint(0, 12, (b, h // 8, w // 8), generator=g)
lab[lab == 11] = 255
labels = {'seg': torch.nn.functional.interpolate(lab[:, None].float(), size=(h, w), mode='nearest')[:, 0].long()}


def cpu_taps(dtype):
    s = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    x = hardnet_ref.bg_inputs_to_tensor(s, inputs['seg'], inputs['depth'].to(dtype), inputs['depth_mask'])
    taps = {}
    hardnet_ref._TRAINING = True
    try:
        with torch.no_grad():
            _, orig = hardnet_ref.hardnet_forward(s, x.to(dtype), (h, w), taps=taps)
    finally:
        hardnet_ref._TRAINING = False
    taps['finalConv'] = orig
    return taps


t64, t32 = cpu_taps(torch.float64), cpu_taps(torch.float32)
L = pflib.load()
pflib.check(L.pf_set_option(b'train_table_batch', a.table_batch), 'opt')
pflib.check(L.pf_set_option(b'use_tuned_table', a.use_table), 'opt')
pflib.check(L.pf_set_option(b'train_blocked_sum', a.blocked), 'opt')
params = {'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])]},
          'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True},
          'training': {'lr': 2e-3, 'mom': 0.9, 'wd': 1e-4, 'clip_grad_norm': 5.0}}
tr = BGTrainer(params)
tr.load_state_dict(sd)
tr.forward_backward({k: v.cuda() for k, v in inputs.items()}, {k: v.cuda() for k, v in labels.items()})
torch.cuda.synchronize()
print('path stats', tr.path_stats())
names = {'base.0': 'base.0', 'base.1': 'base.1', 'base.2': 'base.2', 'base.3': 'base.3', 'base.4': 'base.4.out', 'base.7': 'base.7.out',
         'base.10': 'base.10.out', 'base.13': 'base.13.out', 'base.16': 'base.16.out', 'denseBlocksUp.0': 'denseBlocksUp.0.out',
         'denseBlocksUp.1': 'denseBlocksUp.1.out', 'denseBlocksUp.2': 'denseBlocksUp.2.out', 'denseBlocksUp.3': 'denseBlocksUp.3.out',
         'finalConv': 'finalConv'}


def rel(x, y):
    return float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30))


for tap, tname in names.items():
    off, c, th, tw = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    pflib.check(L.pf_train_tensor_view(tr._t, tname.encode(), 0, b, h, w, h, w, ctypes.byref(off), ctypes.byref(c), ctypes.byref(th),
                                       ctypes.byref(tw)), 'view')
    n = b * c.value * th.value * tw.value
    got = tr._ws[off.value:off.value + 4 * n].view(torch.float32).view(b, c.value, th.value, tw.value).cpu()
    ref = t64[tap]
    print('%-18s %4d ch %4dx%-4d  HIP vs f64 %.3e   ATen f32 vs f64 %.3e   ratio %.2f' % (
        tap, c.value, th.value, tw.value, rel(got, ref), rel(t32[tap], ref), rel(got, ref) / max(rel(t32[tap], ref), 1e-30)))
