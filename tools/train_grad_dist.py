#!/usr/bin/env python
"""Gradient distances to float64 (rel. L2) of the HIP training step with and without blocked summation, next to ATen fp32's.
    python tools/train_grad_dist.py [H W]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_train as t  # noqa: E402
from oracle import hardnet_ref  # noqa: E402
from panoptic_forecasting_amd import lib as pflib, synth  # noqa: E402
from panoptic_forecasting_amd.bg_train import BGTrainer  # noqa: E402

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 256)
sd = t._sd()
inputs = synth.make_bg_inputs(b=2, h=h, w=w, seed=21)
labels = {'seg': t._labels(2, h, w, 5)}
r32 = hardnet_ref.bg_train_step({k: v.clone() for k, v in sd.items()}, inputs, labels, clip_grad_norm=None, apply_update=False)
sd64 = {k: (v.double() if v.dtype == torch.float32 else v.clone()) for k, v in sd.items()}
in64 = dict(inputs)
in64['depth'] = inputs['depth'].double()
r64 = hardnet_ref.bg_train_step(sd64, in64, labels, clip_grad_norm=None, apply_update=False)
aten = {k: t._rel(r32['grads'][k], g) for k, g in r64['grads'].items()}
L = pflib.load()
res = {}
for blocked in (0, 1):
    pflib.check(L.pf_set_option(b'train_blocked_sum', blocked), 'opt')
    tr = BGTrainer(t._params())
    tr.load_state_dict(sd)
    tr.forward_backward(t._cuda(inputs), t._cuda(labels))
    got = tr.named_grads()
    res[blocked] = {k: t._rel(got[k].cpu(), g) for k, g in r64['grads'].items()}
med = lambda d: sorted(d.values())[len(d) // 2]
print('%dx%d median: ATen %.3e  HIP one-chain %.3e  HIP blocked %.3e   max: %.3e / %.3e / %.3e' % (
    h, w, med(aten), med(res[0]), med(res[1]), max(aten.values()), max(res[0].values()), max(res[1].values())))
for k in list(aten)[:6] + list(aten)[-4:]:
    print('  %-48s ATen %.3e  one-chain %.3e  blocked %.3e' % (k, aten[k], res[0][k], res[1][k]))
