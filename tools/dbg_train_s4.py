#!/usr/bin/env python
"""Training step with the forward convolutions on the packed-pair kernels (option train_forward_s4) against the default step:
loss and per-parameter gradient distance on the test fixture's batch, and the path statistics."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_train as T  # noqa: E402
from panoptic_forecasting_amd import lib as pflib  # noqa: E402
from panoptic_forecasting_amd.bg_train import BGTrainer  # noqa: E402

z, batches = T._fixture()
a_in, a_lab = (T._cuda(d) for d in batches[0])
L = pflib.load()
out = {}
for mode in (0, 1):
    pflib.check(L.pf_set_option(b'train_forward_s4', mode), 'pf_set_option')
    tr = BGTrainer(T._params())
    tr.load_state_dict(T._sd())
    r = tr.forward_backward(a_in, a_lab, update_running_stats=False)
    out[mode] = (float(r['loss']), {k: v.clone() for k, v in tr.named_grads().items()}, tr.grad.clone(), tr.path_stats())
L.pf_set_option(b'train_forward_s4', 0)
print('loss', out[0][0], out[1][0], abs(out[0][0] - out[1][0]) / abs(out[0][0]))
print('whole-gradient rel', T._rel(out[1][2], out[0][2]))
worst = sorted(((T._rel(out[1][1][k], out[0][1][k]), k) for k in out[0][1]), reverse=True)[:5]
print('worst parameters', worst)
print(out[1][3])
