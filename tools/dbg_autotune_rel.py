#!/usr/bin/env python
"""How far do the gradients of a step with measured conv shapes (training.autotune) sit from the cost model's?  Prints the relative
distance test_measured_conv_shapes_change_rounding_only bounds by 1e-5, N times (the picks depend on timings)."""
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
for kv in filter(None, os.environ.get('PF_OPTS', '').split(',')):
    k, v = kv.split('=')
    pflib.check(L.pf_set_option(k.encode(), int(v)), 'pf_set_option')
pflib.check(L.pf_set_option(b'use_tuned_table', 0), 'pf_set_option')
ref = BGTrainer(T._params())
ref.load_state_dict(T._sd())
want = ref.forward_backward(a_in, a_lab, update_running_stats=False)
g_ref = ref.grad.clone()
L.pf_set_option(b'use_tuned_table', 1)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    tr = BGTrainer(T._params(autotune=True))
    tr.load_state_dict(T._sd())
    got = tr.forward_backward(a_in, a_lab, update_running_stats=False)
    g1 = tr.grad.clone()
    tr.forward_backward(a_in, a_lab, update_running_stats=False)
    print('rel %.3e  loss rel %.3e  second step identical: %s' % (T._rel(g1, g_ref), abs(float(got['loss']) - float(want['loss'])) / abs(float(want['loss'])), torch.equal(tr.grad, g1)))
