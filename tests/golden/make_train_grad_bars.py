#!/usr/bin/env python
"""tests/golden/train_grad_bars.json from a measurement on the GPU box.

    (on MI355X)  python -m pytest tests/test_gpu_train.py -m gpu -q -k forward_backward_vs_oracle     # writes gpurun_out/r03_train_grad_dist.json
    (here)       python tests/golden/make_train_grad_bars.py [gpurun_out/r03_train_grad_dist.json]

Per tensor of the bg network: bar = 1.5 x the measured rel. L2 distance between the HIP training step's gradient and the float64
oracle's, floor 1e-4.  The training step is bit-reproducible, so the measured distances only move when a training kernel changes
its summation order; the measurement itself is committed as profiles/r03_train_grad_dist.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'r03_train_grad_dist.json')
g = json.load(open(src))
bars = {size: {k: max(1.5 * d, 1e-4) for k, d in v['hip_vs_f64'].items()} for size, v in g.items()}
json.dump(g, open(os.path.join(ROOT, 'profiles', 'r03_train_grad_dist.json'), 'w'), indent=0, sort_keys=True)
json.dump({'note': '1.5 x the rel. L2 distance HIP <-> float64 measured on MI355X for each tensor (profiles/r03_train_grad_dist.json), floor 1e-4',
           'bars': bars}, open(os.path.join(ROOT, 'tests', 'golden', 'train_grad_bars.json'), 'w'), indent=0, sort_keys=True)
for size, v in g.items():
    print(size, 'HIP max %.3e  fp32 ATen max %.3e' % (v['max_hip'], v['max_aten']))
