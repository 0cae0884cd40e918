#!/usr/bin/env python
"""by_batch_pipelined of bench.py at several pipeline depths (model replicas / streams): python tools/pipeline_depth.py [2 3 4]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

torch.cuda.set_device(0)
sd = bench.calibrated_state_dict()
for d in [int(v) for v in sys.argv[1:]] or [2, 3, 4]:
    r = bench.pipelined_leg(sd, torch.device('cuda', 0), 'short', depth=d)
    print('depth %d: ' % d + ', '.join('B%s %.0f frames/s' % (k, v['value']) for k, v in r.items()))
