import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_train as T
from panoptic_forecasting_amd import synth
from panoptic_forecasting_amd.bg_train import BGTrainer
h, w = 128, 256
sd = T._sd()
inputs = synth.make_bg_inputs(b=2, h=h, w=w, seed=21)
labels = {'seg': T._labels(2, h, w, 5)}
from panoptic_forecasting_amd import lib as pflib
for kv in filter(None, os.environ.get('PF_OPTS', '').split(',')):
    k, v = kv.split('=')
    pflib.check(pflib.load().pf_set_option(k.encode(), int(v)), 'pf_set_option')
tr = BGTrainer(T._params()); tr.load_state_dict(sd)
out = tr.forward_backward(T._cuda(inputs), T._cuda(labels))
g64, ref, bars, sd64 = T._oracle_grads(sd, inputs, labels, '128x256')
got = tr.named_grads()
dist = {k: T._rel(got[k].cpu(), g) for k, g in g64.items()}
aten = {k: T._rel(ref['grads'][k], g) for k, g in g64.items()}
top = sorted(dist.items(), key=lambda kv: -kv[1])[:4]
print(os.environ.get('PF_S4_ACT_SCALE'), os.environ.get('PF_OPTS'), 'max', top, 'median', sorted(dist.values())[len(dist)//2], 'aten stem', aten['model.base.0.conv.weight'], 'aten max', max(aten.values()))
