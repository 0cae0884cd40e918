"""Diagnostic (GPU box): gradients of the device training step against the oracle in float64 — per parameter, per input
range of the last block's weights, and per activation gradient of the layers the oracle exposes."""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hardnet_ref
from panoptic_forecasting_amd import synth, lib as _lib
from panoptic_forecasting_amd.bg_train import BGTrainer
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 128)
with open(os.path.join(G, 'calib_seed1234.json')) as f:
    sd = synth.make_state_dict(seed=1234, calib=json.load(f))
params = {'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])]},
          'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True}, 'training': {}}
inputs = synth.make_bg_inputs(b=2, h=h, w=w, seed=21)
g = torch.Generator().manual_seed(4005)
lab = torch.randint(0, 12, (2, h // 8, w // 8), generator=g); lab[lab == 11] = 255
labels = {'seg': torch.nn.functional.interpolate(lab[:, None].float(), size=(h, w), mode='nearest')[:, 0].long()}
tr = BGTrainer(params); tr.load_state_dict(sd)
out = tr.forward_backward({k: v.cuda() for k, v in inputs.items()}, {k: v.cuda() for k, v in labels.items()})
torch.cuda.synchronize()

acts = {}
orig = hardnet_ref._conv_bn_relu
def tap(sd_, p, x, stride=1):
    y = orig(sd_, p, x, stride)
    y.retain_grad()
    acts[p[len('model.'):]] = y
    return y
hardnet_ref._conv_bn_relu = tap
sd64 = {k: (v.double() if v.dtype == torch.float32 else v.clone()) for k, v in sd.items()}
in64 = dict(inputs); in64['depth'] = inputs['depth'].double()
ref = hardnet_ref.bg_train_step(sd64, in64, labels, clip_grad_norm=None, apply_update=False)
hardnet_ref._conv_bn_relu = orig
r32 = hardnet_ref.bg_train_step({k: v.clone() for k, v in sd.items()}, inputs, labels, clip_grad_norm=None, apply_update=False)
print('loss hip %.7f  f64 %.7f  f32 %.7f' % (float(out['loss']), float(ref['loss']), float(r32['loss'])))
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
got = tr.named_grads()
print('%-52s %9s %9s' % ('parameter', 'hip/f64', 'f32/f64'))
for key, off, shape, trainable in tr.layout:
    if trainable:
        print('%-52s %.2e  %.2e' % (key, rel(got[key].cpu(), ref['grads'][key]), rel(r32['grads'][key], ref['grads'][key])))
# activation + gradient tensors of single-writer tensors and block slices
L = _lib.load()
spec = tr.spec
def view(name, grad):
    off, c, th, tw = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(L.pf_train_tensor_view(tr._t, name.encode(), int(grad), 2, h, w, h, w, ctypes.byref(off), ctypes.byref(c), ctypes.byref(th), ctypes.byref(tw)), 'view')
    n = 2 * c.value * th.value * tw.value
    return tr._ws[off.value:off.value + 4 * n].view(torch.float32).view(2, c.value, th.value, tw.value).cpu()
print('%-40s %9s %9s' % ('tensor (op output)', 'act', 'grad'))
for op in spec.conv_ops():
    if op.name not in acts:
        continue
    name = spec.tensors[op.dst].name
    a = view(name, 0)[:, op.dst_choff:op.dst_choff + op.cout]
    ga = view(name, 1)[:, op.dst_choff:op.dst_choff + op.cout]
    t = acts[op.name]
    print('%-40s %.2e  %.2e' % (op.name, rel(a, t.detach()), rel(ga, t.grad) if t.grad is not None else -1))
