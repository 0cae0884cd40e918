"""The torch oracle (oracle/hardnet_ref.py) against outputs of the reference BGModel (fixtures g3_*),
and the build's architecture table / synthetic checkpoint against the reference modules (g4)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import hardnet_ref
from panoptic_forecasting_amd import hardnet_arch as arch
from panoptic_forecasting_amd import synth

G = os.path.join(os.path.dirname(__file__), 'golden')
LOGIT_TOL = 1e-4   # same ATen kernels as the reference; only BN/conv call order could differ


def calibrated_sd():
    with open(os.path.join(G, 'calib_seed1234.json')) as f:
        return synth.make_state_dict(seed=1234, calib=json.load(f))


@pytest.mark.parametrize('size', ['64x128', '96x160'])
def test_oracle_matches_reference_bg(size):
    z = np.load(os.path.join(G, 'g3_%s.npz' % size))
    sd = calibrated_sd()
    inputs = {'seg': torch.from_numpy(z['seg_in']).long(), 'depth': torch.from_numpy(z['depth']),
              'depth_mask': torch.from_numpy(z['mask'])}
    h, w = inputs['seg'].shape[-2:]
    out = hardnet_ref.bg_predict(sd, inputs, final_size=(h, w))
    assert out['orig_size_logits'].shape == z['orig_size_logits'].shape
    err = np.abs(out['orig_size_logits'].numpy() - z['orig_size_logits']).max()
    assert err <= LOGIT_TOL, err
    assert np.abs(out['logits'].numpy() - z['logits']).max() <= LOGIT_TOL
    agree = (out['seg'].numpy() == z['seg']).mean()
    assert agree >= 0.9999, agree


def test_arch_table_matches_reference_modules():
    with open(os.path.join(G, 'g4_arch.json')) as f:
        g4 = json.load(f)
    spec = arch.Spec(36, 11)
    convs = spec.conv_ops()
    assert len(convs) == len(g4['convs']) == 70
    dims = {spec.input_tensor: (1024, 2048)}
    by_name = {}
    for o in spec.ops:
        ih, iw = dims[o.srcs[0].tensor]
        if o.kind in (arch.OP_STEM, arch.OP_CONV):
            oh = (ih + 2 * (o.k // 2) - o.k) // o.stride + 1
            ow = (iw + 2 * (o.k // 2) - o.k) // o.stride + 1
            by_name[o.name] = (o.cin, o.cout, o.k, o.stride, oh, ow)
        elif o.kind == arch.OP_POOL:
            oh, ow = ih // 2, iw // 2
        elif o.kind == arch.OP_UPSAMPLE:
            oh, ow = dims[o.srcs[1].tensor]
        else:
            oh, ow = ih, iw
        dims[o.dst] = (oh, ow)
    for r in g4['convs']:
        name = r['name'][:-len('.conv')] if r['name'].endswith('.conv') else r['name']
        assert by_name[name] == (r['cin'], r['cout'], r['k'], r['stride'], r['oh'], r['ow']), name
    assert abs(spec.flops(1024, 2048) / 1e9 - 75.32) < 0.01


def test_synthetic_checkpoint_has_reference_keys():
    with open(os.path.join(G, 'g4_arch.json')) as f:
        g4 = json.load(f)
    sd = synth.make_state_dict()
    assert sorted(sd.keys()) == g4['state_dict_keys']
    for k, v in sd.items():
        assert list(v.shape) == g4['shapes'][k], k
