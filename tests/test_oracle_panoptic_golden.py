"""The C oracle of the fg -> panoptic merge against outputs of the reference's own predict_panoptic /
predict_semantics / export functions (fixtures g5_*.npz, generator tests/golden/make_golden_fg.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import panoptic as op


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name))
    counts = [int(c) for c in z['counts']]
    split = lambda a: list(torch.from_numpy(np.ascontiguousarray(a)).split(counts))
    case = {'masks': split(z['mask_prob']), 'boxes': split(z['boxes']), 'depths': split(z['depths']),
            'classes': split(z['classes']), 'background': torch.from_numpy(z['background']).long(),
            'seg': torch.from_numpy(z['seg']).long()}
    if 'bg_depth' in z:
        h, w = z['seg'].shape[1:]
        case['bg_depth'] = torch.from_numpy(z['bg_depth'].astype(np.float32))
        case['bg_depth_mask'] = torch.from_numpy(np.unpackbits(z['bg_depth_mask'])[:len(counts) * h * w]
                                                 .reshape(len(counts), h, w).astype(bool))
    return case


CASES = [('g5_panoptic.npz', dict(panoptic=True)),
         ('g5_panoptic_z.npz', dict(panoptic=True)),
         ('g5_panoptic_ulbr.npz', dict(panoptic=True, use_depth_sorting=False, use_bbox_ulbr=True)),
         ('g5_semantic.npz', dict(panoptic=False))]


@pytest.mark.parametrize('name,kw', CASES)
def test_oracle_merge_equals_reference(name, kw):
    c = load_case(name)
    h, w = c['seg'].shape[1:]
    got = op.merge(c['masks'], c['boxes'], c['depths'], c['classes'], h, w, background=c['background'],
                   background_depth=c.get('bg_depth'), background_depth_mask=c.get('bg_depth_mask'), **kw)
    assert torch.equal(got, c['seg'])
    assert (c['seg'] > 100).any() or not kw['panoptic']      # instances were actually pasted


def test_oracle_encode_equals_reference():
    z = np.load(os.path.join(GOLDEN, 'g5_encode.npz'))
    rgb, ids, present = op.encode(torch.from_numpy(z['seg']).long(), convert=True)
    assert np.array_equal(ids, z['converted'])
    assert np.array_equal(rgb, z['rgb'])
    assert present == [0] + [int(i) for i in z['seg_ids']] if 0 in present else present == [int(i) for i in z['seg_ids']]
    cats = [i // 1000 if i > 100 else i for i in present if i != 0]
    assert cats == [int(i) for i in z['cat_ids']]
