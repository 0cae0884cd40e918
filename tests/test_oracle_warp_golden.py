"""The C oracle (oracle/warp_splat_ref.c) against outputs of the reference itself (fixtures g1_*).

Bit-exact: seg, depth (compared as u32 bit patterns) and the int64 result2d.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import warp_splat as oracle

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'g1_*x*.npz')))


def load_inputs(z, img=False):
    return {'intrinsics': torch.from_numpy(z['K']), 'extrinsics': torch.from_numpy(z['E']),
            'target_T': torch.from_numpy(z['T']), 'depth': torch.from_numpy(z['depth']),
            'depth_mask': torch.from_numpy(z['mask']),
            'seg': torch.from_numpy(z['img'] if img else z['seg'])}


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f) for f in FILES])
@pytest.mark.parametrize('ind', [None, 0, 1, 2])
@pytest.mark.parametrize('is_img', [False, True])
def test_oracle_matches_reference(path, ind, is_img):
    z = np.load(path)
    out = oracle.predict(load_inputs(z, is_img), only_this_ind=ind, is_img=is_img)
    tag = '%s_%d' % ('all' if ind is None else str(ind), int(is_img))
    assert np.array_equal(out['seg'].numpy(), z['seg_' + tag])
    assert np.array_equal(out['depth'].numpy().view(np.uint32), z['depth_bits_' + tag])
    if not is_img:
        assert np.array_equal(out['result2d'].numpy(), z['result2d_' + tag].astype(np.int64))


def test_fixture_inverses_match_host_lapack():
    """The product computes K^-1/E^-1 with torch.inverse on the host; the fixtures captured the same bits."""
    for path in FILES:
        z = np.load(path)
        assert np.array_equal(torch.inverse(torch.from_numpy(z['K'])).numpy(), z['Kinv'])
        assert np.array_equal(torch.inverse(torch.from_numpy(z['E'])).numpy(), z['Einv'])
