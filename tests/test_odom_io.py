"""f1: odometry files -> target_T, pinned on outputs of the reference's own PCTransformDataset constructor
(tests/golden/make_golden_odom.py ran it in the build container; g1_odom.npz holds its inputs and outputs)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _fixture():
    return np.load(os.path.join(GOLDEN, 'g1_odom.npz'))


def _write_tree(z, root):
    """Re-create the directory the fixture was generated from."""
    for si, name in enumerate(z['samples']):
        city, seq, frame = str(name).split('/')
        frame = int(frame)
        for sub in ('timestamp_sequence', 'vehicle_sequence'):
            os.makedirs(os.path.join(root, sub, 'val', city), exist_ok=True)
        for k, fr in enumerate(range(frame - 19, frame + 11)):
            stem = '%s_%s_%06d' % (city, seq, fr)
            with open(os.path.join(root, 'timestamp_sequence', 'val', city, stem + '_timestamp.txt'), 'w') as f:
                f.write('%d' % int(z['times_ns_%d' % si][k]))
            with open(os.path.join(root, 'vehicle_sequence', 'val', city, stem + '_vehicle.json'), 'w') as f:
                json.dump({'speed': float(z['speeds_%d' % si][k]), 'yawRate': float(z['yaw_rates_%d' % si][k])}, f)


@pytest.mark.parametrize('gap', [3, 9])
def test_predicted_chain_from_files_equals_reference_dataset(tmp_path, gap):
    from panoptic_forecasting_amd import odom_io
    z = _fixture()
    root = str(tmp_path / 'cityscapes')
    _write_tree(z, root)
    entries = {}
    for si, name in enumerate(z['samples']):
        city, seq, frame = str(name).split('/')
        entries[(city, seq, int(frame), 19 - gap)] = z['odom_preds_gap%d_%d' % (gap, si)]
    npz = str(tmp_path / 'odometry_val.npz')
    odom_io.write_npz(npz, entries)
    with odom_io.OdometryFile(npz) as odom:
        for si, name in enumerate(z['samples']):
            city, seq, frame = str(name).split('/')
            T = odom_io.snippet_target_T(root, 'val', city, seq, int(frame), gap, odom=odom)
            ref = z['target_T_gap%d_%d' % (gap, si)]
            assert T.shape == (3, 4, 4)
            # same float64 formulas in the same order; what the device path sees is the float32 cast, which must be equal
            assert np.allclose(T, ref, rtol=0, atol=1e-13)
            assert np.array_equal(T.astype(np.float32), ref.astype(np.float32))
        with pytest.raises(KeyError):
            odom.rows('nowhere', '000000', 1, 1)


def test_h5_path_without_h5py_points_at_the_npz_twin(tmp_path):
    from panoptic_forecasting_amd import odom_io
    try:
        import h5py  # noqa: F401
        pytest.skip('h5py is installed here')
    except ImportError:
        pass
    p = str(tmp_path / 'odometry_val.h5')
    open(p, 'wb').close()
    with pytest.raises(ImportError):
        odom_io.OdometryFile(p)
    odom_io.write_npz(str(tmp_path / 'odometry_val.npz'), {('a', 'b', 19, 16): np.ones((18, 2))})
    assert odom_io.OdometryFile(p).rows('a', 'b', 19, 16).shape == (18, 2)      # falls back to the twin


def test_key_and_indices():
    from panoptic_forecasting_amd import odom_io
    assert odom_io.odom_key('ulm', '000007', 25, 10) == 'ulm/000007/25/10'
    assert list(odom_io.input_indices(3)) == [10, 13, 16] and list(odom_io.input_indices(9)) == [4, 7, 10]
