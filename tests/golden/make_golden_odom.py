#!/usr/bin/env python
"""Golden vectors for the predicted-odometry ego chain — BUILD CONTAINER ONLY.

Runs the reference's OWN ``PCTransformDataset.__init__`` (data/datasets/pc_transform_dataset.py:22-186, unmodified,
imported from /root/reference) over a small synthetic Cityscapes-shaped directory written to a temp folder
(timestamp_sequence/*.txt, vehicle_sequence/*_vehicle.json, ``val_3d_info.pkl``) and an ``odometry_val.h5`` content
served by an in-memory stand-in for ``h5py.File`` (h5py is not installed here; the stand-in only hands back the arrays
this script put in, keyed by the reference's own ``'%s/%s/%d/%d'`` names).  What is captured is the reference's
``self.ego_transforms[(city, seq, frame, start_frame)]`` — the three cumulative target_T matrices per sample — for
gap_len 3 (short term) and 9 (mid term), together with every input the chain consumed.

Output: tests/golden/g1_odom.npz.  The .npz holds numbers only (inputs + expected outputs).
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()

ODOM = {}      # name -> array: the content of the odometry h5 file


class _Dataset:
    def __init__(self, a):
        self._a = a

    def __getitem__(self, k):
        return self._a[k]


class _File:
    def __init__(self, path, mode='r'):
        assert mode == 'r' and os.path.basename(path) == 'odometry_val.h5', path

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __getitem__(self, name):
        return _Dataset(ODOM[name])


def install_stubs():
    h5 = types.ModuleType('h5py')
    h5.File = _File
    sys.modules['h5py'] = h5
    # the dataset only needs id2label[id].hasInstances (Cityscapes label table: ids 24..33 are the instance classes)
    cs = types.ModuleType('cityscapesscripts')
    helpers = types.ModuleType('cityscapesscripts.helpers')
    labels = types.ModuleType('cityscapesscripts.helpers.labels')
    L = types.SimpleNamespace
    labels.id2label = {i: L(hasInstances=(24 <= i <= 33)) for i in range(-1, 34)}
    cs.helpers, helpers.labels = helpers, labels
    sys.modules.update({'cityscapesscripts': cs, 'cityscapesscripts.helpers': helpers,
                        'cityscapesscripts.helpers.labels': labels})
    for pkg in ('panoptic_forecasting.data.datasets',):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(_ref_import.REF_ROOT, pkg.replace('.', '/'))]
        sys.modules[pkg] = m


def main():
    import pandas as pd
    install_stubs()
    from panoptic_forecasting.data import data_utils
    if not hasattr(data_utils, 'read_json_file'):
        # reference defect: pc_transform_dataset.py:141 calls data_utils.read_json_file, which the
        # published data_utils.py does not define; the only possible meaning is json.load of the path
        def read_json_file(path):
            with open(path) as f:
                return json.load(f)
        data_utils.read_json_file = read_json_file
    from panoptic_forecasting.data.datasets.pc_transform_dataset import PCTransformDataset
    rng = np.random.Generator(np.random.PCG64(77))
    samples = [('aachen', '000003', 19), ('bonn', '000041', 33), ('ulm', '000007', 25)]
    tmp = tempfile.mkdtemp()
    cs_dir = os.path.join(tmp, 'cityscapes')
    data_dir = os.path.join(tmp, 'meta')
    os.makedirs(data_dir)
    rec = {'samples': np.array(['%s/%s/%d' % s for s in samples])}
    for si, (city, seq, frame) in enumerate(samples):
        os.makedirs(os.path.join(cs_dir, 'timestamp_sequence', 'val', city), exist_ok=True)
        os.makedirs(os.path.join(cs_dir, 'vehicle_sequence', 'val', city), exist_ok=True)
        t = 1.5e18 + np.cumsum(rng.uniform(0.055, 0.062, 30)) * 1e9       # nanoseconds, ~17 Hz with jitter
        speeds = rng.uniform(0.0, 14.0, 30)
        yaws = rng.normal(0.0, 0.05, 30)
        if si == 2:
            yaws[12:16] = 1e-5                                             # the straight-line branch (< 0.000175 rad/s)
        for k, fr in enumerate(range(frame - 19, frame + 11)):
            with open(os.path.join(cs_dir, 'timestamp_sequence', 'val', city, '%s_%s_%06d_timestamp.txt' % (city, seq, fr)), 'w') as f:
                f.write('%d' % int(t[k]))
            with open(os.path.join(cs_dir, 'vehicle_sequence', 'val', city, '%s_%s_%06d_vehicle.json' % (city, seq, fr)), 'w') as f:
                json.dump({'speed': float(speeds[k]), 'yawRate': float(yaws[k])}, f)
        rec['times_ns_%d' % si] = np.array([int(x) for x in t], dtype=np.int64)
        rec['speeds_%d' % si] = speeds
        rec['yaw_rates_%d' % si] = yaws
    pd.DataFrame({'city': [s[0] for s in samples], 'seq': [s[1] for s in samples],
                  'frame': [s[2] for s in samples]}).to_pickle(os.path.join(data_dir, 'val_3d_info.pkl'))
    for gap in (3, 9):
        start = 19 - gap                                                   # last input frame index (input_inds[-1])
        ODOM.clear()
        for si, (city, seq, frame) in enumerate(samples):
            preds = np.stack([rng.uniform(1.0, 13.0, 18), rng.normal(0.0, 0.04, 18)], axis=1).astype(np.float32)
            ODOM['%s/%s/%d/%d' % (city, seq, frame, start)] = preds
            rec['odom_preds_gap%d_%d' % (gap, si)] = preds
        params = {'data': {'data_dir': data_dir, 'cityscapes_dir': cs_dir, 'seg_dir': os.path.join(tmp, 'seg'),
                           'odom_pred_dir': data_dir, 'gap_len': gap}}
        ds = PCTransformDataset('val', params)
        for si, (city, seq, frame) in enumerate(samples):
            rec['target_T_gap%d_%d' % (gap, si)] = np.asarray(ds.ego_transforms[(city, seq, frame, start)], dtype=np.float64)
            assert rec['target_T_gap%d_%d' % (gap, si)].shape == (3, 4, 4)
    np.savez_compressed(os.path.join(HERE, 'g1_odom.npz'), **rec)
    print('wrote g1_odom.npz')


if __name__ == '__main__':
    main()
