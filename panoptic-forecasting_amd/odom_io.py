"""File side of the ego chain: Cityscapes timestamps / vehicle JSON + the odometry forecaster's export.

What the reference reads before ``PCTransformModel.predict`` is entered (``data/datasets/pc_transform_dataset.py:103-186``):

  ``timestamp_sequence/{split}/{city}/{city}_{seq}_{fr:06d}_timestamp.txt``   one integer, nanoseconds
  ``vehicle_sequence/{split}/{city}/{city}_{seq}_{fr:06d}_vehicle.json``      ``speed`` [m/s], ``yawRate`` [rad/s]
  ``{odom_pred_dir}/{odom_name}_{split}.h5``                                  written by experiments/export_cityscapes_odom.py:37-54:
        dataset ``'%s/%s/%d/%d' % (city, seq, frame, start_frame)`` = ``[n_future, 2]`` rows ``[speed, yaw_rate]``

for the 30 frames ``frame-19 .. frame+10`` of a snippet whose annotated frame is ``frame`` (index 19).  ``OdometryFile``
opens the export (HDF5 through ``h5py`` when it is installed; the same keys in an ``.npz`` archive otherwise — this
image has no h5py, and ``convert_h5_to_npz`` is the one-liner to run where it exists); ``snippet_target_T`` turns one
snippet into the ``target_T [3,4,4]`` the warp consumes, through ``ego.target_T_predicted`` / ``ego.target_T_measured``.
Pinned by tests/golden/g1_odom.npz: outputs of the reference's own dataset constructor on a synthetic directory.
"""
import json
import os

import numpy as np

from . import ego

BASE_INPUT_INDS = np.array([0, 3, 6])       # pc_transform_dataset.py:81
TARGET_INDEX = 19                           # the annotated frame inside the 30-frame snippet (:80)


def odom_key(city, seq, frame, start_frame):
    """export_cityscapes_odom.py:52 / pc_transform_dataset.py:153 (frame is NOT zero-padded here, unlike the depth H5)."""
    return '%s/%s/%d/%d' % (city, seq, int(frame), int(start_frame))


def input_indices(gap_len, target=TARGET_INDEX):
    """pc_transform_dataset.py:94: indices (inside the snippet) of the three input frames."""
    return BASE_INPUT_INDS + target - (6 + gap_len)


class OdometryFile:
    """Read-only view of ``odometry_{split}.h5`` (or its ``.npz`` twin): ``rows(city, seq, frame, start_frame)``."""

    def __init__(self, path):
        self.path = path
        self._h5 = self._npz = None
        if path.endswith('.npz'):
            self._npz = np.load(path)
        else:
            try:
                import h5py
            except ImportError as e:
                twin = os.path.splitext(path)[0] + '.npz'
                if not os.path.exists(twin):
                    raise ImportError('%s is HDF5 and h5py is not installed; convert it once with odom_io.convert_h5_to_npz '
                                      '(needs h5py) or pass the .npz twin' % path) from e
                self._npz = np.load(twin)
            else:
                self._h5 = h5py.File(path, 'r')

    def rows(self, city, seq, frame, start_frame):
        key = odom_key(city, seq, frame, start_frame)
        src = self._h5 if self._h5 is not None else self._npz
        if key not in src:
            raise KeyError('%s has no odometry forecast %r' % (self.path, key))
        return np.asarray(src[key][:] if self._h5 is not None else src[key])

    def close(self):
        if self._h5 is not None:
            self._h5.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def write_npz(path, entries):
    """entries: {(city, seq, frame, start_frame): [n,2] array} -> .npz with the H5 key names."""
    np.savez_compressed(path, **{odom_key(*k): np.asarray(v, dtype=np.float32) for k, v in entries.items()})


def convert_h5_to_npz(h5_path, npz_path=None):
    import h5py
    out = {}
    with h5py.File(h5_path, 'r') as f:
        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                out[name] = obj[:]
        f.visititems(visit)
    npz_path = npz_path or os.path.splitext(h5_path)[0] + '.npz'
    np.savez_compressed(npz_path, **out)
    return npz_path


def read_snippet(cityscapes_dir, split, city, seq, frame):
    """(times [30] seconds, speeds [30], yaw_rates [30]) of frames frame-19 .. frame+10 (:103-116,132-147)."""
    times, speeds, yaws = [], [], []
    for fr in range(frame - TARGET_INDEX, frame + 11):
        stem = '%s_%s_%06d' % (city, seq, fr)
        with open(os.path.join(cityscapes_dir, 'timestamp_sequence', split, city, stem + '_timestamp.txt')) as f:
            times.append(float(f.read()) / 1e9)
        with open(os.path.join(cityscapes_dir, 'vehicle_sequence', split, city, stem + '_vehicle.json')) as f:
            v = json.load(f)
        speeds.append(v.get('speed'))
        yaws.append(v.get('yawRate'))
    return np.asarray(times), speeds, yaws


def snippet_target_T(cityscapes_dir, split, city, seq, frame, gap_len, odom=None, target=TARGET_INDEX):
    """``target_T [3,4,4]`` float64 for one snippet.  ``odom``: an ``OdometryFile`` (predicted-odometry branch, the
    mid-/short-term *forecasting* configuration, :156-186) or None (measured odometry up to the target, :219-231)."""
    times, speeds, yaws = read_snippet(cityscapes_dir, split, city, seq, frame)
    inds = input_indices(gap_len, target)
    if odom is None:
        return ego.target_T_measured(speeds, yaws, times, inds, target)
    preds = odom.rows(city, seq, frame, inds[-1])
    return ego.target_T_predicted(speeds, yaws, times, preds, inds, target, gap_len)
