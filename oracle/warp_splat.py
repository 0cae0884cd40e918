"""ctypes front end of the C oracle ``warp_splat_ref.c`` (TEST INFRASTRUCTURE ONLY).

``predict(inputs, only_this_ind, is_img)`` mirrors the reference call
``PCTransformModel(params).predict(inputs, labels)`` —
/root/reference/panoptic_forecasting/models/pc_transform/pc_transform_model.py:26-150 —
on CPU tensors, returning the same dict (``seg``, ``depth``, ``result2d``) plus debug taps.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE, 'liboracle_warp.so'])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboracle_warp.so')
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.pfo_warp_splat.restype = ctypes.c_int
        _LIB.pfo_warp_splat.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 5 + \
            [ctypes.c_int] * 4 + [ctypes.c_void_p] * 7
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def predict(inputs, only_this_ind=None, is_img=False, debug=False):
    K = inputs['intrinsics'].float().cpu()
    E = inputs['extrinsics'].float().cpu()
    depth, mask, T, seg = inputs['depth'], inputs['depth_mask'], inputs['target_T'], inputs['seg']
    if only_this_ind is not None:                       # pc_transform_model.py:33-37
        s = slice(only_this_ind, only_this_ind + 1)
        depth, mask, T, seg = depth[:, s], mask[:, s], T[:, s], seg[:, s]
    b, t, h, w = depth.shape
    c = 3 if is_img else 1
    Kinv = torch.inverse(K)                              # :51  (LAPACK on the host, like the reference)
    Einv = torch.inverse(E)                              # :71
    f32 = lambda x: np.ascontiguousarray(x.detach().cpu().numpy().astype(np.float32))
    d_np, T_np = f32(depth), f32(T)
    m_np = np.ascontiguousarray(mask.detach().cpu().numpy().astype(np.uint8))
    s_np = np.ascontiguousarray(seg.detach().cpu().numpy().astype(np.uint8))
    out_seg = np.zeros((b, h, w, c), np.uint8)
    out_depth = np.zeros((b, h, w), np.float32)
    out_r2d = np.zeros((b, t, h, w, 2), np.int64)
    n = h * w
    uvz = np.zeros((b, t, n, 3), np.float32) if debug else None
    inds = np.zeros((b, 4, t, n), np.int64) if debug else None
    arg = np.zeros((b, n), np.int64) if debug else None
    zmax = np.zeros(1, np.float32)
    rc = lib().pfo_warp_splat(_p(d_np), _p(m_np), _p(s_np), c, _p(f32(Kinv)), _p(f32(E)), _p(T_np),
                              _p(f32(Einv)), _p(f32(K)), b, t, h, w, _p(out_seg), _p(out_depth),
                              _p(out_r2d), _p(uvz), _p(inds), _p(arg), _p(zmax))
    if rc != 0:
        raise MemoryError('oracle allocation failed')
    seg_out = torch.from_numpy(out_seg if is_img else out_seg[..., 0]).to(seg.dtype)
    res = {'seg': seg_out, 'depth': torch.from_numpy(out_depth), 'result2d': torch.from_numpy(out_r2d)}
    if debug:
        res.update(uvz=torch.from_numpy(uvz), scatter_inds=torch.from_numpy(inds),
                   argmin=torch.from_numpy(arg), zmax=float(zmax[0]))
    return res
