"""ctypes front end of the C oracle ``panoptic_merge_ref.c`` (TEST INFRASTRUCTURE ONLY).

``merge(...)`` restates the pasting loops of the reference's ``FGModel.predict_panoptic`` / ``FGModel.predict_semantics``
(/root/reference/panoptic_forecasting/models/fg/fg_model.py:548-588, :455-480) on CPU tensors, taking what those
loops consume: per-image lists of mask probabilities, boxes, depths and classes plus the background canvas.
``encode(seg)`` restates export_cityscapes_panoptic_results.py:27-68.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MAX_IDS = 34000


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboracle_panoptic.so')
        if not os.path.exists(path):
            subprocess.check_call(['make', '-s', '-C', _HERE, 'liboracle_panoptic.so'])
        _LIB = ctypes.CDLL(path)
        vp, i = ctypes.c_void_p, ctypes.c_int
        _LIB.pfo_panoptic_merge.restype = i
        _LIB.pfo_panoptic_merge.argtypes = [vp, vp, vp, vp, i, i, vp, i, vp, vp, vp, i, i, i, i, i, i, vp]
        _LIB.pfo_panoptic_encode.restype = i
        _LIB.pfo_panoptic_encode.argtypes = [vp, ctypes.c_size_t, i, vp, vp, vp, i]
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _np(t, dt):
    return None if t is None else np.ascontiguousarray(t.detach().cpu().numpy().astype(dt))


def merge(masks, boxes, depths, classes, h, w, background=None, background_depth=None, background_depth_mask=None,
          use_depth_sorting=True, use_bbox_ulbr=False, panoptic=True):
    """masks/boxes/depths/classes: per-image lists ([n_b,MH,MW] f32 probabilities, [n_b,4], [n_b], [n_b] long);
    background [B,H,W] long or None.  Returns [B,H,W] int64."""
    b = len(masks)
    counts = [int(m.shape[0]) for m in masks]
    offs = np.zeros(b + 1, np.int32)
    offs[1:] = np.cumsum(counts)
    mh, mw = (int(masks[0].shape[-2]), int(masks[0].shape[-1])) if b else (28, 28)
    cat = lambda xs, dt, shape: _np(torch.cat([x.reshape(shape) for x in xs]) if xs else torch.zeros(shape), dt)
    m_np = cat(masks, np.float32, (-1, mh, mw))
    b_np = cat(boxes, np.float32, (-1, 4))
    d_np = cat(depths, np.float32, (-1,)) if depths is not None else None
    c_np = cat(classes, np.int64, (-1,))
    bg = _np(background, np.int64)
    bd = _np(background_depth, np.float32)
    bm = _np(background_depth_mask, np.uint8)
    out = np.zeros((b, h, w), np.int64)
    rc = lib().pfo_panoptic_merge(_p(bg), _p(bd), _p(bm), _p(m_np), mh, mw, _p(b_np), int(bool(use_bbox_ulbr)), _p(d_np),
                                  _p(c_np), _p(offs), b, h, w, int(bool(use_depth_sorting)), int(bool(panoptic)),
                                  int(bool(panoptic)), _p(out))
    if rc:
        raise MemoryError('oracle allocation failed')
    return torch.from_numpy(out)


def encode(seg, convert=True):
    """seg [H,W] long -> (rgb [H,W,3] u8, ids [H,W] i32, sorted list of ids present)."""
    s = _np(seg, np.int64)
    rgb = np.zeros(s.shape + (3,), np.uint8)
    ids = np.zeros(s.shape, np.int32)
    present = np.zeros(MAX_IDS, np.uint8)
    lib().pfo_panoptic_encode(_p(s), s.size, int(bool(convert)), _p(rgb), _p(ids), _p(present), MAX_IDS)
    return rgb, ids, [int(i) for i in np.nonzero(present)[0]]
