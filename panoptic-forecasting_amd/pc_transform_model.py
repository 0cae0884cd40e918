"""``task: pc_transform`` — drop-in for reference ``models/pc_transform/pc_transform_model.py``.

Same constructor params (``model.only_this_ind``, ``model.is_img``), same ``predict(inputs, labels)``
dict in / dict out (``seg``, ``depth``, ``result2d``); the work is one ``pf_warp_splat`` call.
"""
import ctypes

import torch

from . import lib as _lib
from .model_api import BaseModel


def host_inverse(m):
    """torch.inverse on the HOST (LAPACK), as the oracle fixes it: the low bits of K^-1 / E^-1 decide
    which pixel floor() picks for near-integer coordinates (pc_transform_model.py:51,71)."""
    return torch.inverse(m.detach().float().cpu()).to(m.device)


def _as_u8(mask):
    """bool tensors are one 0/1 byte per element: reinterpret instead of converting."""
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    return mask if mask.dtype == torch.uint8 else mask.to(torch.uint8)


class WarpSplat:
    """Workspace-caching front end of ``pf_warp_splat``."""

    def __init__(self):
        self._ws = None

    def _workspace(self, b, t, h, w, per_frame, device):
        L = _lib.load()
        need = ctypes.c_size_t()
        _lib.check(L.pf_warp_splat_workspace(b, t, h, w, int(per_frame), ctypes.byref(need)),
                   'pf_warp_splat_workspace')
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != device:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=device)
        return self._ws

    def __call__(self, depth, depth_mask, seg, K, E, target_T, Kinv=None, Einv=None, t_first=0, T=None,
                 per_frame=False, is_img=False, want_result2d=True):
        L = _lib.load()
        b, t_total, h, w = depth.shape
        T = t_total - t_first if T is None else T
        dev = depth.device
        depth = _lib.require_cuda(depth.float(), 'depth')
        mask = _lib.require_cuda(_as_u8(depth_mask), 'depth_mask')
        seg_dtype = seg.dtype
        seg8 = _lib.require_cuda(seg if seg.dtype == torch.uint8 else seg.to(torch.uint8), 'seg')
        Kinv = host_inverse(K) if Kinv is None else Kinv
        Einv = host_inverse(E) if Einv is None else Einv
        mats = [_lib.require_cuda(m.float().contiguous(), n) for m, n in
                ((Kinv, 'Kinv'), (E, 'extrinsics'), (target_T, 'target_T'), (Einv, 'Einv'), (K, 'intrinsics'))]
        c = 3 if is_img else 1
        g = T if per_frame else 1
        shape = (b, g, h, w) if per_frame else (b, h, w)
        out_seg = torch.empty(shape + ((3,) if is_img else ()), dtype=torch.uint8, device=dev)
        out_depth = torch.empty(shape, dtype=torch.float32, device=dev)
        r2d = torch.empty((b, T, h, w, 2), dtype=torch.int64, device=dev) if want_result2d else None
        ws = self._workspace(b, T, h, w, per_frame, dev)
        rc = L.pf_warp_splat(depth.data_ptr(), mask.data_ptr(), seg8.data_ptr(), c,
                             mats[0].data_ptr(), mats[1].data_ptr(), mats[2].data_ptr(), mats[3].data_ptr(),
                             mats[4].data_ptr(), b, t_total, t_first, T, h, w, int(per_frame),
                             out_seg.data_ptr(), out_depth.data_ptr(),
                             r2d.data_ptr() if r2d is not None else None,
                             ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, 'pf_warp_splat')
        if seg_dtype != torch.uint8:
            out_seg = out_seg.to(seg_dtype)
        return out_seg, out_depth, r2d


class PCTransformModel(BaseModel):
    def __init__(self, params):
        super().__init__()
        self.ind = params['model'].get('only_this_ind')
        self.is_img = params['model'].get('is_img')
        self.debug = params['model'].get('debug')
        self._splat = WarpSplat()

    @torch.no_grad()
    def predict(self, inputs, labels=None):
        depth = inputs['depth']
        t_first, T = (0, depth.shape[1]) if self.ind is None else (self.ind, 1)
        seg, dep, r2d = self._splat(depth, inputs['depth_mask'], inputs['seg'], inputs['intrinsics'],
                                    inputs['extrinsics'], inputs['target_T'],
                                    Kinv=inputs.get('intrinsics_inv'), Einv=inputs.get('extrinsics_inv'),
                                    t_first=t_first, T=T, per_frame=False, is_img=bool(self.is_img))
        return {'seg': seg, 'result2d': r2d, 'depth': dep}
