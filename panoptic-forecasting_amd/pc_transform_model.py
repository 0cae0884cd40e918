"""``task: pc_transform`` — drop-in for reference ``models/pc_transform/pc_transform_model.py``.

Same constructor params (``model.only_this_ind``, ``model.is_img``), same ``predict(inputs, labels)``
dict in / dict out (``seg``, ``depth``, ``result2d``); the work is one ``pf_warp_splat`` call.
"""
import ctypes

import torch

from . import lib as _lib
from .model_api import BaseModel


PF_SPLAT_PER_FRAME, PF_SPLAT_PER_SAMPLE_SENTINEL = 1, 2      # include/pfhip.h


def host_inverse(m):
    """torch.inverse on the HOST (LAPACK), as the oracle fixes it: the low bits of K^-1 / E^-1 decide
    which pixel floor() picks for near-integer coordinates (pc_transform_model.py:51,71).  LAPACK hands back column-major
    matrices (a strided view); the library wants dense rows, and a ``.contiguous()`` left to every ``predict`` is a copy
    kernel per matrix per forward - two 4.6 us links in the B = 1 chain of 80 launches (tools/serial_timeline.py) - so the
    copy is made here, once, on the host."""
    return torch.inverse(m.detach().float().cpu()).contiguous().to(m.device)


def add_camera_inverses(inputs):
    """HOST side, before the batch moves to the device (the reference's loop: ``batch2gpu`` in training/train_utils.py:45-62,
    right before ``model.predict``, export_cityscapes_segmentation_results.py:75-85): ``intrinsics_inv`` / ``extrinsics_inv``
    = ``torch.inverse`` of the host tensors - the same LAPACK call and bits ``PCTransformModel.predict`` would produce
    (pc_transform_model.py:51,71) - added to a shallow copy of ``inputs``.  ``predict`` then finds the inverses in the batch
    and never reads a camera tensor back from the device: no stream synchronisation for batches whose camera tensors it has
    never seen (every batch of a real data loader).  Device-resident or absent cameras are left alone (the model's cache
    handles them, one device->host copy per distinct tensor)."""
    K, E = inputs.get('intrinsics'), inputs.get('extrinsics')
    if not (torch.is_tensor(K) and torch.is_tensor(E)) or K.is_cuda or E.is_cuda:
        return inputs
    out = dict(inputs)
    if out.get('intrinsics_inv') is None:
        out['intrinsics_inv'] = torch.inverse(K.detach().float()).contiguous()
    if out.get('extrinsics_inv') is None:
        out['extrinsics_inv'] = torch.inverse(E.detach().float()).contiguous()
    return out


class InverseCache:
    """K^-1 / E^-1 per distinct camera tensor.  The inverse has to come from host LAPACK (see ``host_inverse``), which
    costs a device->host copy = a stream sync; cameras are per-sequence constants, so ``predict`` pays it once per
    (tensor storage, version) and afterwards enqueues without touching the host (include/pfhip.h: calls only enqueue).
    Entries keep the source tensor alive, so a data_ptr can not be recycled for another matrix while it is a key; an
    in-place edit bumps ``_version`` and misses: a hit on (storage, version, shape, dtype, device) enqueues nothing and
    waits for nothing.  Writes that bypass the version counter (``K.data.copy_``, a custom kernel or the C library writing
    into a staging tensor, numpy-aliased memory) are NOT seen by that key; ``verify = True`` (``PF_VERIFY_CAMERA_CACHE=1``
    in the environment, or ``pc_transform_model._inverse_cache.verify = True``) adds a content check for such callers: an
    entry keeps a device clone of the matrix it inverted and a hit is honoured only if the tensor still equals it - a
    device comparison read on the host, i.e. one stream synchronisation per camera tensor per predict (skipped under
    stream capture, where nothing may synchronise).  HOST tensors are always content-checked on a hit (``torch.equal`` on 9 /
    16 floats, no stream involved): numpy-aliased camera matrices edited in place are the likeliest stale-key case.
    A batch that carries ``intrinsics_inv`` / ``extrinsics_inv`` (``add_camera_inverses``) never comes here."""

    def __init__(self, capacity=16, verify=None, verify_every=64):
        import os
        self.capacity, self._d = capacity, {}
        self.verify = bool(int(os.environ.get('PF_VERIFY_CAMERA_CACHE', '0'))) if verify is None else bool(verify)
        # round 6: without `verify`, a DEVICE tensor's hit is still content-checked on its first and then every
        # `verify_every`-th hit (one stream synchronisation per 64 predicts per camera tensor; 0 = never): a write that
        # bypasses the version counter is found within that many frames instead of never.  The reference inverts every call
        # (pc_transform_model.py:51,71); callers that cannot tolerate the window use verify = True or add_camera_inverses.
        self.verify_every = int(os.environ.get('PF_VERIFY_CAMERA_EVERY', verify_every))
        self._hits = {}
        self._warned, self._dev_misses = False, 0

    def __call__(self, m):
        if m.is_inference():
            # no version counter to key on (reading `_version` raises): inverted every call, like the reference
            return host_inverse(m)
        key = (m.data_ptr(), m._version, tuple(m.shape), m.dtype, str(m.device))
        hit = self._d.get(key)
        capturing = m.is_cuda and torch.cuda.is_current_stream_capturing()
        if m.is_cuda:
            check = self.verify and not capturing
            if hit is not None and not check and not capturing and self.verify_every > 0:
                n = self._hits.get(key, 0)
                self._hits[key] = n + 1
                check = n % self.verify_every == 0
        else:
            check = True
        if hit is not None and check and not torch.equal(m, hit[2]):
            hit = None               # same storage, same version, other numbers
        if hit is None:
            self._d.pop(key, None)
            self._hits.pop(key, None)
            if len(self._d) >= self.capacity:
                old = next(iter(self._d))
                self._d.pop(old)
                self._hits.pop(old, None)
            self._dev_misses += int(m.is_cuda)
            if self._dev_misses == 9 and not self._warned:       # K and E of four batches were cached constants at most
                import warnings
                self._warned = True
                warnings.warn('panoptic_forecasting_amd: camera matrices arrive as new device tensors: every predict() reads them '
                              'back and inverts on the host (two stream synchronisations).  Take the inverses on the host batch '
                              'with pc_transform_model.add_camera_inverses(inputs) before moving it to the GPU (INTEGRATION.md A).')
            hit = (m, host_inverse(m), m.detach().clone())
            self._d[key] = hit
        return hit[1]


_inverse_cache = InverseCache()


def _as_u8(mask):
    """bool tensors are one 0/1 byte per element: reinterpret instead of converting; other dtypes mean "!= 0" (a float
    mask of 0.5 is True for the reference's ``bool_mask &`` arithmetic, not truncated to 0)."""
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    return mask if mask.dtype == torch.uint8 else (mask != 0).view(torch.uint8)


def _seg_as_u8(seg):
    """The device splat carries labels as bytes.  Wider integer maps are accepted when every value fits (checked: one
    device reduction + sync per call for non-u8 inputs — pass u8 to stay asynchronous); anything else is an error, never
    a silent wrap-around (the reference gathers in the input dtype, pc_transform_model.py:120-131)."""
    if seg.dtype == torch.uint8:
        return seg
    if seg.is_floating_point():
        raise _lib.PfError('seg must be an integer label map or a u8 image (got %s)' % seg.dtype)
    if seg.numel() and (int(seg.max()) > 255 or int(seg.min()) < 0):
        raise _lib.PfError('seg holds values outside 0..255 (min %d, max %d): the device splat gathers u8 payloads'
                           % (int(seg.min()), int(seg.max())))
    return seg.to(torch.uint8)


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
                 per_frame=False, is_img=False, want_result2d=True, per_sample_sentinel=False):
        L = _lib.load()
        b, t_total, h, w = depth.shape
        T = t_total - t_first if T is None else T
        dev = depth.device
        depth = _lib.require_cuda(depth.float(), 'depth')
        mask = _lib.require_cuda(_as_u8(depth_mask), 'depth_mask')
        seg_dtype = seg.dtype
        seg8 = _lib.require_cuda(_seg_as_u8(seg), 'seg')
        Kinv = _inverse_cache(K) if Kinv is None else Kinv
        Einv = _inverse_cache(E) if Einv is None else Einv
        mats = [_lib.require_cuda(m.float().contiguous(), n) for m, n in
                ((Kinv, 'Kinv'), (E, 'extrinsics'), (target_T, 'target_T'), (Einv, 'Einv'), (K, 'intrinsics'))]
        c = 3 if is_img else 1
        g = T if per_frame else 1
        shape = (b, g, h, w) if per_frame else (b, h, w)
        out_seg = torch.empty(shape + ((3,) if is_img else ()), dtype=torch.uint8, device=dev)
        out_depth = torch.empty(shape, dtype=torch.float32, device=dev)
        r2d = torch.empty((b, T, h, w, 2), dtype=torch.int64, device=dev) if want_result2d else None
        ws = self._workspace(b, T, h, w, per_frame, dev)
        flags = (PF_SPLAT_PER_FRAME if per_frame else 0) | (PF_SPLAT_PER_SAMPLE_SENTINEL if per_sample_sentinel else 0)
        rc = L.pf_warp_splat(depth.data_ptr(), mask.data_ptr(), seg8.data_ptr(), c,
                             mats[0].data_ptr(), mats[1].data_ptr(), mats[2].data_ptr(), mats[3].data_ptr(),
                             mats[4].data_ptr(), b, t_total, t_first, T, h, w, flags,
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
        # opt-in (not a reference key): sentinel max+1 per sample instead of over the batch of the call
        self.per_sample_sentinel = bool(params['model'].get('per_sample_sentinel', False))
        self._splat = WarpSplat()

    @torch.no_grad()
    def predict(self, inputs, labels=None):
        depth = inputs['depth']
        t_first, T = (0, depth.shape[1]) if self.ind is None else (self.ind, 1)
        seg, dep, r2d = self._splat(depth, inputs['depth_mask'], inputs['seg'], inputs['intrinsics'],
                                    inputs['extrinsics'], inputs['target_T'],
                                    Kinv=inputs.get('intrinsics_inv'), Einv=inputs.get('extrinsics_inv'),
                                    t_first=t_first, T=T, per_frame=False, is_img=bool(self.is_img),
                                    per_sample_sentinel=self.per_sample_sentinel)
        return {'seg': seg, 'result2d': r2d, 'depth': dep}
