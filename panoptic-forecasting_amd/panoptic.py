"""fg -> panoptic merge, panoptic export encoding and panoptic PQ.

The step after the bg hot path in the reference: ``FGModel.predict_panoptic`` (models/fg/fg_model.py:489-595) pastes
the forecast instance masks, farthest first, over the exported bg label map and
experiments/export_cityscapes_panoptic_results.py writes the result as a COCO-panoptic style PNG + JSON.  The fg
*networks* (MaskRCNN head, ConvLSTM, trajectory GRU) are outside this build's scope; what they hand over — mask
logits ``[N,28,28]``, boxes, depths, classes — is the input here, and everything from the sigmoid (:532) onwards runs
as two HIP kernels behind ``pf_panoptic_merge`` / ``pf_panoptic_encode`` (csrc/panoptic_merge.hip).
"""
import ctypes
import json
import os

import numpy as np
import torch

from . import lib as _lib

TRAINID2ID = (7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33)
PANOPTIC_PNG = '%s_%s_%06d_pred_panoptic.png'


class PanopticMerger:
    """Holds the reference's merge switches (``model.use_depth_sorting`` fg_model.py:37, ``use_bbox_ulbr`` :34)."""

    def __init__(self, params=None, use_depth_sorting=True, use_bbox_ulbr=False):
        if params is not None:
            use_depth_sorting = bool(params.get('model', {}).get('use_depth_sorting'))
            use_bbox_ulbr = bool(params.get('use_bbox_ulbr'))
        self.use_depth_sorting = use_depth_sorting
        self.use_bbox_ulbr = use_bbox_ulbr
        self._ws = None

    def _workspace(self, n, device):
        need = ctypes.c_size_t()
        _lib.check(_lib.load().pf_panoptic_merge_workspace(n, ctypes.byref(need)), 'pf_panoptic_merge_workspace')
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != device:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=device)
        return self._ws

    @torch.no_grad()
    def merge(self, mask_probs, boxes, depths, classes, counts, background=None, background_depth=None,
              background_depth_mask=None, panoptic=True, size=(1024, 2048), out_dtype=torch.int64):
        """mask_probs [N,MH,MW] f32 (after sigmoid), boxes [N,4], depths [N] or None, classes [N] i64 — all images
        concatenated, ``counts[b]`` instances belong to image b; background [B,H,W] u8/i32/i64 or None.
        Returns the merged map [B,H,W] (``(class+11)*1000 + id`` values when ``panoptic`` else ``class+11``)."""
        L = _lib.load()
        b = len(counts)
        n = int(sum(counts))
        dev = mask_probs.device if n else (background.device if background is not None else torch.device('cuda'))
        h, w = (background.shape[-2], background.shape[-1]) if background is not None else size
        offs = torch.zeros(b + 1, dtype=torch.int32)
        offs[1:] = torch.cumsum(torch.tensor(list(counts), dtype=torch.int32), 0)
        offs = offs.to(dev)
        ptr = lambda t: None if t is None else t.data_ptr()
        bg = bg_kind = None
        if background is not None:
            if background.dtype not in (torch.uint8, torch.int32, torch.int64):
                background = background.to(torch.int64)
            bg = _lib.require_cuda(background.contiguous(), 'background')
            bg_kind = {torch.uint8: 0, torch.int32: 1, torch.int64: 2}[bg.dtype]
        bd = _lib.require_cuda(background_depth.float().contiguous(), 'background_depth') if background_depth is not None else None
        bm = None
        if background_depth_mask is not None:
            bm = background_depth_mask.contiguous()
            bm = _lib.require_cuda(bm.view(torch.uint8) if bm.dtype == torch.bool else bm.to(torch.uint8), 'background_depth_mask')
        m = bx = dp = cl = ws = None
        mh = mw = 28
        if n:
            m = _lib.require_cuda(mask_probs.float().contiguous(), 'mask_probs')
            mh, mw = m.shape[-2], m.shape[-1]
            bx = _lib.require_cuda(boxes.float().contiguous(), 'boxes')
            cl = _lib.require_cuda(classes.to(torch.int64).contiguous(), 'classes')
            if depths is not None:
                dp = _lib.require_cuda(depths.float().contiguous(), 'depths')
            ws = self._workspace(n, dev)
        if out_dtype not in (torch.int32, torch.int64):
            raise _lib.PfError('out_dtype must be int32 or int64')
        out = torch.empty((b, h, w), dtype=out_dtype, device=dev)
        rc = L.pf_panoptic_merge(ptr(bg), bg_kind or 0, ptr(bd), ptr(bm), ptr(m), mh, mw, ptr(bx), int(self.use_bbox_ulbr),
                                 ptr(dp), ptr(cl), offs.data_ptr(), n, b, h, w, int(self.use_depth_sorting and dp is not None),
                                 int(panoptic), int(panoptic), out.data_ptr(), int(out_dtype == torch.int64),
                                 ptr(ws), ws.numel() if ws is not None else 0, _lib.stream_ptr())
        _lib.check(rc, 'pf_panoptic_merge')
        return out

    def predict_panoptic(self, pred, classes, background=None, background_depth=None, background_depth_mask=None,
                         panoptic=True):
        """The tail of FGModel.predict_panoptic (:530-588) / predict_semantics (:441-480) given what the fg networks
        produced: ``pred = {'masks' [N,MH,MW] logits, 'boxes' [N,4], 'depths' [N]}`` and the per-image class lists."""
        counts = [len(c) for c in classes]
        probs = torch.sigmoid(pred['masks'])                                       # :532
        bg = torch.stack(list(background)) if isinstance(background, (list, tuple)) else background
        bd = torch.stack(list(background_depth)) if isinstance(background_depth, (list, tuple)) else background_depth
        bm = background_depth_mask
        if isinstance(bm, (list, tuple)):
            bm = torch.stack([x.reshape(x.shape[-2:]) for x in bm])
        seg = self.merge(probs, pred['boxes'], pred.get('depths'), torch.cat(list(classes)), counts, bg, bd, bm, panoptic)
        return {'seg': seg}


@torch.no_grad()
def encode(seg, convert=True, want_ids=False):
    """export_cityscapes_panoptic_results.py:27-68 on the device: seg [B,H,W] i32/i64 ->
    (rgb [B,H,W,3] u8, ids [B,H,W] i32 | None, segments_info per image)."""
    L = _lib.load()
    if seg.dtype not in (torch.int32, torch.int64):
        seg = seg.to(torch.int64)
    seg = _lib.require_cuda(seg.contiguous(), 'seg')
    b, h, w = seg.shape
    max_ids = L.pf_panoptic_max_ids()
    rgb = torch.empty((b, h, w, 3), dtype=torch.uint8, device=seg.device)
    ids = torch.empty((b, h, w), dtype=torch.int32, device=seg.device) if want_ids else None
    present = torch.empty((b, max_ids), dtype=torch.uint8, device=seg.device)
    _lib.check(L.pf_panoptic_encode(seg.data_ptr(), int(seg.dtype == torch.int64), int(convert), b, h, w, rgb.data_ptr(),
                                    ids.data_ptr() if ids is not None else None, present.data_ptr(), _lib.stream_ptr()),
               'pf_panoptic_encode')
    infos = []
    for row in present.cpu().numpy():
        info = []
        for v in np.nonzero(row)[0]:
            v = int(v)
            if v == 0:
                continue                                                           # :58-59
            info.append({'category_id': v // 1000 if v > 100 else v, 'id': v})     # :60-67
        infos.append(info)
    return rgb, ids, infos


def export_panoptic(seg, meta, result_dir, export_name, no_convert=False, annotations=None):
    """The per-batch body of export_results (export_cityscapes_panoptic_results.py:96-125): writes
    ``<result_dir>/<export_name>/{city}_{seq}_{frame:06d}_pred_panoptic.png`` and appends the annotation records."""
    from PIL import Image
    annotations = [] if annotations is None else annotations
    seg_dir = os.path.join(result_dir, export_name)
    os.makedirs(seg_dir, exist_ok=True)
    rgb, _, infos = encode(seg, convert=not no_convert)
    rgb = rgb.cpu().numpy()
    for b in range(rgb.shape[0]):
        city, seq, target = meta['city'][b], meta['seq'][b], int(meta['target_frame'][b])
        name = PANOPTIC_PNG % (city, seq, target)
        annotations.append({'file_name': name, 'image_id': '%s_%s_%06d' % (city, seq, target), 'segments_info': infos[b]})
        Image.fromarray(rgb[b]).save(os.path.join(seg_dir, name))
    return annotations


def write_annotations(annotations, result_dir, export_name):
    """:166-169."""
    path = os.path.join(result_dir, '%s.json' % export_name)
    with open(path, 'w', encoding='utf-8') as f:
        json.dump({'annotations': annotations}, f, ensure_ascii=False, indent=4)
    return path


def decode_png(rgb):
    """R + 256 G + 256^2 B (the id encoding the evaluation script reads back)."""
    rgb = np.asarray(rgb).astype(np.int64)
    return rgb[..., 0] + 256 * rgb[..., 1] + 65536 * rgb[..., 2]
