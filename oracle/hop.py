"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

numpy/torch restatement of the on-disk hop between the reference's two tasks:

  export side  /root/reference/panoptic_forecasting/experiments/export_cityscapes_segmentation_results.py
               :34-38  (label id -> trainId LUT, ``--convert_to_trainid``)
               :119-124 (depth -> round(clamp(d+1, 0, 255) * 256) -> uint16 PNG)
  load side    /root/reference/panoptic_forecasting/data/datasets/bg_dataset.py
               :224-230 (x/256 - 1, mask = d > 0, d[~mask] = -1) and :166-170 (clamp to [min_depth, max_depth])
  one-hot      /root/reference/panoptic_forecasting/models/bg/bg_model.py:53-59

Pinned by tests/golden/g2_glue.npz (generator tests/golden/make_golden.py::gen_g2; the one-hot there is
the output of the reference's own ``BGModel._inp2onehot``).
"""
import numpy as np
import torch

# Cityscapes label table (public dataset constants, cityscapesscripts.helpers.labels): id -> trainId
_PAIRS = {7: 0, 8: 1, 11: 2, 12: 3, 13: 4, 17: 5, 19: 6, 20: 7, 21: 8, 22: 9, 23: 10,
          24: 11, 25: 12, 26: 13, 27: 14, 28: 15, 31: 16, 32: 17, 33: 18}


def id2trainid_lut():
    """export :34-38: ``result = zeros_like(seg); for label in labels: result[seg == label.id] = label.trainId``."""
    lut = np.zeros(256, np.uint8)
    lut[:34] = 255
    for i, t in _PAIRS.items():
        lut[i] = t
    return lut


def export_depth_u16(d):
    """:119-124."""
    return ((d + 1).clamp(0, 255) * 256).round().numpy().astype(np.uint16)


def load_depth(q, min_depth=0.1, max_depth=200.0):
    """bg_dataset.py:224-230,166-170 -> (depth f32, mask bool)."""
    d = torch.from_numpy(q.astype(np.float32)) / 256.0 - 1
    m = d > 0
    d[~m] = -1
    d[m & (d > max_depth)] = max_depth
    d[m & (d < min_depth)] = min_depth
    return d, m


def onehot(seg, n_cls=11):
    """bg_model.py:53-59 (without the in-place mutation of the caller's tensor)."""
    seg = seg.long()
    m = seg < n_cls
    oh = torch.nn.functional.one_hot(torch.where(m, seg, torch.zeros_like(seg)), n_cls) * m.unsqueeze(-1)
    return oh.permute(0, 1, 4, 2, 3).float()
