#!/usr/bin/env python
"""Golden fixtures for the fg -> panoptic merge (scope row f3), produced by the REFERENCE itself (build container only).

    python tests/golden/make_golden_fg.py

``FGModel.predict_panoptic`` / ``FGModel.predict_semantics`` (/root/reference/panoptic_forecasting/models/fg/fg_model.py:489-595,
:395-487) are run unmodified; only the fg NETWORKS are out of scope, so the model object is created without
``__init__`` and its ``forward`` is replaced by a stub that returns prepared mask logits / trajectories — everything
after that call (sigmoid, depth sort, model_utils.paste_mask + F.grid_sample, thresholding, per-class ids, the paste
rules) is the reference's own code and arithmetic.  Likewise ``convert_labels`` / ``create_pan_img`` /
``get_segments_info`` of experiments/export_cityscapes_panoptic_results.py are imported and called as they are
(``cityscapesscripts.helpers.labels`` is absent: a stand-in module exposes the public Cityscapes trainId -> id table).

The reference hard-codes the canvas size 1024 x 2048 (fg_model.py:566), so the fixtures are full size; label maps
compress to a few hundred KB.  Fixtures (data only):
  g5_panoptic.npz       predict_panoptic, use_depth_sorting, no background depth (the shipped configuration)
  g5_panoptic_z.npz     predict_panoptic with background_depth + background_depth_mask (z-test rule)
  g5_panoptic_ulbr.npz  predict_panoptic, use_bbox_ulbr, no depth sorting
  g5_semantic.npz       predict (class+11 values, things kept in the background)
  g5_encode.npz         convert_labels + create_pan_img + get_segments_info of the g5_panoptic result
"""
import os
import sys
import types
from collections import namedtuple

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import  # noqa: E402

_ref_import.install()
for name, sub in [('panoptic_forecasting.models.fg', 'models/fg'), ('panoptic_forecasting.experiments', 'experiments'),
                  ('panoptic_forecasting.utils', 'utils'), ('panoptic_forecasting.training', 'training')]:
    m = types.ModuleType(name)
    m.__path__ = [os.path.join(_ref_import.REF_ROOT, 'panoptic_forecasting', sub)]
    sys.modules.setdefault(name, m)
# names the export script imports but the functions used here never touch
sys.modules['panoptic_forecasting.data'].build_dataset = None
sys.modules['panoptic_forecasting.models'].build_model = None
for stub in ('panoptic_forecasting.utils.misc', 'panoptic_forecasting.training.train_utils'):
    sys.modules.setdefault(stub, types.ModuleType(stub))
cfg = types.ModuleType('panoptic_forecasting.utils.config')
cfg.load_config = None
sys.modules.setdefault('panoptic_forecasting.utils.config', cfg)
if 'tqdm' not in sys.modules:
    try:
        import tqdm  # noqa: F401
    except ImportError:
        t = types.ModuleType('tqdm')
        t.tqdm = lambda x, **k: x
        sys.modules['tqdm'] = t
# stand-in for cityscapesscripts.helpers.labels: the public trainId -> id table only
_Label = namedtuple('Label', ['id', 'trainId'])
_TID2ID = [7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33]
cs = types.ModuleType('cityscapesscripts')
csh = types.ModuleType('cityscapesscripts.helpers')
csl = types.ModuleType('cityscapesscripts.helpers.labels')
csl.trainId2label = {t: _Label(i, t) for t, i in enumerate(_TID2ID)}
sys.modules.update({'cityscapesscripts': cs, 'cityscapesscripts.helpers': csh, 'cityscapesscripts.helpers.labels': csl})

from panoptic_forecasting.models.fg.fg_model import FGModel  # noqa: E402
from panoptic_forecasting.experiments import export_cityscapes_panoptic_results as ref_export  # noqa: E402

torch.set_grad_enabled(False)
H, W = 1024, 2048


def blocky_background(seed, b):
    g = torch.Generator().manual_seed(seed)
    low = torch.randint(0, 19, (b, H // 64, W // 64), generator=g)
    low[torch.rand(low.shape, generator=g) < 0.05] = 255
    return low.repeat_interleave(64, 1).repeat_interleave(64, 2).long()


def make_case(seed, counts, ulbr=False):
    g = torch.Generator().manual_seed(seed)
    n = sum(counts)
    # mask logits: smooth blobs so the 0.5 contour is non-trivial
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 28), torch.linspace(-1, 1, 28), indexing='ij')
    logits = []
    for _ in range(n):
        cx, cy = (torch.rand(2, generator=g) - 0.5) * 0.4
        rx, ry = 0.5 + torch.rand(2, generator=g) * 0.5
        blob = 4.0 * (1.0 - ((xx - cx) / rx) ** 2 - ((yy - cy) / ry) ** 2)
        logits.append(blob + 0.7 * torch.randn(28, 28, generator=g))
    logits = torch.stack(logits)
    cxs = torch.rand(n, generator=g) * (W + 200) - 100            # some boxes hang over the border
    cys = torch.rand(n, generator=g) * (H + 100) - 50
    ws = 20 + torch.rand(n, generator=g) * 400
    hs = 20 + torch.rand(n, generator=g) * 300
    ws[0], hs[0] = 3.3, 2.2                                        # a tiny instance
    ws[1], hs[1] = 1900.5, 900.25                                  # a huge one
    boxes = torch.stack([cxs, cys, ws, hs], 1)
    if ulbr:
        boxes = torch.stack([cxs - ws / 2, cys - hs / 2, cxs + ws / 2, cys + hs / 2], 1)
    depths = 5 + torch.rand(n, generator=g) * 60
    depths[3] = depths[2]                                          # a depth tie inside image 0 (stable order)
    depths[counts[0] + 1] = depths[counts[0]]                      # and one inside image 1
    classes = torch.randint(0, 8, (n,), generator=g)
    return logits, boxes, depths, classes


def run_reference(method, logits, boxes, depths, classes, counts, background, bg_depth=None, bg_mask=None,
                  use_depth_sorting=True, ulbr=False):
    model = FGModel.__new__(FGModel)
    torch.nn.Module.__init__(model)
    model.use_depth_sorting = use_depth_sorting
    model.use_bbox_ulbr = ulbr
    model.use_depth_inp = True
    model.only_loc_feats = False
    n = sum(counts)
    out_t = 3
    traj = torch.zeros(n, out_t, 9)
    output_inds = torch.full((n,), out_t - 1, dtype=torch.long)
    traj[:, out_t - 1, :4] = boxes
    traj[:, out_t - 1, 8] = depths
    model.forward = lambda *a, **k: {'unnormalized_trajectory': traj, 'masks': logits}
    split = lambda t: list(t.split(counts))
    inputs = {'trajectories': split(torch.zeros(n, 3, 9)), 'bbox_masks': split(torch.ones(n, 6)),
              'bbox_vel_masks': split(torch.ones(n, 6)), 'feats': split(torch.zeros(n, 1)), 'classes': split(classes)}
    if background is not None:
        inputs['background'] = [x.clone() for x in background]
    if bg_depth is not None:
        inputs['background_depth'] = [x.clone() for x in bg_depth]
        # predict_panoptic indexes a [1,H,W] depth slice with the mask (fg_model.py:561-564): each mask item is [1,H,W]
        inputs['background_depth_mask'] = [x.clone().unsqueeze(0) if method == 'predict_panoptic' else x.clone() for x in bg_mask]
    labels = {'trajectories': split(torch.zeros(n, out_t, 9)), 'output_inds': split(output_inds)}
    return getattr(model, method)(inputs, labels)['seg']


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print(name, os.path.getsize(os.path.join(HERE, name)) // 1024, 'KiB')


def pack_inputs(logits, boxes, depths, classes, counts, background):
    return dict(mask_prob=torch.sigmoid(logits).numpy(), boxes=boxes.numpy(), depths=depths.numpy(),
                classes=classes.numpy(), counts=np.array(counts, np.int32),
                background=background.numpy().astype(np.uint8))


def main():
    counts = [14, 9]
    bg = blocky_background(3, 2)
    lg, bx, dp, cl = make_case(11, counts)
    seg = run_reference('predict_panoptic', lg, bx, dp, cl, counts, bg)
    save('g5_panoptic.npz', seg=seg.numpy().astype(np.int32), **pack_inputs(lg, bx, dp, cl, counts, bg))

    g = torch.Generator().manual_seed(5)
    bgd = 10 + 50 * torch.rand(2, H // 32, W // 32, generator=g).repeat_interleave(32, 1).repeat_interleave(32, 2)
    bgm = (torch.rand(2, H // 16, W // 16, generator=g) < 0.9).repeat_interleave(16, 1).repeat_interleave(16, 2)
    # bg_depth is stored as float16 (small file); the expectation is generated from exactly those values
    bgd16 = torch.from_numpy(bgd.numpy().astype(np.float16).astype(np.float32))
    segz = run_reference('predict_panoptic', lg, bx, dp, cl, counts, bg, bgd16, bgm)
    save('g5_panoptic_z.npz', seg=segz.numpy().astype(np.int32), bg_depth=bgd16.numpy().astype(np.float16),
         bg_depth_mask=np.packbits(bgm.numpy()), **pack_inputs(lg, bx, dp, cl, counts, bg))

    counts_u = [7, 5]
    lgu, bxu, dpu, clu = make_case(23, counts_u, ulbr=True)
    segu = run_reference('predict_panoptic', lgu, bxu, dpu, clu, counts_u, bg, use_depth_sorting=False, ulbr=True)
    save('g5_panoptic_ulbr.npz', seg=segu.numpy().astype(np.int32), **pack_inputs(lgu, bxu, dpu, clu, counts_u, bg))

    segs = run_reference('predict_semantics', lg, bx, dp, cl, counts, bg)
    save('g5_semantic.npz', seg=segs.numpy().astype(np.int32), **pack_inputs(lg, bx, dp, cl, counts, bg))

    one = seg[0].numpy()
    conv = ref_export.convert_labels(one)
    pan = np.array(ref_export.create_pan_img(conv))
    info = ref_export.get_segments_info(conv)
    save('g5_encode.npz', seg=one.astype(np.int32), converted=conv.astype(np.int32), rgb=pan,
         seg_ids=np.array([s['id'] for s in info], np.int64), cat_ids=np.array([s['category_id'] for s in info], np.int64))


if __name__ == '__main__':
    main()
