#!/usr/bin/env python
"""Export driver with the reference's flags — drop-in for ``experiments/export_cityscapes_segmentation_results.py``.

    python -u panoptic-forecasting_amd/export_bg.py --config_file configs/bg/bg_val_short.yaml \\
        --load_model pretrained_models/bg/bg_model.pt --no_convert --export_name exported_predictions_short_trainids \\
        --working_dir experiments/pretrained_bg/

i.e. ``scripts/bg/run_export_bg_val.sh`` with the python path changed.  Same flags (``:170-181`` + the base set of
``utils/config.py:34-45``), same output tree (``<working_dir>/<export_name|exported_predictions>/<split>/<city>/…png``,
``:65-70``), same conversion rules (``--no_convert`` / ``--convert_to_trainid`` / ``--is_img``), same depth exports
(``--save_depth`` ``.npy`` / ``--save_depth_as_png`` u16 code), same fill of missing frames (``:129-165``).  The model
comes from this package's registry (tasks ``bg``, ``pc_transform``, ``bg_forecast``), the per-batch conversion +
quantisation run on the device (``hop_io.export_batch``).  ``--viz`` (colour maps) and ``--save_disp_as_png`` are
visualisation-only branches of the reference and are refused here.

Datasets are outside the hot path: ``--dataset reference`` (default) builds them with the reference's own
``panoptic_forecasting.data.build_dataset`` when that package is importable; ``--synthetic N`` exports N synthetic
Cityscapes-shaped samples instead (smoke runs, tests).  Under ``torchrun`` the samples are sharded round-robin over the
ranks (``dist.shard_indices``) — files are independent, no collective is needed.
"""
import contextlib
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if __package__ in (None, ''):                     # run as a script: make the package importable under its alias
    sys.path.insert(0, os.path.dirname(_HERE))
    import panoptic_forecasting_amd  # noqa: F401
    __package__ = 'panoptic_forecasting_amd'

from . import config as pfconfig   # noqa: E402
from . import dist as pfdist       # noqa: E402
from . import hop_io               # noqa: E402
from . import synth                # noqa: E402
from .pc_transform_model import add_camera_inverses  # noqa: E402
from .registry import build_model  # noqa: E402

EXTRA_FLAGS = (
    ('--viz', dict(action='store_true')),
    ('--is_img', dict(action='store_true')),
    ('--save_depth', dict(action='store_true')),
    ('--save_depth_as_png', dict(action='store_true')),
    ('--save_disp_as_png', dict(action='store_true')),
    ('--disp_factor', dict(type=float)),
    ('--export_name', {}),
    ('--no_convert', dict(action='store_true')),
    ('--convert_to_trainid', dict(action='store_true')),
    # not reference flags:
    ('--synthetic', dict(type=int, default=0, help='export N synthetic samples instead of a dataset')),
    ('--dataset', dict(default='reference', choices=['reference'])),
    ('--pipeline_depth', dict(type=int, default=3, help='batches in flight: N model replicas on N HIP streams, batch k + 1 is enqueued '
                                                        'before the files of batch k are written (1 = the plain serial loop)')),
    ('--dry_run', dict(action='store_true', help='with --synthetic: no model and no GPU - every rank writes a constant placeholder map '
                                                 'per sample of its shard (gloo rendezvous), then the barrier and the fill of missing frames '
                                                 'run as in a real export: a check of the driver\'s sharding, never a prediction')),
)


def to_device(item):
    """``training/train_utils.batch2gpu`` for one value: tensors move, containers recurse, everything else stays."""
    if isinstance(item, dict):
        return {k: to_device(v) for k, v in item.items()}
    if isinstance(item, (list, tuple)):
        return type(item)(to_device(v) for v in item)
    return item.cuda(non_blocking=True) if torch.is_tensor(item) else item


class SyntheticDataset(torch.utils.data.Dataset):
    """N samples shaped like the reference datasets' items ({'inputs','labels','meta'}) for the task at hand."""

    def __init__(self, params, n, split='val'):
        self.n, self.split, self.task = n, split, params.get('task')
        self.dry = bool(params.get('dry_run'))
        mp = params.get('model', {})
        self.h, self.w = mp.get('final_h') or 1024, mp.get('final_w') or 2048
        gaps = params.get('data', {}).get('gap_len', 3)
        self.gap = gaps[0] if isinstance(gaps, (list, tuple)) else gaps

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        meta = {'city': 'synth', 'seq': '%06d' % i, 'frame': 19, 'target_frame': 19}
        if self.dry:
            # a dry run carries no tensors; rank r takes r x 0.2 s per sample, so that ranks finish their shards at different
            # times and a fill of missing frames that did not wait for the barrier would count the slower rank's frames
            import time
            time.sleep(0.2 * pfdist.env_rank()[0])
            return {'inputs': {}, 'labels': {}, 'meta': meta}
        if self.task == 'bg':
            inp = synth.make_bg_inputs(b=1, h=self.h, w=self.w, seed=i)
        else:
            inp = synth.make_inputs(b=1, h=self.h, w=self.w, seed=i, gap_len=self.gap, predicted=self.gap > 3)
        inp = {k: v[0] for k, v in inp.items()}
        return {'inputs': inp, 'labels': {}, 'meta': meta}


def collate(items):
    out = {'inputs': {k: torch.stack([it['inputs'][k] for it in items]) for k in items[0]['inputs']}, 'labels': {},
           'meta': {k: [it['meta'][k] for it in items] for k in items[0]['meta']}}
    return out


def build_datasets(params):
    if params.get('synthetic'):
        splits = params.get('data', {}).get('data_splits') or ['val']
        return {s: SyntheticDataset(params, params['synthetic'], s) for s in splits}, collate
    try:
        from panoptic_forecasting.data import build_dataset
    except ImportError as e:
        raise SystemExit('export_bg: the dataset classes are the reference\'s (panoptic_forecasting.data); install that '
                         'package next to this one or use --synthetic N (%s)' % e)
    data = build_dataset(params, test=True)
    return data, params.get('collate_fn')


def dry_write(meta, base):
    """--dry_run: a constant 8x16 placeholder map under the name a real export would give the sample's prediction."""
    import numpy as np
    written = []
    for city, seq, target in zip(meta['city'], meta['seq'], meta['target_frame']):
        os.makedirs(os.path.join(base, city), exist_ok=True)
        path = os.path.join(base, city, hop_io.LABEL_PNG % (city, seq, int(target)))
        hop_io.write_png(path, np.full((8, 16), int(seq) % 19, np.uint8))
        written.append(path)
    return written


def export_split(model, dataset, split, params, collate_fn):
    if params.get('viz') or params.get('save_disp_as_png'):
        raise SystemExit('export_bg: --viz / --save_disp_as_png are visualisation branches of the reference; not built here')
    working_dir = params['working_dir']
    name = params.get('export_name') or 'exported_predictions'
    base = os.path.join(working_dir, name, split)
    rank, world, _ = pfdist.env_rank()
    full_dataset = dataset           # .split / .background_dir live on the dataset, not on the shard view of it
    if world > 1:
        dataset = torch.utils.data.Subset(dataset, pfdist.shard_indices(len(dataset), rank, world))
    tr = params.get('training', {})
    loader = torch.utils.data.DataLoader(dataset, batch_size=tr.get('batch_size', 2), collate_fn=collate_fn,
                                         num_workers=tr.get('num_data_workers', 0), pin_memory=False)
    written = []
    # Round 6: the loop is a STREAM of batches, so several are kept in flight (--pipeline_depth N, default 3; measured at B = 1 / 2:
    # one 970 / 1380, two 1300 / 1820, three 1495 / 2030, four 1575 / 2090 frames/s - tools/pipeline_depth.py): replica k % N of the
    # model enqueues batch k on its own HIP stream, and only then are the files of batch k - N + 1 written.  At the reference's
    # batch size (2) one forward leaves most of the chip idle (78 dependent launches of 5-30 us); the warp/splat + stem of the
    # next batch now run beside the network of the current one, and the host-side PNG encoding overlaps both.  predict()'s
    # contract is untouched (it only ever enqueued); files are byte-identical to the serial loop's (tests/test_gpu_hop_io.py).
    models = model if isinstance(model, (list, tuple)) else [model]
    depth = 1 if (params.get('no_gpu') or params.get('dry_run')) else len(models)
    streams = [torch.cuda.Stream() for _ in range(depth)] if depth > 1 else [None]
    in_flight = []

    def finish(entry):
        preds, meta, st = entry
        ctx = torch.cuda.stream(st) if st is not None else contextlib.nullcontext()
        with ctx:        # the conversion kernels + the copy to the host follow the forward on ITS stream
            return hop_io.export_batch(preds, meta, base, no_convert=bool(params.get('no_convert')),
                                       convert_to_trainid=bool(params.get('convert_to_trainid')),
                                       is_img=bool(params.get('is_img')), save_depth=bool(params.get('save_depth')),
                                       save_depth_as_png=bool(params.get('save_depth_as_png')))
    for k, batch in enumerate(loader):
        if params.get('dry_run'):
            written += dry_write(batch['meta'], base)
            continue
        # K^-1 / E^-1 on the HOST tensors, before the move (same LAPACK bits as the reference's torch.inverse inside predict,
        # pc_transform_model.py:51,71): predict() then never reads a camera back from the device - no stream sync per batch
        inputs = add_camera_inverses(batch['inputs'])
        st = streams[k % depth]
        ctx = torch.cuda.stream(st) if st is not None else contextlib.nullcontext()
        with ctx, torch.no_grad():
            inputs = inputs if params.get('no_gpu') else to_device(inputs)
            preds = models[k % depth].predict(inputs, batch.get('labels'))
        in_flight.append((preds, batch['meta'], st))
        if len(in_flight) >= depth:
            written += finish(in_flight.pop(0))
    while in_flight:
        written += finish(in_flight.pop(0))
    if depth > 1:
        torch.cuda.synchronize()
    # every rank has written its shard before anyone looks for missing frames (export_results :129-165 runs after the
    # loop of a single process; here the loop is spread over the ranks): one barrier per split, taken by ALL ranks
    if pfdist.is_dist():
        torch.distributed.barrier()
    if params.get('is_img'):
        return written
    cs_dir = params.get('data', {}).get('cityscapes_dir')
    if cs_dir is None or (params.get('synthetic') and not params.get('dry_run')):
        print('DID NOT RECEIVE CITYSCAPES DIR. SKIPPING.')
        return written
    if rank == 0:
        gt_dir = os.path.join(cs_dir, 'gtFine', getattr(full_dataset, 'split', split))
        n = hop_io.fill_missing(base, gt_dir, cities=params.get('data', {}).get('cities'),
                                background_dir=getattr(full_dataset, 'background_dir', None),
                                no_convert=bool(params.get('no_convert')))
        print('NUM MISSING: ', n)
    return written


def main(argv=None):
    params = pfconfig.load_config(EXTRA_FLAGS, argv)
    torch.manual_seed(params['seed'])
    dry = bool(params.get('dry_run'))
    if dry and not params.get('synthetic'):
        raise SystemExit('export_bg: --dry_run writes placeholders, it needs --synthetic N')
    rank, world, local = pfdist.init_distributed_mode(backend='gloo' if dry else None)
    if not params.get('no_gpu') and not dry:
        torch.cuda.set_device(local)
    data, collate_fn = build_datasets(params)
    model = None
    if not dry:
        n = 1 if params.get('no_gpu') else max(1, int(params.get('pipeline_depth') or 1))
        model = [build_model(params) for _ in range(n)]       # replicas: own plan + workspace each (weights are 16.5 MB)
        for m in model:
            m.eval()
    written = []
    for split, dataset in data.items():
        written += export_split(model, dataset, split, params, collate_fn)
    if pfdist.is_dist():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return written


if __name__ == '__main__':
    main()
