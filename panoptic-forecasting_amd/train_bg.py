#!/usr/bin/env python
"""Training driver for ``task: bg`` with the reference's flags — drop-in for ``experiments/train_model.py`` as launched by
``scripts/bg/run_bg_train.sh``:

    python -u panoptic-forecasting_amd/train_bg.py --config_file configs/bg/bg_train.yaml --working_dir experiments/bg/
    python -u panoptic-forecasting_amd/train_bg.py --continue_training --working_dir experiments/bg/

What it keeps from the reference loop (``training/train.py:66-305``): the ``training.*`` keys (``batch_size``,
``num_epochs``, ``steps_per_epoch``, ``lr``/``mom``/``wd``, ``clip_grad`` | ``clip_grad_norm``, ``accumulate_steps``,
``lr_decay_type`` step|poly with ``lr_decay_factor``/``lr_decay_steps``, ``val_interval``), the epoch structure (train,
validate with the eval-mode loss, keep the best), the files in ``working_dir`` (``config.yaml``, ``model_checkpoint`` and
``best_model`` = bare state_dicts with the reference's 418 keys, ``training_checkpoint`` = {epoch, optimizer,
best_val_result, best_val_epoch, step}) and the per-epoch reseeding (``rank*10000 + epoch``).  What is native: one device
call per micro-batch for forward + loss + backward, one for clip + SGD (``bg_train.BGTrainer``), and under ``torchrun``
ONE all-reduce of the flat gradient per update instead of DDP buckets.

Datasets are outside the hot path: with the reference package importable, ``--dataset reference`` builds
``BGDataset`` through its own ``build_dataset``; ``--synthetic N`` trains on N synthetic Cityscapes-shaped crops per epoch.
"""
import os
import random
import sys
import time

import numpy as np
import torch
import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
if __package__ in (None, ''):                     # run as a script: make the package importable under its alias
    sys.path.insert(0, os.path.dirname(_HERE))
    import panoptic_forecasting_amd  # noqa: F401
    __package__ = 'panoptic_forecasting_amd'

from . import config as pfconfig   # noqa: E402
from . import dist as pfdist       # noqa: E402
from . import synth                # noqa: E402

EXTRA_FLAGS = (
    ('--synthetic', dict(type=int, default=0, help='train on N synthetic crops per epoch instead of a dataset')),
    ('--dataset', dict(default='reference', choices=['reference'])),
)


def learning_rate(tr, epoch):
    """lr in effect DURING ``epoch`` (1-based) — ``train_utils.build_scheduler`` stepped once per finished epoch and
    ``start_epoch`` times up front on resume (``train.py:162-164``): StepLR -> lr * factor^(epoch // steps); the 'poly'
    entry is MultiplicativeLR with factor (1 - e/num_epochs) applied at every step e = 1..epoch."""
    lr = float(tr['lr'])
    kind = tr.get('lr_decay_type')
    if kind == 'step':
        return lr * float(tr.get('lr_decay_factor')) ** (epoch // int(tr.get('lr_decay_steps')))
    if kind == 'poly':
        n = int(tr['num_epochs'])
        for e in range(1, epoch + 1):
            lr *= 1.0 - e / n
        return lr
    return lr


def seed_all(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class SyntheticCrops:
    """``n`` crops per epoch in the batch-dict layout of ``BGDataset`` + ``collate_fn`` (``bg_dataset.py:203-261``)."""

    def __init__(self, n, size, batch, num_classes, rank=0, world=1):
        self.n, self.size, self.batch, self.num_classes, self.rank, self.world = n, size, batch, num_classes, rank, world

    def __len__(self):
        return max(1, self.n // (self.batch * self.world))

    def batches(self, epoch):
        for i in range(len(self)):
            seed = (epoch * 100003 + i) * self.world + self.rank
            inp = synth.make_bg_inputs(b=self.batch, h=self.size, w=self.size, seed=seed, num_classes=self.num_classes)
            g = torch.Generator().manual_seed(7000 + seed)
            lab = torch.randint(0, self.num_classes + 1, (self.batch, self.size // 16, self.size // 16), generator=g)
            lab[lab == self.num_classes] = 255
            lab = torch.nn.functional.interpolate(lab[:, None].float(), size=(self.size, self.size), mode='nearest')[:, 0].long()
            yield {'inputs': inp, 'labels': {'seg': lab}}


class DatasetBatches:
    """A torch Dataset of the reference (``BGDataset``) behind the same ``batches(epoch)`` interface: random batches with
    ``drop_last`` for training, cycled to ``steps_per_epoch * accumulate_steps`` when that key is set
    (``train.py:104-116``), sequential for validation; a DistributedSampler shards both under torchrun."""

    def __init__(self, dataset, params, rank, world, train):
        from torch.utils.data import DataLoader, DistributedSampler, RandomSampler, SequentialSampler
        tr = params['training']
        bs = int(tr.get('batch_size', 1000)) if train else int(tr.get('val_batch_size') or tr.get('batch_size', 1000))
        if world > 1:
            self.sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=train)
        else:
            self.sampler = RandomSampler(dataset) if train else SequentialSampler(dataset)
        workers = int(tr.get('num_data_workers', 0) if train else tr.get('num_val_data_workers', tr.get('num_data_workers', 0)))
        self.loader = DataLoader(dataset, batch_size=bs, sampler=self.sampler, drop_last=train, collate_fn=params.get('collate_fn'),
                                 num_workers=workers)
        steps = tr.get('steps_per_epoch') if train else None
        self.steps = int(steps) * max(1, int(tr.get('accumulate_steps', 1))) if steps else None

    def __len__(self):
        return self.steps or len(self.loader)

    def batches(self, epoch):
        if hasattr(self.sampler, 'set_epoch'):
            self.sampler.set_epoch(epoch)
        if self.steps is None:
            yield from self.loader
            return
        done = 0
        while done < self.steps:
            for batch in self.loader:
                yield batch
                done += 1
                if done >= self.steps:
                    return


def validation_loss(trainer, params, val_loader, epoch):
    """Eval-mode loss over the validation split (``train.py:238-268``): the trained parameters are folded into an inference
    plan (``BGModel``) and every batch goes through ``pf_seg_loss``."""
    from .bg_model import BGModel
    model = BGModel(params)
    model.load_state_dict(trainer.state_dict())
    model.cuda().eval()
    total, n = torch.zeros((), dtype=torch.float64, device='cuda'), 0
    with torch.no_grad():
        for batch in val_loader.batches(epoch):
            batch = to_device(batch)
            total += model.loss(batch['inputs'], batch['labels'])['loss'].double()
            n += 1
    if pfdist.is_dist():
        cnt = torch.tensor([float(n)], dtype=torch.float64, device='cuda')
        torch.distributed.all_reduce(total)
        torch.distributed.all_reduce(cnt)
        n = int(cnt.item())
    return float(total) / max(n, 1)


def to_device(item):
    if isinstance(item, dict):
        return {k: to_device(v) for k, v in item.items()}
    return item.cuda(non_blocking=True) if torch.is_tensor(item) else item


def save_state(path, sd):
    tmp = path + '.tmp'
    torch.save(sd, tmp)
    os.replace(tmp, path)


def main(argv=None):
    params = pfconfig.load_config(EXTRA_FLAGS, argv)
    if params.get('task', 'bg') != 'bg':
        raise SystemExit('train_bg.py trains task: bg (got %r)' % params.get('task'))
    rank, world, _local = pfdist.init_distributed_mode()
    params['distributed'] = world > 1
    tr = params.setdefault('training', {})
    wd_dir = params['working_dir']
    os.makedirs(wd_dir, exist_ok=True)
    if rank == 0 and not params.get('continue_training'):                    # misc.copy_config
        with open(os.path.join(wd_dir, 'config.yaml'), 'w') as f:
            yaml.safe_dump({k: v for k, v in params.items() if isinstance(v, (dict, list, str, int, float, bool, type(None)))}, f)
    seed_all(int(params.get('seed', 1)))

    from .bg_train import BGTrainer
    data = params.setdefault('data', {})
    if params.get('synthetic'):
        data.setdefault('num_classes', 11)
        data.setdefault('depth_norm_params', [20.0, 15.0])
        crop = data.get('crop_size', 800)
        crop = crop if isinstance(crop, int) else int(crop[-1])
        loader = SyntheticCrops(int(params['synthetic']), crop, int(tr.get('batch_size', 8)), data['num_classes'], rank, world)
        val_loader = None
    else:
        try:
            from panoptic_forecasting.data import build_dataset
        except ImportError as e:
            raise SystemExit('the reference package is not importable (%s): pass --synthetic N' % e)
        datasets = build_dataset(params)       # injects data.num_classes / depth_norm_params / collate_fn (bg_dataset.py:62-66)
        loader = DatasetBatches(datasets['train'], params, rank, world, train=True)
        val_loader = DatasetBatches(datasets['val'], params, rank, world, train=False) if 'val' in datasets else None

    trainer = BGTrainer(params)
    ckpt, best_path, train_path = (os.path.join(wd_dir, n) for n in ('model_checkpoint', 'best_model', 'training_checkpoint'))
    start_epoch, best_val, best_epoch = 1, 1e7, -1
    if params.get('continue_training'):
        trainer.load_state_dict(torch.load(ckpt, map_location='cpu'))
        st = torch.load(train_path, map_location='cpu')
        start_epoch, best_val, best_epoch = st['epoch'], st['best_val_result'], st['best_val_epoch']
        trainer.steps = st['step']
        trainer.load_optimizer_state_dict(st['optimizer'])     # torch.optim.SGD's own format (the reference's) or round 2's flat one
    elif params.get('load_model'):
        trainer.load_state_dict(torch.load(params['load_model'], map_location='cpu'))
    else:
        from .bg_model import BGModel
        trainer.load_state_dict(BGModel(params).state_dict())            # the reference's init (+ pretrain_path if given)
    num_epochs = int(tr.get('num_epochs', 100))
    val_interval = int(tr.get('val_interval', 1))
    seed_all(rank * 10000 + start_epoch)
    for epoch in range(start_epoch, num_epochs + 1):
        t0 = time.time()
        lr = learning_rate(tr, epoch)          # the reference steps its scheduler once before epoch 1 (train.py:162-164)
        sums = torch.zeros(2, dtype=torch.float64, device='cuda')
        n_batches = 0
        for batch in loader.batches(epoch):
            batch = to_device(batch)
            out = trainer.train_step(batch['inputs'], batch['labels'], lr=lr)
            sums += torch.stack([out['loss'].double(), out['accuracy'].double()])
            n_batches += 1
        if pfdist.is_dist():
            torch.distributed.all_reduce(sums)
            sums /= world
        train_loss, train_acc = (sums / max(n_batches, 1)).tolist()
        if (epoch + 1) % val_interval != 0:
            continue
        epoch_loss = validation_loss(trainer, params, val_loader, epoch) if val_loader is not None else train_loss   # train.py:236-268
        if rank == 0:
            sd = trainer.state_dict()
            if epoch_loss < best_val:
                best_val, best_epoch = epoch_loss, epoch
                save_state(best_path, sd)
            save_state(ckpt, sd)
            save_state(train_path, {'epoch': epoch + 1, 'optimizer': trainer.optimizer_state_dict(),
                                    'best_val_result': best_val, 'best_val_epoch': best_epoch, 'step': trainer.steps})
            print('EPOCH %d EVAL: train loss %.5f acc %.4f lr %.3g  best %.5f @%d  (%.1f s, %d batches/rank)'
                  % (epoch, train_loss, train_acc, lr, best_val, best_epoch, time.time() - t0, n_batches), flush=True)
        seed_all(rank * 10000 + epoch + 1)


if __name__ == '__main__':
    main()
