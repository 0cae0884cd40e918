"""One rank of tests/test_gpu_train.py::test_bgmodel_under_distributed_data_parallel (test infrastructure, not product code).

The reference trains data-parallel by wrapping the model it got from the registry in ``DistributedDataParallel(DistWrapper(model))``
(training/train.py:96-103, models/dist_wrapper.py:13-26) and running its ordinary loop body on it (train.py:185-216).  This
worker does exactly that with THIS package's registry model - a DistWrapper-shaped module (forward = ``model.loss``), DDP's
reducer hooks on the parameters, ``loss.backward(); clip_grad_norm_; opt.step(); opt.zero_grad()`` - for two steps, and rank 0
saves the averaged gradients of the first step and the parameters after the second.

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment;  python tests/ddp_worker.py OUT.pt
"""
import json
import os
import sys

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


class DistWrapper(nn.Module):
    """Shape of the reference's wrapper (models/dist_wrapper.py:13-26): forward(inputs, labels) = model.loss(inputs, labels)."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, inputs, labels):
        return self.model.loss(inputs, labels)


def micro_batches(step, rank, h=64, w=128, b=2):
    """Deterministic micro-batch of (step, rank): the test process rebuilds the same ones."""
    from panoptic_forecasting_amd import synth
    seed = 100 + 10 * step + rank
    inputs = synth.make_bg_inputs(b=b, h=h, w=w, seed=seed)
    g = torch.Generator().manual_seed(5000 + seed)
    lab = torch.randint(0, 12, (b, h // 8, w // 8), generator=g)
    lab[lab == 11] = 255
    labels = {'seg': nn.functional.interpolate(lab[:, None].float(), size=(h, w), mode='nearest')[:, 0].long()}
    return inputs, labels


def params():
    return {'task': 'bg', 'no_gpu': False, 'load_model': None, 'load_best_model': False,
            'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])]},
            'model': {'model_type': 'bg', 'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True},
            'training': {'lr': 2e-3, 'mom': 0.9, 'wd': 1e-4, 'clip_grad_norm': 5.0}}


def state_dict():
    from panoptic_forecasting_amd import synth
    with open(os.path.join(ROOT, 'tests', 'golden', 'calib_seed1234.json')) as f:
        return synth.make_state_dict(seed=1234, calib=json.load(f))


def main():
    import contextlib
    import torch.distributed as dist
    from panoptic_forecasting_amd.registry import build_model
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)      # two ranks share the one GPU: RCCL refuses that, gloo does not
    torch.cuda.set_device(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = build_model(params())
    model.load_state_dict(state_dict())
    model.cuda()
    ddp = torch.nn.parallel.DistributedDataParallel(DistWrapper(model), device_ids=[0])      # train.py:99
    tp = params()['training']
    opt = torch.optim.SGD([p for p in ddp.parameters() if p.requires_grad], lr=tp['lr'], momentum=tp['mom'], weight_decay=tp['wd'])
    saved = {}
    for step in range(2):
        ddp.train()
        inputs, labels = micro_batches(step, rank)
        inputs = {k: v.cuda() for k, v in inputs.items()}
        labels = {k: v.cuda() for k, v in labels.items()}
        loss_dict = ddp(inputs, labels)                       # train.py:192
        loss = loss_dict['loss'].mean() / 1
        loss.backward()                                       # DDP's reducer averages the gradients over the ranks here
        if step == 0:
            saved['grads'] = {k: p.grad.detach().cpu().clone() for k, p in ddp.module.model.named_parameters() if p.grad is not None}
            saved['loss0'] = float(loss)
        nn.utils.clip_grad_norm_(ddp.parameters(), tp['clip_grad_norm'])
        opt.step()
        opt.zero_grad()
        if step == 0:
            saved['params_step1'] = {k: p.detach().cpu().clone() for k, p in ddp.module.model.named_parameters()}
    saved['params'] = {k: p.detach().cpu().clone() for k, p in ddp.module.model.named_parameters()}
    gathered = [None] * world
    dist.all_gather_object(gathered, {k: float(v.double().norm()) for k, v in saved['params'].items()})
    if rank == 0:
        saved['param_norms_by_rank'] = gathered
        torch.save(saved, sys.argv[1])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
