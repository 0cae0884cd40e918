"""The N>1 path on CPU: two gloo ranks shard the samples, gather PQ accumulators, and must reproduce the
single-process result (integer TP/FP/FN exactly, sum-IoU to 1e-12) — SURVEY.md 8e / test tier T3."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from panoptic_forecasting_amd import dist as pfdist
    from panoptic_forecasting_amd import pq
    r, w, _ = pfdist.init_distributed_mode(backend='gloo')
    assert (r, w) == (rank, world) and pfdist.is_dist()
    g = torch.Generator().manual_seed(42)
    pred = torch.randint(0, 11, (n_items, 24, 48), generator=g)
    gt = torch.randint(0, 12, (n_items, 24, 48), generator=g)
    gt[gt == 11] = 255
    mine = pfdist.shard_indices(n_items, rank, world)
    acc = pq.pq_accumulate(pred[mine], gt[mine], 11)
    allacc = pfdist.gather_accumulators(acc)
    assert allacc.shape == (world, 11, 4)
    t = pfdist.max_over_ranks(float(rank + 1), torch.device('cpu'))
    assert t == float(world)
    if rank == 0:
        torch.save({'sum': allacc.sum(0), 'single': pq.pq_accumulate(pred, gt, 11)}, out_path)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gather_equals_single_process(tmp_path):
    out = str(tmp_path / 'acc.pt')
    mp.spawn(_worker, args=(2, _free_port(), 7, out), nprocs=2, join=True)
    r = torch.load(out)
    assert torch.equal(r['sum'][:, 1:], r['single'][:, 1:])
    assert (r['sum'][:, 0] - r['single'][:, 0]).abs().max() < 1e-12


def test_shard_indices_partition():
    from panoptic_forecasting_amd import dist as pfdist
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in pfdist.shard_indices(37, r, world))
        assert seen == list(range(37))
