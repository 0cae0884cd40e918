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


def _run_bench(*flags, env=None):
    import json
    import subprocess
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(flags), capture_output=True, text=True,
                       timeout=600, env=e)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` from a plain shell must become the launcher of 2 ranks (it used to run 1 rank and print
    n_gpus: 1); --dry-run does the rendezvous + the sharded metric gather on gloo and no GPU work."""
    r, line = _run_bench('--gpus', '2', '--dry-run')
    assert r.returncode == 0, r.stderr[-2000:]
    assert line['n_gpus'] == 2 and line['gather_ok'] is True and line['dry_run'] is True


def test_bench_refuses_a_world_that_differs_from_gpus():
    """Launched with WORLD_SIZE=1 semantics but --gpus 2 in a launcher environment of another size: error, not a silent
    1-rank run."""
    r, line = _run_bench('--gpus', '2', '--dry-run', env={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and line is None
    assert 'rank(s) joined' in r.stderr


def test_bench_refuses_more_ranks_than_gpus():
    """Without --dry-run the launcher checks the node's GPU count first (0 in the CPU container)."""
    import torch as _t
    if _t.cuda.is_available() and _t.cuda.device_count() >= 2:
        return
    r, line = _run_bench('--gpus', '2')
    assert r.returncode != 0 and line is None
    assert 'GPU(s)' in r.stderr
