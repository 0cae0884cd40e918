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


def test_bench_gpus_8_dry_run_starts_eight_ranks_and_gathers_eight_ways():
    """BASELINE configs[4] has 8 ranks; no 8-GPU node has run it yet (SCALE_r0N.json: skipped).  The launcher, the rank
    bookkeeping and the 8-way metric gather run here on gloo so that hardware is not the first place they run:
    8 processes, every rank contributes a different accumulator, rank 0 checks count, order and sum."""
    r, line = _run_bench('--gpus', '8', '--dry-run', env={'OMP_NUM_THREADS': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    assert line['n_gpus'] == 8 and line['gather_ok'] is True and line['dry_run'] is True
    assert len(line['per_rank_ms']) == 8 and len(set(line['devices'])) == 8
    assert '8 rank(s)' in line['sharding']


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


def test_export_driver_two_ranks_shard_then_barrier_then_fill(tmp_path):
    """`export_bg.py --synthetic 5 --dry_run` under two gloo ranks (torch.distributed.run, as scripts/bg/run_export_bg_val.sh would
    be started on a node): the samples are sharded round-robin, every rank writes its own frames, ALL ranks take the barrier
    and only then does rank 0 fill the ground-truth frames nobody predicted (export_cityscapes_segmentation_results.py:129-165).
    Rank 1 is slower by construction: without the barrier rank 0 would count its frames as missing."""
    import glob
    import subprocess
    import numpy as np
    from panoptic_forecasting_amd import hop_io
    gt = tmp_path / 'cs' / 'gtFine' / 'val' / 'synth'
    gt.mkdir(parents=True)
    for i in range(7):                      # 7 ground-truth frames, 5 of them predicted
        hop_io.write_png(str(gt / (hop_io.LABEL_PNG % ('synth', '%06d' % i, 19))), np.zeros((8, 16), np.uint8))
    wd = tmp_path / 'work'
    wd.mkdir()
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'panoptic-forecasting_amd', 'export_bg.py'),
           '--working_dir', str(wd), '--synthetic', '5', '--dry_run', '--no_convert', '--no_gpu',
           '--extra_args', 'data.cityscapes_dir', str(tmp_path / 'cs'), '--extra_args', 'data.data_splits', '[val]',
           '--extra_args', 'task', 'bg_forecast', '--extra_args', 'training.batch_size', '1']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert 'NUM MISSING:  2' in r.stdout, r.stdout[-2000:]
    files = sorted(glob.glob(str(wd / 'exported_predictions' / 'val' / 'synth' / '*.png')))
    assert len(files) == 7, files
    vals = [int(hop_io.read_png(f).max()) for f in files]
    assert vals == [0, 1, 2, 3, 4, 255, 255], vals      # both ranks' placeholders are intact, the two unpredicted frames got the fill value
