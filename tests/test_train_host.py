"""Host side of the training driver (scope row f4): the learning-rate schedule against torch's schedulers as the reference
builds them (training/train_utils.py:13-24, stepped once per epoch + once before the first, train.py:162-164,228-229), and
the flat-gradient exchange on two gloo ranks."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

from panoptic_forecasting_amd import train_bg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('kind', ['step', 'poly', None])
def test_learning_rate_matches_torch_schedulers(kind):
    tr = {'lr': 2e-3, 'lr_decay_type': kind, 'lr_decay_factor': 0.1, 'lr_decay_steps': 3, 'num_epochs': 10}
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=2e-3)
    sch = {'step': lambda: torch.optim.lr_scheduler.StepLR(opt, 3, 0.1),
           'poly': lambda: torch.optim.lr_scheduler.MultiplicativeLR(opt, lambda e: 1 - e / 10),
           None: lambda: None}[kind]()
    for steps_taken in range(0, 9):
        assert abs(opt.param_groups[0]['lr'] - train_bg.learning_rate(tr, steps_taken)) <= 1e-15
        opt.step()
        if sch is not None:
            sch.step()


def test_synthetic_crops_layout_and_sharding():
    a = train_bg.SyntheticCrops(8, 32, 2, 11, rank=0, world=2)
    b = train_bg.SyntheticCrops(8, 32, 2, 11, rank=1, world=2)
    assert len(a) == 2
    ba, bb = next(a.batches(1)), next(b.batches(1))
    assert set(ba) == {'inputs', 'labels'} and set(ba['inputs']) == {'seg', 'depth', 'depth_mask'}
    assert ba['inputs']['seg'].shape == (2, 3, 32, 32) and ba['labels']['seg'].shape == (2, 32, 32)
    assert set(ba['labels']['seg'].unique().tolist()) <= set(range(11)) | {255}
    assert not torch.equal(ba['inputs']['depth'], bb['inputs']['depth'])          # ranks see different samples
    assert torch.equal(ba['inputs']['depth'], next(a.batches(1))['inputs']['depth'])   # and the same ones when re-run


def test_flat_gradient_all_reduce_two_gloo_ranks(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(textwrap.dedent('''
        import sys, torch
        sys.path.insert(0, %r)
        from panoptic_forecasting_amd import dist as pfdist
        rank, world, _ = pfdist.init_distributed_mode(backend='gloo')
        g = torch.arange(10, dtype=torch.float32) * (rank + 1)
        pfdist.all_reduce_mean_(g)
        assert torch.equal(g, torch.arange(10, dtype=torch.float32) * 1.5), g
        print('ok', rank)
    ''' % ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29631')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29631', str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count('ok') == 2


def test_reference_parameter_order_matches_fixture():
    """bg_train.reference_parameter_order against the reference's own ``model.parameters()`` order, recorded by
    tests/golden/make_golden_train.py (``keys`` of fixture G6): index i of a torch.optim.SGD state_dict is that order."""
    import numpy as np
    from conftest import GOLDEN
    from panoptic_forecasting_amd import bg_train, hardnet_arch as arch
    want = [str(k) for k in np.load(os.path.join(GOLDEN, 'g6_train_64x128.npz'))['keys']]
    layout, _ = bg_train.param_layout(arch.Spec(36, 11))
    mine = [key for key, _, _, trainable in layout if trainable]
    assert sorted(mine) == sorted(want)
    assert mine != want                      # the op-table order differs (conv1x1_up.i before denseBlocksUp.i) ...
    assert bg_train.reference_parameter_order(mine) == want   # ... the checkpoint order is the reference's
