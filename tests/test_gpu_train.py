"""Scope row f4 on the device: ``pf_train_forward_backward`` / ``pf_sgd_step`` (through ``bg_train.BGTrainer``) and the
autograd drop-in (``BGModel.loss`` in training mode) against the oracle's training step and against the reference's own
two-batch run (fixture g6_train_64x128.npz).

Tolerances: kernel-level parity (every op kind of the training path, a HarDBlock-shaped mini network, float64 autograd as
the checker) is 1e-4 relative L2 per gradient tensor, 1e-5 on the loss.  The whole 70-layer network is compared with
bars derived from fp32's own conditioning on this problem (see ``_oracle_grads``)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import hardnet_ref
from panoptic_forecasting_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
LOSS_REL = 1e-4
# parameters after two clipped SGD steps: (relative size of the update, ~1e-2) x (gradient conditioning, _oracle_grads)
POST_REL = 2e-3


def _params(**training):
    tr = {'lr': 2e-3, 'mom': 0.9, 'wd': 1e-4, 'clip_grad_norm': 5.0}
    tr.update(training)
    return {'task': 'bg', 'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])]},
            'model': {'model_type': 'bg', 'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True}, 'training': tr}


def _sd():
    with open(os.path.join(G, 'calib_seed1234.json')) as f:
        return synth.make_state_dict(seed=1234, calib=json.load(f))


def _fixture():
    z = np.load(os.path.join(G, 'g6_train_64x128.npz'))
    batches = []
    for s in range(2):
        batches.append(({'seg': torch.from_numpy(z['seg'][s]).long(), 'depth': torch.from_numpy(z['depth'][s]),
                         'depth_mask': torch.from_numpy(z['mask'][s])}, {'seg': torch.from_numpy(z['labels'][s]).long()}))
    return z, batches


def _cuda(d):
    return {k: v.cuda() for k, v in d.items()}


def _labels(b, h, w, seed):
    g = torch.Generator().manual_seed(4000 + seed)
    lab = torch.randint(0, 12, (b, max(h // 8, 1), max(w // 8, 1)), generator=g)
    lab[lab == 11] = 255
    return torch.nn.functional.interpolate(lab[:, None].float(), size=(h, w), mode='nearest')[:, 0].long()


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-12))


def _record_grad_distances(tag, hip, aten):
    """gpurun_out/r03_train_grad_dist.json: per tensor, rel. L2 distance HIP <-> float64 and fp32-ATen <-> float64 (the bars in
    tests/golden/train_grad_bars.json are 1.5x the former, measured on MI355X)"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r03_train_grad_dist.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = {}
    if os.path.exists(path):
        with open(path) as f:
            data = json.load(f)
    data[tag] = {'hip_vs_f64': hip, 'aten_f32_vs_f64': aten, 'max_hip': max(hip.values()), 'max_aten': max(aten.values())}
    with open(path, 'w') as f:
        json.dump(data, f, indent=0, sort_keys=True)
    print('gradient distance to float64 (rel. L2), %s: HIP max %.3e (median %.3e), fp32 ATen max %.3e' % (
        tag, max(hip.values()), sorted(hip.values())[len(hip) // 2], max(aten.values())))


def _oracle_grads(sd, inputs, labels, tag):
    """(fp64 gradients, fp32 result, per-key bar).  The gradient of this 70-layer ReLU network is ill-conditioned in fp32:
    forward round-off (1e-7 after the first layer) grows ~1.3x per layer to 3e-5..1e-4 at the output, every pre-activation
    closer to zero than that flips its ReLU mask between two fp32 implementations, and a flipped element changes the
    gradient by its full value - relative gradient error ~ sqrt(relative forward error).  The reference's own ATen fp32
    gradients are 0.7 % (median) .. 1.5-4 % (worst tensor) away from the float64 gradients at these sizes (rel. L2), and so
    are the HIP path's (profiles/r03_train_grad_dist.json: HIP median 7.5e-3 / max 1.7e-2, ATen 6.9e-3 / 1.5e-2 at
    128x256).  The bar per tensor is 1.5x the distance HIP <-> float64 MEASURED on MI355X for that tensor
    (tests/golden/train_grad_bars.json, floor 1e-4 = the kernel-level bar of test_mini_network_training_step_vs_autograd);
    the measured distances of every run are printed and written to gpurun_out/r03_train_grad_dist.json."""
    r32 = hardnet_ref.bg_train_step({k: v.clone() for k, v in sd.items()}, inputs, labels, clip_grad_norm=None, apply_update=False)
    sd64 = {k: (v.double() if v.dtype == torch.float32 else v.clone()) for k, v in sd.items()}
    in64 = dict(inputs)
    in64['depth'] = inputs['depth'].double()
    r64 = hardnet_ref.bg_train_step(sd64, in64, labels, clip_grad_norm=None, apply_update=False)
    with open(os.path.join(G, 'train_grad_bars.json')) as f:
        bars = json.load(f)['bars'][tag]
    return r64['grads'], r32, bars, sd64


@pytest.mark.parametrize('size', [(64, 128), (128, 256)])
def test_forward_backward_vs_oracle(size):
    from panoptic_forecasting_amd.bg_train import BGTrainer
    h, w = size
    sd = _sd()
    if size == (64, 128):
        z, batches = _fixture()
        inputs, labels = batches[0]
    else:
        inputs = synth.make_bg_inputs(b=2, h=h, w=w, seed=21)
        labels = {'seg': _labels(2, h, w, 5)}
    tr = BGTrainer(_params())
    tr.load_state_dict(sd)
    out = tr.forward_backward(_cuda(inputs), _cuda(labels))
    g64, ref, bars, sd64 = _oracle_grads(sd, inputs, labels, '%dx%d' % size)
    assert abs(float(out['loss']) - float(ref['loss'])) <= LOSS_REL * abs(float(ref['loss']))
    assert abs(float(out['accuracy']) - float(ref['accuracy'])) <= 2e-4        # an argmax near-tie may flip a pixel
    got = tr.named_grads()
    dist = {k: _rel(got[k].cpu(), g) for k, g in g64.items()}
    _record_grad_distances('%dx%d' % size, dist, {k: _rel(ref['grads'][k], g) for k, g in g64.items()})
    aten = {k: _rel(ref['grads'][k], g) for k, g in g64.items()}
    aten_med = sorted(aten.values())[len(aten) // 2]
    for k, g in g64.items():
        # two criteria.  The correctness criterion is INDEPENDENT of this implementation: the HIP gradient may be at most twice as
        # far from float64 as the reference's own fp32 (ATen) gradient of that tensor is - or as ATen's median tensor, for the
        # few tensors where ATen happens to land within 1e-5 (measured: HIP / max(ATen, median) <= 0.95 at both sizes,
        # profiles/r03_train_grad_dist.json).  The per-tensor bars measured on this implementation (train_grad_bars.json)
        # stay as a regression tripwire
        assert dist[k] <= max(2.0 * max(aten[k], aten_med), 1e-4), (k, dist[k], aten[k], aten_med)
        assert dist[k] <= bars[k], (k, dist[k], bars[k])
    # the last block is well conditioned at the level of the head: tight bar there
    for k in ('model.finalConv.weight', 'model.finalConv.bias', 'model.denseBlocksUp.3.layers.3.norm.weight'):
        assert _rel(got[k].cpu(), g64[k]) <= 2e-4, k
    # running statistics were updated in theta (momentum 0.1, unbiased variance); sd64 holds the float64 oracle's
    post = tr.state_dict()
    for k in sd64:
        if k.endswith('running_mean') or k.endswith('running_var'):
            assert torch.allclose(post[k].double(), sd64[k], rtol=2e-3, atol=1e-4), k
    if size == (64, 128):      # and against the reference's own numbers (clipped by 5 / (norm + 1e-6))
        assert abs(float(out['loss']) - z['loss'][0]) <= LOSS_REL * z['loss'][0]
        coef = min(1.0, 5.0 / (z['grad_norm'][0] + 1e-6))
        for name in z.files:
            if name.startswith('grad::'):
                # the fixture is the reference's own fp32 (ATen) gradient: two fp32 implementations, each within its bar of float64
                assert _rel(got[name[6:]].cpu() * coef, torch.from_numpy(z[name])) <= bars[name[6:]] + 1.5 * _rel(ref['grads'][name[6:]], g64[name[6:]]), name


def test_timed_configuration_800x800_vs_oracle():
    """The configuration tools/bench_train.py and bench.py's `train_step` TIME - 800x800 crops (configs/bg/bg_train.yaml:25,48) -
    against the oracle, with the kernels of that configuration: csrc/train_tuned.inc is keyed on the timed shapes at batch 8, so
    the table is consulted as for batch 8 (option train_table_batch) while the batch itself is 2, which the CPU oracle (fp32 and
    float64 autograd of the 70-layer network) can afford.  At 800x800 the 50- and 25-pixel levels have odd widths (rows padded
    to 52 / 28 floats): conv + BatchNorm outputs kept in padded rows, one backward-data conv over all input ranges, the padded
    weight-gradient inputs.  pf_train_path_stats proves those forms and the table's rows ran (and no generic-kernel fallback).
    Criterion: the one of test_forward_backward_vs_oracle that is independent of this implementation - the distance to float64 of
    every gradient tensor RELATIVE to the distance of the reference's own fp32 (ATen) gradient of that tensor (or of ATen's median
    tensor) - with the SAME factor as at the small sizes, 2.  Round 5 needed 3 here (worst tensor 2.95x): the gradient error of this
    ReLU network follows the forward pass's round-off (masks flip), and at 800x800 the HIP forward was 1.2-1.5x as far from float64
    as torch-CPU fp32 - large grids run the unsplit workgroup shapes, ONE fp32 FMA chain over all 9 * Cin terms per output, where
    ATen's blocked GEMM adds shorter partial chains.  Round 6: the 3x3 convolutions of a training step add every round of 8 input
    channels into a second accumulator set (conv_dma.hip KACC, option train_blocked_sum): the forward is now 0.79-0.95x ATen's
    distance at every block output (tools/train_fwd_error.py, profiles/r06_experiments.md; +0.17 ms per step).
    Loss 1e-4, head-level tensors 2e-4, running statistics as at the small sizes."""
    from panoptic_forecasting_amd import lib as pflib
    from panoptic_forecasting_amd.bg_train import BGTrainer
    h = w = 800
    b = 2
    sd = _sd()
    inputs = synth.make_bg_inputs(b=b, h=h, w=w, seed=31)
    labels = {'seg': _labels(b, h, w, 7)}
    L = pflib.load()
    pflib.check(L.pf_set_option(b'train_table_batch', 8), 'pf_set_option')
    try:
        tr = BGTrainer(_params())
        tr.load_state_dict(sd)
        out = tr.forward_backward(_cuda(inputs), _cuda(labels))
        torch.cuda.synchronize()
        stats = tr.path_stats()
    finally:
        L.pf_set_option(b'train_table_batch', 0)
    print('training path at 800x800:', stats)
    assert stats['table_shapes'] + stats['packed_pair_forward_convs'] >= 60, stats   # rows of csrc/train_tuned.inc (73 of the step's 126 geometries have one) or conv_s4 (forward)
    assert stats['padded_output_layers'] >= 20, stats         # conv + BatchNorm layers of the 50- / 25-pixel levels, forward
    assert stats['single_backward_data_convs'] >= 20, stats   # ... and their input gradients, one conv per layer
    assert stats['generic_kernel_launches'] == 0, stats
    r32 = hardnet_ref.bg_train_step({k: v.clone() for k, v in sd.items()}, inputs, labels, clip_grad_norm=None, apply_update=False)
    sd64 = {k: (v.double() if v.dtype == torch.float32 else v.clone()) for k, v in sd.items()}
    in64 = dict(inputs)
    in64['depth'] = inputs['depth'].double()
    r64 = hardnet_ref.bg_train_step(sd64, in64, labels, clip_grad_norm=None, apply_update=False)
    assert abs(float(out['loss']) - float(r32['loss'])) <= LOSS_REL * abs(float(r32['loss']))
    assert abs(float(out['accuracy']) - float(r32['accuracy'])) <= 2e-4
    got = tr.named_grads()
    g64 = r64['grads']
    dist = {k: _rel(got[k].cpu(), g) for k, g in g64.items()}
    aten = {k: _rel(r32['grads'][k], g) for k, g in g64.items()}
    _record_grad_distances('800x800', dist, aten)
    aten_med = sorted(aten.values())[len(aten) // 2]
    for k in g64:
        assert dist[k] <= max(2.0 * max(aten[k], aten_med), 1e-4), (k, dist[k], aten[k], aten_med)
    med = sorted(dist.values())[len(dist) // 2]
    assert med <= 1.5 * aten_med, (med, aten_med)
    worst = max(dist[k] / max(aten[k], aten_med) for k in g64)
    print('800x800 gradient distances: median %.2f x ATen median, worst tensor %.2f x' % (med / aten_med, worst))
    for k in ('model.finalConv.weight', 'model.finalConv.bias', 'model.denseBlocksUp.3.layers.3.norm.weight'):
        assert dist[k] <= 2e-4, (k, dist[k])
    post = tr.state_dict()
    for k in sd64:
        if k.endswith('running_mean') or k.endswith('running_var'):
            assert torch.allclose(post[k].double(), sd64[k], rtol=2e-3, atol=1e-4), k


def test_two_training_steps_vs_reference_fixture():
    """train_step x2 with the values of configs/bg/bg_train.yaml: parameters, momentum and running statistics after the
    second step against the reference loop's (the clip / weight decay / momentum arithmetic of pf_sgd_step included)."""
    from panoptic_forecasting_amd.bg_train import BGTrainer
    z, batches = _fixture()
    tr = BGTrainer(_params())
    tr.load_state_dict(_sd())
    for s, (inputs, labels) in enumerate(batches):
        out = tr.train_step(_cuda(inputs), _cuda(labels))
        assert abs(float(out['loss']) - z['loss'][s]) <= (LOSS_REL if s == 0 else 1e-3) * z['loss'][s]
    post = tr.state_dict()
    keys = [str(k) for k in z['post_keys']]
    l2 = np.array([float(post[k].double().norm()) for k in keys])
    assert np.all(np.abs(l2 - z['post_l2']) <= POST_REL * z['post_l2'] + 1e-6)
    for name in z.files:
        if name.startswith('post::'):
            ref = torch.from_numpy(z[name])
            assert _rel(post[name[6:]], ref) <= POST_REL, name
    assert int(post['model.base.0.norm.num_batches_tracked']) == 2


def test_gradient_accumulation_and_loss_scale():
    from panoptic_forecasting_amd.bg_train import BGTrainer
    z, batches = _fixture()
    tr = BGTrainer(_params())
    tr.load_state_dict(_sd())
    a_in, a_lab = (_cuda(d) for d in batches[0])
    b_in, b_lab = (_cuda(d) for d in batches[1])
    tr.forward_backward(a_in, a_lab, update_running_stats=False)
    ga = tr.grad.clone()
    tr.forward_backward(b_in, b_lab, update_running_stats=False)
    gb = tr.grad.clone()
    tr.forward_backward(a_in, a_lab, update_running_stats=False, loss_scale=0.5)
    tr.forward_backward(b_in, b_lab, accumulate=True, update_running_stats=False, loss_scale=0.5)
    assert _rel(tr.grad, 0.5 * (ga + gb)) <= 1e-5
    # bit-reproducible: every reduction runs in a fixed order
    tr.forward_backward(a_in, a_lab, update_running_stats=False)
    assert torch.equal(tr.grad, ga)


@pytest.mark.parametrize('graph', [False, True])
def test_weight_gradient_stream_and_graph_replays_change_no_bit(graph):
    """training.weight_gradient_stream (library option train_side_stream: the weight gradients on the plan's own stream, forked
    and joined inside the call) against everything on the caller's stream; eagerly, and captured + replayed four times."""
    from panoptic_forecasting_amd.bg_train import BGTrainer
    z, batches = _fixture()
    a_in, a_lab = (_cuda(d) for d in batches[0])
    got = {}
    for side in (False, True):
        tr = BGTrainer(_params(use_hip_graph=graph, weight_gradient_stream=side))
        assert tr.side_stream == side and tr.use_graph == graph
        tr.load_state_dict(_sd())
        res = []
        for _ in range(6):      # with the graph: eager, capture + replay, 4 more replays
            tr.grad.fill_(3.0)
            out = tr.forward_backward(a_in, a_lab, update_running_stats=False)
            torch.cuda.synchronize()
            res.append((tr.grad.clone(), float(out['loss'])))
        for g, l in res[1:]:
            assert torch.equal(g, res[0][0]) and l == res[0][1]
        got[side] = res[0]
    assert torch.equal(got[False][0], got[True][0]) and got[False][1] == got[True][1]


def test_two_pass_bilinear_transpose_is_the_one_pass_sum():
    """The loss head's transposed 4x interpolation runs as a row pass + a column pass once its planes are large (here 2 x 11 x
    512 x 512 outputs); option upsample_bwd_two_pass = 0 keeps the one-pass gather.  Same nesting, same order: same bits."""
    from panoptic_forecasting_amd import lib as pflib, synth
    from panoptic_forecasting_amd.bg_train import BGTrainer
    L = pflib.load()
    inp = {k: v.cuda() for k, v in synth.make_bg_inputs(b=2, h=512, w=512, seed=5).items()}
    gen = torch.Generator().manual_seed(9)
    lab = torch.randint(0, 12, (2, 512, 512), generator=gen)
    lab[lab == 11] = 255
    lab = {'seg': lab.cuda()}
    got = {}
    for two in (1, 0):
        pflib.check(L.pf_set_option(b'upsample_bwd_two_pass', two), 'pf_set_option')
        try:
            tr = BGTrainer(_params())
            tr.load_state_dict(_sd())
            out = tr.forward_backward(inp, lab, update_running_stats=False)
            torch.cuda.synchronize()
            got[two] = (tr.grad.clone(), float(out['loss']))
        finally:
            L.pf_set_option(b'upsample_bwd_two_pass', 1)
    assert got[1][1] == got[0][1] and torch.isfinite(got[1][0]).all() and float(got[1][0].abs().max()) > 0
    assert torch.equal(got[1][0], got[0][0])


def test_measured_conv_shapes_change_rounding_only():
    """pf_train_autotune (training.autotune): every forward / backward-data convolution's workgroup shape measured on first sight.
    Same function, other tilings: gradients agree to rounding with the cost-model shapes; the choices can be read back, and a
    second step with them is bit-identical to the first."""
    import ctypes
    from panoptic_forecasting_amd import lib as pflib
    from panoptic_forecasting_amd.bg_train import BGTrainer
    z, batches = _fixture()
    a_in, a_lab = (_cuda(d) for d in batches[0])
    L = pflib.load()
    pflib.check(L.pf_set_option(b'use_tuned_table', 0), 'pf_set_option')     # the reference point: the cost model alone
    try:
        ref = BGTrainer(_params())
        ref.load_state_dict(_sd())
        want = ref.forward_backward(a_in, a_lab, update_running_stats=False)
        g_ref = ref.grad.clone()
    finally:
        L.pf_set_option(b'use_tuned_table', 1)
    tr = BGTrainer(_params(autotune=True))
    tr.load_state_dict(_sd())
    got = tr.forward_backward(a_in, a_lab, update_running_stats=False)      # measures, then runs with the measured shapes
    g1 = tr.grad.clone()
    n = ctypes.c_int()
    pflib.check(L.pf_train_tuned_shapes(tr._t, None, 0, ctypes.byref(n)), 'pf_train_tuned_shapes')
    assert n.value >= 10
    rows = (ctypes.c_int * (10 * n.value))()
    pflib.check(L.pf_train_tuned_shapes(tr._t, rows, n.value, ctypes.byref(n)), 'pf_train_tuned_shapes')
    for i in range(n.value):
        ks, stride, cin, cout, h, w, b, accum, wm, nt = rows[i * 10:i * 10 + 10]
        assert ks in (1, 3) and stride in (1, 2) and b == a_in['seg'].shape[0] and wm in (0, 1, 2, 4) and 0 <= nt <= 4 and accum in (0, 1)
    assert abs(float(got['loss']) - float(want['loss'])) <= 1e-6 * abs(float(want['loss']))
    assert _rel(g1, g_ref) <= 1e-5
    tr.forward_backward(a_in, a_lab, update_running_stats=False)
    assert torch.equal(tr.grad, g1)


@pytest.mark.parametrize('clip', ['norm', 'value', 'none'])
def test_sgd_step_vs_torch(clip):
    import ctypes
    from panoptic_forecasting_amd import lib as _lib
    L = _lib.load()
    n = 100003
    g = torch.Generator().manual_seed(3)
    theta = torch.randn(n, generator=g)
    mask = torch.rand(n, generator=g) < 0.9
    need = ctypes.c_size_t()
    _lib.check(L.pf_sgd_workspace(ctypes.byref(need)), 'pf_sgd_workspace')
    ws = torch.empty(need.value, dtype=torch.uint8, device='cuda')
    p_ref = torch.nn.Parameter(theta[mask].clone())
    opt = torch.optim.SGD([p_ref], lr=0.05, momentum=0.9, weight_decay=1e-2)
    d_theta, d_mom = theta.cuda(), torch.zeros(n, device='cuda')
    d_mask = mask.to(torch.uint8).cuda()
    for step in range(3):
        grad = torch.randn(n, generator=g) * 3
        p_ref.grad = grad[mask].clone()
        if clip == 'norm':
            torch.nn.utils.clip_grad_norm_([p_ref], 5.0)
        elif clip == 'value':
            torch.nn.utils.clip_grad_value_([p_ref], 0.7)
        opt.step()
        d_grad = grad.cuda()
        _lib.check(L.pf_sgd_step(d_theta.data_ptr(), d_grad.data_ptr(), d_mom.data_ptr(), d_mask.data_ptr(), n, 0.05, 0.9, 1e-2,
                                 5.0 if clip == 'norm' else 0.0, 0.7 if clip == 'value' else 0.0, int(step == 0), ws.data_ptr(),
                                 ws.numel(), _lib.stream_ptr()), 'pf_sgd_step')
        assert torch.allclose(d_theta.cpu()[mask], p_ref.detach(), rtol=1e-5, atol=1e-6)
        assert torch.equal(d_theta.cpu()[~mask], theta[~mask])            # buffers (running statistics) are not stepped


def test_bgmodel_training_loss_is_a_drop_in_for_the_reference_loop():
    """train.py:186-210 verbatim on the registry's model: model.train(); loss = model.loss(...)['loss']; loss.backward();
    clip_grad_norm_; torch.optim.SGD.step() — then eval-mode predict on the updated parameters."""
    from panoptic_forecasting_amd.registry import build_model
    z, batches = _fixture()
    sd = _sd()
    m = build_model(_params())
    m.load_state_dict(sd)
    m.cuda()
    model_params = [p for p in m.parameters() if p.requires_grad]
    assert len(model_params) == len(hardnet_ref.trainable_keys(sd))
    opt = torch.optim.SGD(model_params, lr=2e-3, weight_decay=1e-4, momentum=0.9)
    osd = {k: v.clone() for k, v in sd.items()}
    bufs = None
    for s, (inputs, labels) in enumerate(batches):
        m.train()
        res = m.loss(_cuda(inputs), _cuda(labels))
        loss = res['loss'].mean() / 1
        loss.backward()
        total = torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
        if s == 0:
            assert abs(float(total) - z['grad_norm'][0]) <= 0.1 * z['grad_norm'][0]      # see _oracle_grads on conditioning
            grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
            g64, r32, bars, _ = _oracle_grads(sd, inputs, labels, '64x128')
            # both sides hold CLIPPED gradients, each scaled by its own 5 / (total norm + 1e-6) - and the total norm is a sum over
            # the ill-conditioned encoder tensors too (it may differ by a few per cent): compare the gradients before clipping
            coef_ref = min(1.0, 5.0 / (z['grad_norm'][0] + 1e-6)), min(1.0, 5.0 / (float(total) + 1e-6))
            for name in z.files:
                if name.startswith('grad::'):
                    k = name[6:]
                    d = _rel(grads[k].cpu() / coef_ref[1], torch.from_numpy(z[name]) / coef_ref[0])
                    assert d <= bars[k] + 1.5 * _rel(r32['grads'][k], g64[k]), (name, d)
        opt.step()
        opt.zero_grad()
        ref = hardnet_ref.bg_train_step(osd, inputs, labels, momentum_bufs=bufs)
        bufs = ref['momentum_bufs']
        assert abs(float(res['loss']) - float(ref['loss'])) <= 1e-3 * float(ref['loss'])
    post = m.state_dict()
    assert set(post.keys()) == set(sd.keys())
    for name in z.files:
        if name.startswith('post::'):
            assert _rel(post[name[6:]].cpu(), torch.from_numpy(z[name])) <= POST_REL, name
    # the inference plan is rebuilt from the trained parameters
    m.eval()
    inp = synth.make_bg_inputs(b=1, h=64, w=128, seed=3)
    out = m.predict(_cuda(inp), None)
    want = hardnet_ref.bg_predict({k: v.cpu() for k, v in post.items()}, inp)
    assert (out['orig_size_logits'].cpu() - want['orig_size_logits']).abs().max() <= 1e-3


def _mini_net():
    """Every op kind of the training path in one small, well-conditioned network: stride-2 stem, 3x3 convs with one and
    two input ranges, a conv writing into a slice of a wider tensor, 1x1 conv, pool, upsample + 1x1 over [up, skip],
    plain final conv, bilinear head."""
    from panoptic_forecasting_amd import hardnet_arch as arch
    from tests.helpers import MiniSpec
    sp = MiniSpec(6)
    S = arch.Src
    c0 = sp.conv('c0', [S(0, 0, 6)], 16, 3, stride=2, bn=True)
    c1 = sp.conv('c1', [S(c0, 0, 16)], 24, 3, bn=True)
    cat = sp.tensor('cat', 28 + 10 + 12)
    sp.conv('c2', [S(c1, 0, 24), S(c0, 0, 16)], 28, 3, dst=cat, dst_choff=10, bn=True)
    sp.conv('c2b', [S(c1, 4, 18)], 10, 3, dst=cat, dst_choff=0, bn=True)
    # like the last layer of a HarDBlock: three ranges, the first one a slice of the tensor it writes into
    sp.conv('c2c', [S(cat, 0, 10), S(c1, 0, 24), S(c0, 0, 16)], 12, 3, dst=cat, dst_choff=38, bn=True)
    c3 = sp.conv('c3', [S(cat, 0, 50)], 20, 1, bn=True)
    p = sp.pool('p', c3)
    c4 = sp.conv('c4', [S(p, 0, 20)], 34, 3, bn=True)
    up = sp.upsample('up', c4, c3)
    c5 = sp.conv('c5', [S(up, 0, 34), S(c3, 0, 20)], 22, 1, bn=True)
    fin = sp.conv('fin', [S(c5, 0, 22), S(cat, 10, 28)], 11, 1, relu=False, bn=False)
    sp.head(fin)
    return sp


def _mini_torch(sp, params, x, labels):
    """The same op table through torch autograd (float64 checker), with gradients of every tensor retained."""
    from panoptic_forecasting_amd import hardnet_arch as arch
    import torch.nn.functional as F
    leaves = {n: {k: v.double().clone().requires_grad_(k in ('w', 'gamma', 'beta', 'b')) for k, v in pr.items()} for n, pr in params.items()}
    parts = {}           # tensor index -> {choff: produced slice}
    whole = {0: x.double()}
    kept = {}

    def get(src):
        if src.tensor in whole:
            return whole[src.tensor][:, src.choff:src.choff + src.ch]
        pr = parts[src.tensor]
        if src.choff in pr and pr[src.choff].shape[1] == src.ch and sum(p.shape[1] for p in pr.values()) < sp.tensors[src.tensor].channels:
            return pr[src.choff]                       # a slice read while the tensor is still being filled
        t = torch.cat([pr[k] for k in sorted(pr)], 1)
        assert t.shape[1] == sp.tensors[src.tensor].channels
        whole[src.tensor] = t
        return t[:, src.choff:src.choff + src.ch]

    logits = None
    for op in sp.ops:
        if op.kind in (arch.OP_STEM, arch.OP_CONV):
            xin = torch.cat([get(s) for s in op.srcs], 1)
            pr = leaves[op.name]
            y = F.conv2d(xin, pr['w'], None if op.bn else pr['b'], stride=op.stride, padding=op.k // 2)
            if op.bn:
                y = F.batch_norm(y, pr['mean'], pr['var'], pr['gamma'], pr['beta'], training=True, momentum=0.1, eps=1e-5)
            if op.relu:
                y = F.relu(y)
            if op.cout == sp.tensors[op.dst].channels:
                y.retain_grad()
                whole[op.dst] = y
                kept[sp.tensors[op.dst].name] = y
            else:
                y.retain_grad()
                parts.setdefault(op.dst, {})[op.dst_choff] = y
                kept['%s@%d' % (sp.tensors[op.dst].name, op.dst_choff)] = y
        elif op.kind == arch.OP_POOL:
            whole[op.dst] = F.avg_pool2d(get(op.srcs[0]), 2, 2)
            whole[op.dst].retain_grad()
            kept[sp.tensors[op.dst].name] = whole[op.dst]
        elif op.kind == arch.OP_UPSAMPLE:
            like = get(op.srcs[1])
            whole[op.dst] = F.interpolate(get(op.srcs[0]), size=like.shape[-2:], mode='bilinear', align_corners=True)
            whole[op.dst].retain_grad()
            kept[sp.tensors[op.dst].name] = whole[op.dst]
        else:
            logits = get(op.srcs[0])
    full = F.interpolate(logits, size=labels.shape[-2:], mode='bilinear', align_corners=True)
    loss = F.cross_entropy(full, labels.long(), ignore_index=255)
    loss.backward()
    return float(loss), leaves, kept


@pytest.mark.parametrize('taps', [1, 0, 2])
@pytest.mark.parametrize('size', [(24, 40), (34, 70)])
def test_mini_network_training_step_vs_autograd(size, taps):
    """(`taps`: option wgrad_taps - 3x3 stride-1 weight gradients with the taps folded into the matrix rows (wgrad_taps.hip): never /
    where it measured faster / everywhere; 24 x 40 runs its 2 x 64 items and, pooled, the 4 x 32 ones, 34 x 70 the padded copies.)"""
    from tests.helpers import MiniTrain
    from panoptic_forecasting_amd import lib as pflib
    pflib.check(pflib.load().pf_set_option(b'wgrad_taps', taps), 'pf_set_option')
    try:
        _mini_network_step(size)
    finally:
        pflib.load().pf_set_option(b'wgrad_taps', 1)


@pytest.mark.parametrize('size', [(24, 40), (34, 70)])
def test_mini_network_forward_on_packed_pairs_vs_autograd(size):
    """Option train_forward_s4 (csrc/train_s4.hip, off by default): the forward conv + BatchNorm layers of a step on the inference
    path's packed-pair kernels with blocked sums (conv_s4_blocked_kernel) - weights packed on the device every step, activation slices shadowed as fp16 pairs (a slice that
    starts at channel 10 of a tensor shares a 4-channel group with its neighbour), rows padded to 4 on the odd widths - against
    float64 autograd at the bars of the fp32 step."""
    from panoptic_forecasting_amd import lib as pflib
    pflib.check(pflib.load().pf_set_option(b'train_forward_s4', 1), 'pf_set_option')
    try:
        _mini_network_step(size)
    finally:
        pflib.load().pf_set_option(b'train_forward_s4', 0)


def test_forward_on_packed_pairs_is_the_same_step_to_rounding():
    """The real network, fixture batch: with train_forward_s4 the 67 stride-1 conv + BatchNorm layers run on conv_s4 (path statistics),
    the loss agrees to 1e-5, the gradients to the distance either step has from float64 (ReLU masks flip where a pre-activation
    is closer to zero than the round-off: measured 0.9 % of the whole gradient's norm)."""
    from panoptic_forecasting_amd import lib as pflib
    from panoptic_forecasting_amd.bg_train import BGTrainer
    z, batches = _fixture()
    a_in, a_lab = (_cuda(d) for d in batches[0])
    L = pflib.load()
    got = {}
    try:
        for mode in (0, 1):
            pflib.check(L.pf_set_option(b'train_forward_s4', mode), 'pf_set_option')
            tr = BGTrainer(_params())
            tr.load_state_dict(_sd())
            r = tr.forward_backward(a_in, a_lab, update_running_stats=False)
            got[mode] = (float(r['loss']), tr.grad.clone(), tr.path_stats())
            if mode:
                tr.forward_backward(a_in, a_lab, update_running_stats=False)
                assert torch.equal(tr.grad, got[1][1])          # bit-reproducible
    finally:
        L.pf_set_option(b'train_forward_s4', 0)
    assert got[0][2]['packed_pair_forward_convs'] == 0 and got[1][2]['packed_pair_forward_convs'] >= 60, (got[0][2], got[1][2])
    assert got[1][2]['generic_kernel_launches'] == 0
    assert abs(got[1][0] - got[0][0]) <= 1e-5 * abs(got[0][0])
    assert _rel(got[1][1], got[0][1]) <= 3e-2


def _mini_network_step(size):
    from tests.helpers import MiniTrain
    h, w = size
    sp = _mini_net()
    g = torch.Generator().manual_seed(17)
    params = {}
    for op in sp.conv_ops():
        pr = {'w': torch.randn(op.cout, op.cin, op.k, op.k, generator=g) * (2.0 / (op.cin * op.k * op.k)) ** 0.5}
        if op.bn:
            pr.update(gamma=torch.rand(op.cout, generator=g) + 0.5, beta=torch.randn(op.cout, generator=g) * 0.2,
                      mean=torch.zeros(op.cout), var=torch.ones(op.cout))
        else:
            pr['b'] = torch.randn(op.cout, generator=g) * 0.1
        params[op.name] = pr
    x = torch.randn(3, 6, h, w, generator=g)
    lab = torch.randint(0, 12, (3, 2 * h - 3, 2 * w + 1), generator=g)
    lab[lab == 11] = 255
    want_loss, leaves, kept = _mini_torch(sp, params, x, lab)
    net = MiniTrain(sp, params)
    got_loss = net.step(x.cuda(), lab.cuda())
    assert abs(got_loss - want_loss) <= 1e-5 * abs(want_loss)
    bad = []
    for name, t in kept.items():
        if name == 'input' or t.grad is None:
            continue
        tname, _, choff = name.partition('@')           # 'cat@10' = the slice of tensor 'cat' one op produced
        sl = slice(int(choff or 0), int(choff or 0) + t.shape[1])
        a, ga = net.tensor(tname).cpu().double()[:, sl], net.tensor(tname, grad=True).cpu().double()[:, sl]
        if _rel(a, t.detach()) > 1e-5:
            bad.append(('act ' + name, _rel(a, t.detach())))
        if _rel(ga, t.grad) > 1e-4:
            bad.append(('grad ' + name, _rel(ga, t.grad)))
    for op in sp.conv_ops():
        for nm in (('w', 'gamma', 'beta') if op.bn else ('w', 'b')):
            r = _rel(net.param('%s.%s' % (op.name, nm), grad=True).cpu(), leaves[op.name][nm].grad)
            if r > 1e-4:
                bad.append(('d%s %s' % (nm, op.name), r))
        if op.bn:      # running statistics (momentum 0.1, unbiased variance)
            for nm in ('mean', 'var'):
                r = _rel(net.param('%s.%s' % (op.name, nm)).cpu(), leaves[op.name][nm])
                if r > 1e-5:
                    bad.append(('running %s %s' % (nm, op.name), r))
    net.close()
    assert not bad, bad


def test_train_driver_runs_resumes_and_writes_reference_checkpoints(tmp_path):
    """train_bg.py with the reference's flags on synthetic crops: two epochs, then --continue_training for a third; the
    files of training/train.py:269-281 appear and the checkpoint loads into the registry's model with strict keys."""
    from panoptic_forecasting_amd import train_bg
    from panoptic_forecasting_amd.registry import build_model
    wd = str(tmp_path / 'exp')
    cfg = tmp_path / 'cfg.yaml'
    cfg.write_text('task: bg\nmodel:\n  model_type: bg\n  num_inputs: 3\n  use_depth_inps: true\n  convert2onehot: true\n'
                   'data:\n  crop_size: 64\ntraining:\n  batch_size: 2\n  num_epochs: 2\n  lr: 2.0e-3\n  mom: 0.9\n  wd: 1.0e-4\n'
                   '  clip_grad_norm: 5.0\n  lr_decay_type: step\n  lr_decay_factor: 0.1\n  lr_decay_steps: 100\n')
    train_bg.main(['--config_file', str(cfg), '--working_dir', wd, '--synthetic', '4'])
    for name in ('config.yaml', 'model_checkpoint', 'best_model', 'training_checkpoint'):
        assert os.path.exists(os.path.join(wd, name)), name
    st = torch.load(os.path.join(wd, 'training_checkpoint'))
    assert st['epoch'] == 3 and st['step'] == 4          # 2 epochs x 2 batches
    train_bg.main(['--continue_training', '--working_dir', wd, '--synthetic', '4', '--extra_args', 'training.num_epochs', '3'])
    st = torch.load(os.path.join(wd, 'training_checkpoint'))
    assert st['epoch'] == 4 and st['step'] == 6
    sd = torch.load(os.path.join(wd, 'model_checkpoint'))
    m = build_model(_params())
    m.load_state_dict(sd, strict=True)
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())


def test_two_rank_train_driver_equals_one_rank_with_accumulation(tmp_path):
    """train_bg.py as TWO ranks (both on this box's one GPU, gloo for the flat-gradient all-reduce; on a multi-GPU node the
    same code runs one rank per GPU over RCCL) against ONE rank that sees the two ranks' micro-batches one after the other
    with accumulate_steps = 2: identical trainable parameters after an epoch.  (BatchNorm uses per-rank batch statistics in the
    reference's DDP run too - train.py:96-103, no SyncBN - so "one rank on the merged batch" is not the same function; gradient
    accumulation over the same micro-batches is.  The mean of two gradients and the sum of two half-scaled ones are the same
    fp32 numbers: scaling by 1/2 commutes with rounding.)  Running statistics are rank-local and not compared."""
    import subprocess
    import sys
    from panoptic_forecasting_amd import train_bg
    from panoptic_forecasting_amd.bg_model import BGModel
    from panoptic_forecasting_amd.bg_train import BGTrainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    wd = str(tmp_path / 'exp2')
    cfg = tmp_path / 'cfg.yaml'
    cfg.write_text('task: bg\nmodel:\n  model_type: bg\n  num_inputs: 3\n  use_depth_inps: true\n  convert2onehot: true\n'
                   'data:\n  crop_size: 64\ntraining:\n  batch_size: 2\n  num_epochs: 1\n  lr: 2.0e-3\n  mom: 0.9\n  wd: 1.0e-4\n'
                   '  clip_grad_norm: 5.0\n')
    port = 29000 + os.getpid() % 1000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   PF_DIST_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'panoptic-forecasting_amd', 'train_bg.py'), '--config_file', str(cfg),
                                       '--working_dir', wd, '--synthetic', '8', '--seed', '3'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
    two = torch.load(os.path.join(wd, 'model_checkpoint'))
    st = torch.load(os.path.join(wd, 'training_checkpoint'))
    assert st['step'] == 2                          # 8 crops / (2 per batch x 2 ranks) = 2 updates
    assert set(st['optimizer']) == {'state', 'param_groups'}       # torch.optim.SGD's own format (reference train.py:285)

    # one rank, the same initial weights (the driver seeds, then builds the reference's init), the same micro-batches
    params = _params()
    params['training'] = {'batch_size': 2, 'lr': 2e-3, 'mom': 0.9, 'wd': 1e-4, 'clip_grad_norm': 5.0, 'accumulate_steps': 2}
    params['data']['depth_norm_params'] = [20.0, 15.0]
    train_bg.seed_all(3)
    tr = BGTrainer(params)
    tr.load_state_dict(BGModel(params).state_dict())
    loaders = [train_bg.SyntheticCrops(8, 64, 2, 11, rank=r, world=2) for r in range(2)]
    for b0, b1 in zip(loaders[0].batches(1), loaders[1].batches(1)):
        for batch in (b0, b1):
            tr.train_step(_cuda(batch['inputs']), _cuda(batch['labels']))
    assert tr.steps == 2
    one = tr.state_dict()
    worst = 0.0
    for key, _, _, trainable in tr.layout:
        if trainable:
            worst = max(worst, _rel(two[key], one[key]))
    assert worst <= 1e-6, worst
    # and the optimizer state the two-rank run saved loads back into a trainer, equal to the one-rank momentum
    tr2 = BGTrainer(params)
    tr2.load_optimizer_state_dict(st['optimizer'])
    m = tr.trainable.bool()
    assert _rel(tr2.momentum_buf[m].cpu(), tr.momentum_buf[m].cpu()) <= 1e-6
    # ... and into the reference's optimizer class unchanged
    dummy = [torch.nn.Parameter(torch.zeros(shape)) for _, _, shape, _ in tr.trainable_layout()]
    torch.optim.SGD(dummy, lr=2e-3, momentum=0.9).load_state_dict(st['optimizer'])


def test_bgmodel_under_distributed_data_parallel(tmp_path):
    """The reference's data-parallel wrapping, unmodified (training/train.py:96-103: ``DistributedDataParallel(DistWrapper(model))``,
    models/dist_wrapper.py:13-26), around THIS package's registry model: two ranks (gloo, sharing the one GPU) run two steps of the
    reference loop body (tests/ddp_worker.py).  DDP's reducer hooks fire on the parameter gradients the fused device step hands to
    autograd, so after ``backward()`` every rank holds the mean gradient - which must equal what the flat exchange of
    ``train_bg.py`` / ``BGTrainer`` computes for the same micro-batches (the sum of the two half-scaled gradients) to 1e-6, and so
    must the parameters after the first clipped SGD step (after the second: 1e-4, see below).  BatchNorm uses per-rank batch statistics under DDP (no SyncBN in the
    reference), exactly as per micro-batch here."""
    import subprocess
    import sys
    import ddp_worker
    from panoptic_forecasting_amd.bg_train import BGTrainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / 'ddp.pt')
    port = 29000 + (os.getpid() + 7) % 1000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'tests', 'ddp_worker.py'), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
    got = torch.load(out)
    # every rank ends with the same parameters (same averaged gradients, same update)
    n0, n1 = got['param_norms_by_rank']
    assert all(abs(n0[k] - n1[k]) <= 1e-7 * (abs(n0[k]) + 1e-12) for k in n0)

    tr = BGTrainer(ddp_worker.params())
    tr.load_state_dict(ddp_worker.state_dict())
    for step in range(2):
        for r in range(2):
            inputs, labels = ddp_worker.micro_batches(step, r)
            tr.forward_backward(_cuda(inputs), _cuda(labels), accumulate=r > 0, loss_scale=0.5)
        if step == 0:
            flat = {k: v.cpu().clone() for k, v in tr.named_grads().items()}
            assert set(got['grads']) == set(flat)
            worst = max(_rel(got['grads'][k], flat[k]) for k in flat)
            assert worst <= 1e-6, worst
        tr.optimizer_step()
        one = tr.state_dict()
        worst = max(_rel(got['params_step1' if step == 0 else 'params'][key], one[key]) for key, _, _, trainable in tr.layout if trainable)
        # after the first update: torch.optim.SGD + clip_grad_norm_ on DDP's averaged gradients against pf_sgd_step on the flat
        # exchange's - the same arithmetic up to the rounding of the clip coefficient.  After the second: the two runs computed
        # their second gradients at parameters that differ in the last bits, and this network's gradient moves ~sqrt(round-off)
        # for such a perturbation (ReLU masks flip, _oracle_grads): measured 1.7e-5, bar 1e-4 (the reference-fixture test allows
        # 2e-3 for two implementations' two steps)
        assert worst <= (1e-6 if step == 0 else 1e-4), (step, worst)


def test_training_mode_loss_under_no_grad_uses_batch_statistics():
    """model.train() + torch.no_grad(): the reference's BatchNorm still normalises with batch statistics (and updates its running
    statistics); so does BGModel.loss - same value as the grad-enabled call on the same parameters, no autograd node, and not the
    folded running-statistics value of eval()."""
    from panoptic_forecasting_amd.registry import build_model
    z, batches = _fixture()
    inputs, labels = batches[0]
    m = build_model(_params())
    m.load_state_dict(_sd())
    m.cuda()
    m.train()
    with torch.no_grad():
        a = m.loss(_cuda(inputs), _cuda(labels))
    assert a['loss'].grad_fn is None
    assert abs(float(a['loss']) - z['loss'][0]) <= LOSS_REL * z['loss'][0]        # the reference's train-mode loss of this batch
    m2 = build_model(_params())
    m2.load_state_dict(_sd())
    m2.cuda()
    m2.eval()
    e = m2.loss(_cuda(inputs), _cuda(labels))
    assert abs(float(e['loss']) - float(a['loss'])) > 10 * LOSS_REL * float(a['loss'])
    assert int(m.state_dict()['model.base.0.norm.num_batches_tracked']) == 1
