"""Adversarial cases for the two-term fp16 operand path (conv_split / conv_s4; conv_mfma.h) against float64 torch and the
oracle: activations far outside O(1) IN EITHER DIRECTION must come out right - to a tolerance relative to the tensor's own
magnitude - or be FLAGGED (PF_STATUS_RANGE above 65504, PF_STATUS_RANGE_LOW for tensors of tiny values) - never silently
clamped or flushed - and a cancellation-heavy convolution must stay inside the stated operand bound.  The reference is plain
fp32 Conv2d (hardnet.py:16-25).  Measured errors are written to gpurun_out/r04_precision.json (copied to profiles/)."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT

pytestmark = pytest.mark.gpu
REPORT = os.path.join(ROOT, 'gpurun_out', 'r04_precision.json')
RANGE, RANGE_LOW = 1, 2          # include/pfhip.h: PF_STATUS_RANGE, PF_STATUS_RANGE_LOW


def _report(key, value):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    data = {}
    if os.path.exists(REPORT):
        with open(REPORT) as f:
            data = json.load(f)
    data[key] = value
    with open(REPORT, 'w') as f:
        json.dump(data, f, indent=1, sort_keys=True)


@pytest.fixture
def normalize_ranges():
    """pf_set_option('normalize_ranges', v): read when a plan is created; restored to the default (1) afterwards"""
    from panoptic_forecasting_amd import lib as pflib
    L = pflib.load()
    yield lambda v: pflib.check(L.pf_set_option(b'normalize_ranges', int(v)), 'pf_set_option')
    L.pf_set_option(b'normalize_ranges', 1)


@pytest.fixture
def force_conv():
    from panoptic_forecasting_amd import lib as pflib
    L = pflib.load()
    yield lambda kind, p0, p1, p2: pflib.check(L.pf_debug_force_conv(kind, p0, p1, p2), 'pf_debug_force_conv')
    L.pf_debug_force_conv(0, 0, 0, 0)


def _scaled_block(g, scale):
    """The HarDBlock-shaped net of test_gpu_conv.py with its input AND every bias multiplied by ``scale``: a ReLU network is
    positively homogeneous, so every activation of the reference is exactly ``scale`` times the unscaled one."""
    from test_gpu_conv import _block_net
    spec, P = _block_net(g, 12)
    return spec, {k: (w, b * scale) for k, (w, b) in P.items()}


def _check_block(net, ref, tag, unit=1e-5):
    """every tapped tensor within k * unit * max|ref| of float64 torch - relative to the TENSOR's magnitude, no '1 +'"""
    worst = 0.0
    for name, k in [('t0', 1), ('L2', 1), ('out', 2), ('p', 2), ('c6', 3), ('c7', 3), ('c8', 3)]:
        r = ref[name].float()
        mag = r.abs().max().item()
        err = (net.tensor(name).cpu() - r).abs().max().item()
        assert err <= k * unit * mag, (tag, name, err, k * unit * mag)
        worst = max(worst, err / mag)
    return worst


@pytest.mark.parametrize('force', [(5, 2, 0, 0), (4, 2, 0, 0), None], ids=['conv_s4', 'conv_split', 'table'])
@pytest.mark.parametrize('scale', [1e-7, 1e-5, 1e-3, 1.0, 1e3, 1e5, 3e7])
def test_scaled_activations_through_a_hardblock_are_right_or_flagged(scale, force, force_conv):
    """Activations of ~5e-7 .. ~1e8.  Either every tensor agrees with float64 torch to the split tolerance RELATIVE to its
    own magnitude and the status word is clear, or the status word carries PF_STATUS_RANGE (values beyond 65504) or
    PF_STATUS_RANGE_LOW (a tensor whose largest value is below 2^-6, where the pair's absolute 2^-25 floor would cost relative
    precision that fp32 keeps); then the same plan with split_f16 = 0 (fp32 matrix instructions, what BGModel re-runs) must
    be right and unflagged.  A silently clamped or flushed result fails both branches."""
    from helpers import MiniNet
    from test_gpu_conv import _block_ref
    g = torch.Generator().manual_seed(11)
    b, h, w = 1, 24, 40
    x = torch.randn(b, 12, h, w, generator=g) * scale
    spec, P = _scaled_block(g, scale)
    if force:
        force_conv(*force)
    ref = _block_ref(x, P, h, w)
    peak = max(r.abs().max().item() for r in ref.values())
    net = MiniNet(spec, P).run(x.cuda())
    status = net.status()
    flagged = bool(status & (RANGE | RANGE_LOW))
    # (the first conv reads the caller's fp32 input: the dense-input pre-pass has looked at it too)
    if max(peak, x.abs().max().item()) > 65504.0:
        assert status & RANGE, 'activations up to %g were not flagged' % peak
    if x.abs().max().item() < 2.0 ** -6:
        assert status & RANGE_LOW, 'an input of at most %g was not flagged' % x.abs().max().item()
    if scale == 1.0:
        assert not flagged
    rel = None
    if not flagged:
        rel = _check_block(net, ref, 'split')
    net.set_option('split_f16', 0).run(x.cuda())
    assert net.status() == 0, 'the fp32-only plan must never raise a range flag'
    rel32 = _check_block(net, ref, 'fp32')
    _report('hardblock_scale_%g_%s' % (scale, 'table' if not force else 'kind%d' % force[0]),
            {'peak_activation': peak, 'status': status, 'max_rel_err_split': rel, 'max_rel_err_fp32_rerun': rel32})
    net.close()


@pytest.mark.parametrize('force', [(5, 2, 0, 0), (4, 2, 0, 0), None], ids=['conv_s4', 'conv_split', 'table'])
@pytest.mark.parametrize('case', ['zero_input', 'dead_layer'])
def test_exact_zeros_are_not_a_low_range(case, force, force_conv):
    """include/pfhip.h: PF_STATUS_RANGE_LOW needs a NON-ZERO maximum below 2^-6.  Exact zeros lose nothing in the fp16 pair, so
    an all-zero dense input with zero biases (every activation exactly 0) and a layer whose ReLU kills every value (biases of
    -1e3: the tensor t0 and everything behind its consumers' zero contribution) must leave the status word clear - no fp32
    re-run, no PfError after graph replays - and the results must still equal float64 torch."""
    from helpers import MiniNet
    from test_gpu_conv import _block_net, _block_ref
    g = torch.Generator().manual_seed(17)
    b, h, w = 1, 24, 40
    spec, P = _block_net(g, 12)
    P = dict(P)
    if case == 'zero_input':
        x = torch.zeros(b, 12, h, w)
        P = {k: (wt, torch.zeros_like(bs)) for k, (wt, bs) in P.items()}
    else:
        x = torch.randn(b, 12, h, w, generator=g)
        P['t0'] = (P['t0'][0], torch.full_like(P['t0'][1], -1e3))
    if force:
        force_conv(*force)
    ref = _block_ref(x, P, h, w)
    assert ref['t0'].abs().max().item() == 0.0
    net = MiniNet(spec, P).run(x.cuda())
    assert net.status() == 0, 'exact zeros raised status %d' % net.status()
    for name in ('t0', 'L2', 'out', 'c7', 'c8'):
        r = ref[name].float()
        err = (net.tensor(name).cpu() - r).abs().max().item()
        assert err <= 3e-5 * max(r.abs().max().item(), 1e-30) + 0.0, (name, err)
    net.close()


@pytest.mark.parametrize('alpha', [1e-5, 1e-3, 1e3])
def test_reparameterised_block_runs_on_the_pair_path(alpha, force_conv, normalize_ranges):
    """The same function with other numbers: the first conv's weights and bias times alpha, every consumer's columns of that
    tensor divided by alpha.  The reference (fp32) does not care.  With the plan's range normalisation (the stored channel
    = value * 2^k chosen from the folded weights) the pair path stores what it stores for alpha = 1: unflagged and right
    relative to each tensor's magnitude, the tiny tensor t0 included.  Without it (normalize_ranges = 0) alpha = 1e-5 is a
    tensor of ~1e-5 values: it must be FLAGGED (PF_STATUS_RANGE_LOW), never silently short of bits."""
    from helpers import MiniNet
    from test_gpu_conv import _block_net, _block_ref
    g = torch.Generator().manual_seed(13)
    b, h, w = 1, 24, 40
    x = torch.randn(b, 12, h, w, generator=g)
    spec, P = _block_net(g, 12)
    P = dict(P)
    P['t0'] = (P['t0'][0] * alpha, P['t0'][1] * alpha)
    for name, lo in (('L1', 0), ('L2', 10), ('L4', 28)):       # input columns that read t0 (test_gpu_conv._block_net)
        wt = P[name][0].clone()
        wt[:, lo:lo + 20] /= alpha
        P[name] = (wt, P[name][1])
    ref = _block_ref(x, P, h, w)
    force_conv(5, 2, 0, 0)
    net = MiniNet(spec, P).run(x.cuda())
    assert net.status() == 0, net.status()
    rel = _check_block(net, ref, 'normalised')
    net.close()
    normalize_ranges(0)
    raw = MiniNet(spec, P).run(x.cuda())
    st = raw.status()
    t0_max = ref['t0'].abs().max().item()
    if t0_max < 2.0 ** -6:
        assert st & RANGE_LOW, (st, t0_max)
    if t0_max > 65504.0:
        assert st & RANGE, (st, t0_max)
    rel_raw = _check_block(raw, ref, 'raw') if st == 0 else None
    raw.close()
    _report('reparameterised_block_alpha_%g' % alpha, {'t0_max': t0_max, 'max_rel_err_normalised': rel, 'status_without_normalisation': st,
                                                       'max_rel_err_without_normalisation': rel_raw})


def test_range_flag_catches_a_single_outlier_channel(force_conv, normalize_ranges):
    """One output channel with a huge bias pushes ONE value per pixel past 65504 in the middle of the block.  With the values
    stored raw (normalize_ranges = 0) that is flagged; with the plan's range normalisation that channel alone is stored times
    2^-13 or so (its bias is in the plan's estimate of its magnitude): unflagged, and every tensor right relative to its size."""
    from helpers import MiniNet
    from test_gpu_conv import _block_net, _block_ref
    g = torch.Generator().manual_seed(12)
    x = torch.randn(1, 12, 16, 64, generator=g)
    spec, P = _block_net(g, 12)
    force_conv(5, 2, 0, 0)
    net = MiniNet(spec, P).run(x.cuda())
    assert net.status() == 0
    net.close()
    w, bias = P['L2']
    bias = bias.clone()
    bias[7] = 7e4
    P['L2'] = (w, bias)
    net = MiniNet(spec, P).run(x.cuda())
    assert net.status() == 0
    _check_block(net, _block_ref(x, P, 16, 64), 'outlier channel, normalised')
    net.close()
    normalize_ranges(0)
    net = MiniNet(spec, P).run(x.cuda())
    assert net.status() & RANGE
    net.close()


@pytest.mark.parametrize('k,cin,cout', [(3, 91, 28), (1, 126, 63), (3, 48, 10)])
def test_cancellation_heavy_conv_on_both_paths(k, cin, cout, force_conv):
    """sum |w x| >> |sum w x|: every filter sums to zero over its taps and channels and the input is 1000 + noise, so the
    common mode (sum |terms| ~ 1000 sum|w|) cancels and only the noise term survives.  Errors of the fp16-pair path and of the
    fp32-MFMA path against float64, both measured in units of sum|w x| (the quantity rounding errors scale with):
        fp32 path:  <= K 2^-24            (accumulation only; operands exact)
        pair path:  <= 2^-21 + K 2^-24    (2^-23 per operand, 2^-22 for the dropped mid*mid product, + accumulation)
    Worst-case bounds; the measured figures (reported) sit one to two orders below and within a small factor of each other."""
    from helpers import MiniNet, MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    g = torch.Generator().manual_seed(k * 100 + cin)
    b, h, w = 1, 32, 64
    x = 1000.0 + torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g)
    wt = wt - wt.mean(dim=(1, 2, 3), keepdim=True)
    wt = (wt.double() - wt.double().mean(dim=(1, 2, 3), keepdim=True)).float()
    bias = torch.zeros(cout)
    ref = F.conv2d(x.double(), wt.double(), None, padding=k // 2)
    mag = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=k // 2)      # sum |w x| per output
    K = cin * k * k
    out = {}
    for name, force, split in [('pair', (5, 2, 0, 0), 1), ('fp32', None, 0)]:
        spec = MiniSpec(cin)
        t0 = spec.conv('id', [arch.Src(0, 0, cin)], cin, 1, relu=False)          # identity 1x1: puts x into the packed layout
        spec.conv('c', [arch.Src(t0, 0, cin)], cout, k, relu=False)
        eye = torch.eye(cin).view(cin, cin, 1, 1)
        if force:
            force_conv(*force)
        else:
            force_conv(0, 0, 0, 0)
        net = MiniNet(spec, {'id': (eye, torch.zeros(cin)), 'c': (wt, bias)})
        net.set_option('split_f16', split).run(x.cuda())
        assert net.status() == 0
        got = net.tensor('c').cpu().double()
        out[name] = ((got - ref).abs() / mag).max().item()
        net.close()
    interior = (ref.abs() / mag).median().item()
    assert interior < 5e-3, 'the case is not cancellation-heavy: |sum| / sum|.| = %g' % interior
    _report('cancellation_k%d_cin%d_cout%d' % (k, cin, cout),
            {'K': K, 'median_|sum|_over_sum|terms|': interior, 'err_over_sum|terms|_pair': out['pair'],
             'err_over_sum|terms|_fp32': out['fp32'], 'bound_pair': 2.0 ** -21 + K * 2.0 ** -24, 'bound_fp32': K * 2.0 ** -24})
    assert out['fp32'] <= K * 2.0 ** -24, out
    assert out['pair'] <= 2.0 ** -21 + K * 2.0 ** -24, out
    assert out['pair'] <= 2.0 ** -21, out      # in practice the pair path is inside the OPERAND bound alone


def _bg_params(h, w, std, **model_kw):
    p = {'task': 'bg_forecast', 'no_gpu': False, 'load_model': None, 'load_best_model': False,
         'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([std])],
                  'min_depth': 0.1, 'max_depth': 200},
         'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True, 'final_h': h, 'final_w': w,
                   'return_logits': 'orig'}}
    p['model'].update(model_kw)
    return p


def test_whole_network_out_of_range_checkpoint_reruns_in_fp32_or_raises():
    """A checkpoint whose depth normalisation is unusable for fp16 pairs (depth_std = 2e-4: normalised depths of ~1e5-1e6
    enter the stem, activations of 1e5+ follow) through the fused model: the default policy re-runs the forward on the fp32
    matrix instructions and returns what the oracle (torch fp32 on the CPU) returns, to the usual tolerance relative to the
    logits' magnitude; 'raise' fails loudly; a normal checkpoint never re-runs."""
    import test_gpu_bg_forecast as t
    from panoptic_forecasting_amd import lib as pflib
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.registry import build_model
    h, w = 128, 256
    sd = dict(t._sd())
    inp = synth.make_inputs(b=1, h=h, w=w, seed=5, gap_len=3)
    cuda_inp = {k: v.cuda() for k, v in inp.items()}
    ok = build_model(_bg_params(h, w, 15.0))
    ok.load_state_dict(sd)
    ok.predict(cuda_inp, None)['seg']        # (first access of a result = the range check of its forward)
    assert ok.bg.range_reruns == 0 and ok.bg.range_status() == 0

    sd_bad = dict(sd)
    sd_bad['depth_std'] = torch.tensor([2e-4])
    m = build_model(_bg_params(h, w, 2e-4))
    m.load_state_dict(sd_bad)
    out = m.predict(cuda_inp, None)
    assert m.bg.range_reruns == 0            # predict() only enqueued: nothing has been waited for
    out['seg']
    assert m.bg.range_reruns == 1
    ref, _, _ = t.oracle_pipeline(sd_bad, inp, h, w)
    scale = ref['orig_size_logits'].abs().max().item()
    err = (out['orig_size_logits'].cpu() - ref['orig_size_logits']).abs().max().item()
    assert err <= 1e-4 * (1.0 + scale), (err, scale)
    assert (out['seg'].cpu().long() == ref['seg']).float().mean().item() >= 0.999
    _report('whole_network_depth_std_2e-4', {'max_abs_logit': scale, 'err_after_fp32_rerun': err, 'reruns': m.bg.range_reruns})

    r = build_model(_bg_params(h, w, 2e-4, on_range_overflow='raise'))
    r.load_state_dict(sd_bad)
    with pytest.raises(pflib.PfError, match='65504'):
        r.predict(cuda_inp, None)['seg']
    i = build_model(_bg_params(h, w, 2e-4, on_range_overflow='ignore'))
    i.load_state_dict(sd_bad)
    i.predict(cuda_inp, None)['seg']
    assert i.bg.range_status() & RANGE and i.bg.range_reruns == 0
    assert i.bg.range_status_sticky(clear=True) & RANGE and i.bg.range_status_sticky() == 0


def test_overflow_inside_the_fused_front_end_is_flagged(normalize_ranges):
    """base.1's output never leaves LDS in conv_front.hip - it is still a tensor a split kernel reads (base.2, inside the same
    kernel), so its producer must guard it: with base.1's BatchNorm affine scaled by 1e5 the stem output stays in range and the
    overflow arises INSIDE the fused kernel.  Without the plan's range normalisation: default policy = one re-run on the fp32
    matrix instructions, result = oracle; 'ignore': the status word carries the flag and the fused kernel is what ran.  With
    the normalisation (the default) the 1e5 is absorbed into the stored channels' powers of two: no flag, no re-run, and the
    pair path itself returns the oracle's logits."""
    import test_gpu_bg_forecast as t
    from panoptic_forecasting_amd import lib as pflib
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.registry import build_model
    h, w = 128, 256
    sd = dict(t._sd())
    for k in ('model.base.1.norm.weight', 'model.base.1.norm.bias'):
        assert k in sd, [q for q in sd if 'base.1.' in q]
        sd[k] = sd[k] * 1e5
    inp = synth.make_inputs(b=1, h=h, w=w, seed=6, gap_len=3)
    cuda_inp = {k: v.cuda() for k, v in inp.items()}
    ref, _, _ = t.oracle_pipeline(sd, inp, h, w)
    scale = ref['orig_size_logits'].abs().max().item()

    n = build_model(_bg_params(h, w, 15.0))
    n.load_state_dict(sd)
    out = n.predict(cuda_inp, None)
    err_n = (out['orig_size_logits'].cpu() - ref['orig_size_logits']).abs().max().item()
    assert n.bg.range_reruns == 0 and n.bg.range_status() == 0
    assert err_n <= 1e-4 * scale, (err_n, scale)

    normalize_ranges(0)
    i = build_model(_bg_params(h, w, 15.0, on_range_overflow='ignore'))
    i.load_state_dict(sd)
    pflib.profile(True)
    i.predict(cuda_inp, None)['seg']
    labels = [r['label'] for r in pflib.profile_results()]
    pflib.profile(False)
    assert any('conv_front' in l for l in labels), labels
    assert i.bg.range_status() & RANGE and i.bg.range_reruns == 0
    m = build_model(_bg_params(h, w, 15.0))
    m.load_state_dict(sd)
    out = m.predict(cuda_inp, None)
    err = (out['orig_size_logits'].cpu() - ref['orig_size_logits']).abs().max().item()
    assert m.bg.range_reruns == 1
    assert err <= 1e-4 * scale, (err, scale)
    _report('overflow_inside_conv_front', {'max_abs_logit': scale, 'err_after_fp32_rerun': err, 'reruns': m.bg.range_reruns,
                                           'err_normalised_pair_path': err_n})


def _reparameterise(sd, producer, alpha):
    """(gamma, beta) of ``producer``'s BatchNorm times alpha, the input columns of every conv that reads its output divided by
    alpha: the same function for the reference (hardnet.py:16-25, ReLU commutes with a positive factor)."""
    from panoptic_forecasting_amd import hardnet_arch as arch
    sd = dict(sd)
    for k in ('model.%s.norm.weight' % producer, 'model.%s.norm.bias' % producer):
        sd[k] = sd[k] * alpha
    spec = arch.Spec(36, 11)
    src_t = next(op.dst for op in spec.ops if op.name == producer)
    hit = []
    for op in spec.conv_ops():
        off = 0
        for s_ in op.srcs:
            if s_.tensor == src_t:
                key = 'model.%s.conv.weight' % op.name
                wt = sd[key].clone()
                wt[:, off:off + s_.ch] /= alpha
                sd[key] = wt
                hit.append(op.name)
            off += s_.ch
    assert hit, producer
    return sd, hit


@pytest.mark.parametrize('alpha', [1e-5, 1e5])
def test_reparameterised_checkpoint_gives_the_oracles_logits(alpha, normalize_ranges):
    """model.base.3's BatchNorm times alpha and its consumers' weights divided by alpha: the same function, so the logits must
    equal the oracle's (torch fp32 on the CPU, run on the SAME re-parameterised checkpoint) to 1e-4 - on the fp16-pair path
    itself with the plan's range normalisation (no flag, no re-run), and through a flagged fp32 re-run without it."""
    import test_gpu_bg_forecast as t
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.registry import build_model
    h, w = 128, 256
    sd, hit = _reparameterise(t._sd(), 'base.3', alpha)
    inp = synth.make_inputs(b=1, h=h, w=w, seed=7, gap_len=3)
    cuda_inp = {k: v.cuda() for k, v in inp.items()}
    ref, _, _ = t.oracle_pipeline(sd, inp, h, w)
    plain, _, _ = t.oracle_pipeline(dict(t._sd()), inp, h, w)
    assert (ref['orig_size_logits'] - plain['orig_size_logits']).abs().max().item() < 1e-3      # the same function indeed
    m = build_model(_bg_params(h, w, 15.0))
    m.load_state_dict(sd)
    out = m.predict(cuda_inp, None)
    err = (out['orig_size_logits'].cpu() - ref['orig_size_logits']).abs().max().item()
    assert m.bg.range_reruns == 0 and m.bg.range_status() == 0
    assert err <= 1e-4, err
    assert (out['seg'].cpu().long() == ref['seg']).float().mean().item() >= 0.999
    normalize_ranges(0)
    r = build_model(_bg_params(h, w, 15.0))
    r.load_state_dict(sd)
    out_r = r.predict(cuda_inp, None)
    err_r = (out_r['orig_size_logits'].cpu() - ref['orig_size_logits']).abs().max().item()
    st = r.bg.range_status_sticky()
    assert st & (RANGE_LOW if alpha < 1 else RANGE), st
    assert r.bg.range_reruns == 1
    assert err_r <= 1e-4, err_r
    _report('reparameterised_base3_alpha_%g' % alpha, {'consumers': hit, 'err_pair_path_normalised': err, 'status_without_normalisation': st,
                                                      'err_after_fp32_rerun_without_normalisation': err_r})


def test_predict_is_asynchronous_and_a_late_check_still_reruns():
    """predict() only enqueues: two forwards are issued back to back, the first one (an out-of-range checkpoint) is found
    flagged when ITS result is first touched - after the second forward was already enqueued behind it - and is re-run into
    the same output tensors."""
    import test_gpu_bg_forecast as t
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.registry import build_model
    h, w = 128, 256
    sd_bad = dict(t._sd())
    sd_bad['depth_std'] = torch.tensor([2e-4])
    m = build_model(_bg_params(h, w, 2e-4))
    m.load_state_dict(sd_bad)
    inp = [synth.make_inputs(b=1, h=h, w=w, seed=s_, gap_len=3) for s_ in (5, 8)]
    cuda = [{k: v.cuda() for k, v in i.items()} for i in inp]
    m.predict(cuda[0], None)['seg']              # warm-up: plan, workspace
    n0 = m.bg.range_reruns
    a = m.predict(cuda[0], None)
    b = m.predict(cuda[1], None)
    assert list(a.keys()) and m.bg.range_reruns <= n0 + 1     # (the second predict may already have settled the first)
    ra, _, _ = t.oracle_pipeline(sd_bad, inp[0], h, w)
    rb, _, _ = t.oracle_pipeline(sd_bad, inp[1], h, w)
    for out, ref in ((a, ra), (b, rb)):
        scale = ref['orig_size_logits'].abs().max().item()
        err = (out['orig_size_logits'].cpu() - ref['orig_size_logits']).abs().max().item()
        assert err <= 1e-4 * scale, (err, scale)
    assert m.bg.range_reruns == n0 + 2


def test_predict_under_inference_mode():
    """Inference tensors have no version counter (`t._version` raises): `task: bg` settles such a forward before predict()
    returns instead of stamping its inputs; `task: bg_forecast` re-runs from tensors it made itself and stays asynchronous.
    Both give the oracle's result, flagged checkpoints included."""
    import test_gpu_bg_forecast as t
    import test_gpu_bg_model as tb
    from oracle import hardnet_ref
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.registry import build_model
    h, w = 64, 128
    for std in (15.0, 2e-4):                    # a normal checkpoint, and one whose every forward is flagged and re-run
        sd = dict(tb._sd())
        sd['depth_std'] = torch.tensor([std])
        p = _bg_params(h, w, std)
        p['task'] = 'bg'
        p['model']['return_logits'] = False
        m = build_model(p)
        m.load_state_dict(sd)
        frame = synth.make_bg_inputs(b=1, h=h, w=w, seed=3)
        ref = hardnet_ref.bg_predict(sd, frame, final_size=(h, w))
        scale = ref['orig_size_logits'].abs().max().item()
        with torch.inference_mode():
            inp = {k: v.clone().cuda() for k, v in frame.items()}
            assert inp['seg'].is_inference()
            out = m.predict(inp, None)
            err = (out['orig_size_logits'].cpu() - ref['orig_size_logits']).abs().max().item()
        assert err <= 1e-4 * (1.0 + scale), (std, err, scale)
        assert m.range_reruns == (0 if std > 1 else 1)
        f = build_model(_bg_params(h, w, std))
        f.load_state_dict(sd)
        finp = synth.make_inputs(b=1, h=h, w=w, seed=5, gap_len=3)
        fref, _, _ = t.oracle_pipeline(sd, finp, h, w)
        with torch.inference_mode():
            fout = f.predict({k: v.clone().cuda() for k, v in finp.items()}, None)
            ferr = (fout['orig_size_logits'].cpu() - fref['orig_size_logits']).abs().max().item()
        assert ferr <= 1e-4 * (1.0 + fref['orig_size_logits'].abs().max().item()), (std, ferr)


def test_late_rerun_refuses_inputs_refilled_in_place_and_sync_mode_is_safe():
    """`task: bg` reads the CALLER's tensors.  A flagged forward is re-run from them when its result is first touched; if the
    caller refilled them in place in between (a static input buffer), the re-run would put the newer frame's result into the
    older frame's outputs.  The lazy default detects the refill through the tensors' version counters and raises;
    model.range_check = 'sync' settles the forward inside predict(), so the same loop returns the ORIGINAL frame's result."""
    import test_gpu_bg_model as tb
    from oracle import hardnet_ref
    from panoptic_forecasting_amd import lib as pflib
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.registry import build_model
    h, w = 64, 128
    sd_bad = dict(tb._sd())
    sd_bad['depth_std'] = torch.tensor([2e-4])          # normalised depths of 1e5+: every forward is flagged

    def model(**kw):
        p = _bg_params(h, w, 2e-4, **kw)
        p['task'] = 'bg'
        p['model']['return_logits'] = False
        m = build_model(p)
        m.load_state_dict(sd_bad)
        return m

    frames = [synth.make_bg_inputs(b=1, h=h, w=w, seed=s_) for s_ in (3, 4)]
    ref0 = hardnet_ref.bg_predict(sd_bad, frames[0], final_size=(h, w))
    scale = ref0['orig_size_logits'].abs().max().item()

    lazy = model()
    static = {k: v.clone().cuda() for k, v in frames[0].items()}
    lazy.predict(static, None)['seg']                   # warm-up (plan, workspace); re-runs once
    out = lazy.predict(static, None)
    for k in static:
        static[k].copy_(frames[1][k].cuda())            # the next frame lands in the same buffers before `out` was read
    with pytest.raises(pflib.PfError, match='modified in place'):
        out['seg']
    # the failure belongs to THAT result: a later predict on the same model is not killed by it (its housekeeping settles older
    # forwards without raising), and touching the failed result again raises again instead of handing out unchecked tensors
    nxt = lazy.predict({k: v.clone() for k, v in static.items()}, None)
    nxt['seg']
    with pytest.raises(pflib.PfError, match='modified in place'):
        out['orig_size_logits']
    with pytest.raises(pflib.PfError, match='modified in place'):
        dict(out)

    sync = model(range_check='sync')
    static = {k: v.clone().cuda() for k, v in frames[0].items()}
    n0 = sync.range_reruns
    out = sync.predict(static, None)
    assert sync.range_reruns == n0 + 1                  # settled (and re-run on fp32) before predict() returned
    for k in static:
        static[k].copy_(frames[1][k].cuda())
    err = (out['orig_size_logits'].cpu() - ref0['orig_size_logits']).abs().max().item()
    assert err <= 1e-4 * (1.0 + scale), (err, scale)

    # untouched inputs: the lazy path re-runs as before
    keep = {k: v.clone().cuda() for k, v in frames[0].items()}
    out = lazy.predict(keep, None)
    err = (out['orig_size_logits'].cpu() - ref0['orig_size_logits']).abs().max().item()
    assert err <= 1e-4 * (1.0 + scale), (err, scale)
    # dict(out) / {**out} of an unchecked result settle it too (LazyResult is not a dict subclass)
    out = lazy.predict(keep, None)
    n1 = lazy.range_reruns
    plain = dict(out)
    assert type(plain) is dict and lazy.range_reruns == n1 + 1
    assert (plain['orig_size_logits'].cpu() - ref0['orig_size_logits']).abs().max().item() <= 1e-4 * (1.0 + scale)
