"""Fused bg_forecast (splat x3 -> hop -> HarDNet) vs the same pipeline composed from the oracles."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


def _sd():
    from panoptic_forecasting_amd import synth
    with open(os.path.join(G, 'calib_seed1234.json')) as f:
        return synth.make_state_dict(seed=1234, calib=json.load(f))


def _params(h, w, **kw):
    p = {'task': 'bg_forecast', 'no_gpu': False, 'load_model': None, 'load_best_model': False,
         'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])],
                  'min_depth': 0.1, 'max_depth': 200},
         'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True, 'final_h': h, 'final_w': w}}
    p['model'].update(kw)
    return p


def oracle_pipeline(sd, inp, h, w):
    """Reference two-stage pipeline on the CPU: 3 x pc_transform(only_this_ind=t) -> export hop -> load hop -> bg."""
    from oracle import hardnet_ref
    from oracle import warp_splat as ow
    from panoptic_forecasting_amd import synth
    segs, deps = [], []
    for t in range(3):
        o = ow.predict(inp, only_this_ind=t)
        # export side (export_cityscapes_segmentation_results.py:34-38,119-124)
        tid = torch.from_numpy(synth.ID2TRAINID)[o['seg'].long()]
        q = ((o['depth'] + 1).clamp(0, 255) * 256).round().numpy().astype(np.uint16)
        # load side (bg_dataset.py:224-230,166-170)
        d = torch.from_numpy(q.astype(np.float32)) / 256.0 - 1
        m = d > 0
        d[~m] = -1
        d[m & (d > 200)] = 200
        d[m & (d < 0.1)] = 0.1
        segs.append(tid)
        deps.append(d)
    seg = torch.stack(segs, 1).long()
    dep = torch.stack(deps, 1)
    return hardnet_ref.bg_predict(sd, {'seg': seg, 'depth': dep, 'depth_mask': dep > 0}, final_size=(h, w)), seg, dep


@pytest.mark.parametrize('h,w,b,gap,predicted', [(128, 256, 2, 3, False), (192, 320, 1, 9, False), (192, 320, 2, 9, True)],
                         ids=['short', 'mid_measured_odom', 'mid_predicted_odom'])
def test_fused_matches_two_stage_oracle(h, w, b, gap, predicted):
    """BASELINE configs[1] and configs[2] (dt=9 with the predicted-odometry ego chain, pc_transform_dataset.py:156-186)
    through the fused model."""
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.registry import build_model
    sd = _sd()
    m = build_model(_params(h, w, return_logits=True))
    m.load_state_dict(sd)
    inp = synth.make_inputs(b=b, h=h, w=w, seed=21, gap_len=gap, predicted=predicted)
    ref, seg_w, dep_w = oracle_pipeline(sd, inp, h, w)
    out = m.predict({k: v.cuda() for k, v in inp.items()}, None)
    # the splat stage is bit-exact, so the hop inputs are too
    assert torch.equal(torch.from_numpy(synth.ID2TRAINID)[out['warped_seg'].cpu().long()].long(), seg_w)
    err = (out['orig_size_logits'].cpu() - ref['orig_size_logits']).abs().max().item()
    agree = (out['seg'].cpu().long() == ref['seg']).float().mean().item()
    assert err <= 1e-3, err
    assert agree >= 0.999, agree


def test_fp32_mfma_only_option():
    """pf_set_option("split_f16", 0) ("split_bf16", its round-1 name, is still accepted - used below): every convolution on the fp32 matrix/vector
    pipes - the logits agree with the torch-CPU oracle an order of magnitude closer than the stated 1e-3 (pure fp32 FMA
    chains, only the summation order differs).  The default (3x3/1x1 layers on the 16-bit matrix pipe with every operand
    split into two fp16 terms, 22 significand bits) is held to the SAME 1e-4."""
    from panoptic_forecasting_amd import lib as pflib
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.registry import build_model
    h, w = 256, 512          # large enough for the heuristic to pick the split kernels when they are enabled
    sd = _sd()
    inp = synth.make_inputs(b=1, h=h, w=w, seed=4, gap_len=3)
    ref, _, _ = oracle_pipeline(sd, inp, h, w)
    errs = {}
    L = pflib.load()
    try:
        for split in (1, 0):
            pflib.check(L.pf_set_option(b'split_bf16', split), 'pf_set_option')
            m = build_model(_params(h, w, return_logits=True))
            m.load_state_dict(sd)
            pflib.profile(True)
            out = m.predict({k: v.cuda() for k, v in inp.items()}, None)
            labels = [r['label'] for r in pflib.profile_results()]
            pflib.profile(False)
            assert any(k in l for l in labels for k in ('conv_split', 'conv_s4', 'conv_front')) == bool(split), labels   # the kernels on fp16 pairs
            errs[split] = (out['orig_size_logits'].cpu() - ref['orig_size_logits']).abs().max().item()
            assert (out['seg'].cpu().long() == ref['seg']).float().mean().item() >= 0.999
    finally:
        L.pf_set_option(b'split_bf16', 1)
    assert errs[0] <= 1e-4, errs
    assert errs[1] <= 1e-4, errs


@pytest.mark.parametrize('b,split,term', [(1, 1, 'short'), (1, 0, 'short'), (2, 1, 'short'), (4, 1, 'short'), (8, 1, 'short'),
                                          (16, 1, 'short'), (16, 0, 'short'), (3, 1, 'short'), (16, 1, 'mid'), (2, 1, 'mid'),
                                          (32, 1, 'short'), (32, 0, 'short')],
                         ids=['B1_split', 'B1_fp32', 'B2_split', 'B4_split', 'B8_split', 'B16_split', 'B16_fp32', 'B3_nearest_row',
                              'B16_mid_term', 'B2_mid_term', 'B32_headline_sub_batch', 'B32_fp32'])
def test_full_size_timed_configuration_vs_oracle(b, split, term):
    """Every configuration bench.py times (1024x2048; the B = 1, 2, 4, 8, 16, 32 rows of csrc/conv_tuned.inc and
    conv_s4_tuned.inc - headline sub-batches of 32 and the by_batch legs, incl. the conv_pair launches B = 1 selects - split
    kernels on and off; `mid` = BASELINE
    configs[2]: gap_len 9 with the predicted-odometry ego chain, pc_transform_dataset.py:156-186) against the oracle
    pipeline at its own size: logits of the first and the last frame of the batch, argmax agreement and bit-exact warped
    inputs.  The per-layer kernel choice is keyed on (shape, B), so the small-size tests above never execute these table
    rows.  B=3 is not a measured batch size: it takes the rows of the nearest one (4); B=32 - the sub-batch of the
    headline (128 resident frames on 4 streams) - has rows of its own (measured at 32)."""
    from panoptic_forecasting_amd import lib as pflib
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.registry import build_model
    h, w = 1024, 2048
    tol = 1e-4
    kw = dict(gap_len=3) if term == 'short' else dict(gap_len=9, predicted=True)
    sd = _sd()
    m = build_model(_params(h, w, return_logits='orig', split_f16=split, emulate_disk_hop=True, seg_is_label_id=True,
                            per_sample_sentinel=True))
    m.load_state_dict(sd)
    parts = [synth.make_inputs(b=1, h=h, w=w, seed=40 + i, **kw) for i in range(b)]
    inp = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}
    pflib.profile(True)
    out = m.predict({k: v.cuda() for k, v in inp.items()}, None)
    labels = [r['label'] for r in pflib.profile_results()]
    pflib.profile(False)
    assert any(k in l for l in labels for k in ('conv_split', 'conv_s4', 'conv_front')) == bool(split), labels
    assert m.bg.range_reruns == 0 and m.bg.range_status() == 0
    for i in sorted({0, b - 1}):
        ref, seg_w, dep_w = oracle_pipeline(sd, parts[i], h, w)
        assert torch.equal(torch.from_numpy(synth.ID2TRAINID)[out['warped_seg'][i:i + 1].cpu().long()].long(), seg_w)
        err = (out['orig_size_logits'][i:i + 1].cpu() - ref['orig_size_logits']).abs().max().item()
        agree = (out['seg'][i:i + 1].cpu().long() == ref['seg']).float().mean().item()
        assert err <= tol, (i, err)
        assert agree >= 0.999, (i, agree)


@pytest.mark.parametrize('h,w,b,kw', [(512, 1024, 16, {}), (256, 512, 16, {}), (384, 768, 6, {}), (768, 1536, 3, {}),
                                      (512, 1024, 16, {'split_f16': 0}), (384, 768, 6, {'split_f16': 0}), (520, 1040, 2, {})],
                         ids=['512x1024_B16_rows_of_B4', '256x512_B16_rows_of_B1', '384x768_B6_rows_of_B1', '768x1536_B3_rows_of_B2',
                              '512x1024_B16_strict_fp32', '384x768_B6_strict_fp32', '520x1040_B2_widths_not_multiples_of_32'])
def test_second_resolution_heuristic_kernel_choice_vs_oracle(h, w, b, kw):
    """Image sizes the kernel tables were not measured at; 512x1024, B = 16 is the configuration of bench.py's
    `other_resolution` leg.  No row of csrc/conv_s4_tuned.inc / conv_tuned.inc has these shapes: every layer's kernel comes from
    conv_select.cpp's rule for unmeasured sizes (the row of the same layer whose measured launch had as many pixels - the B = 4
    rows for 512x1024 at B = 16, B = 1 for 256x512 at 16 and 384x768 at 6 (0.84), B = 2 for 768x1536 at 3 (1.7) - else the
    cost model, the reuse-vs-occupancy rule and the `conv_split` / `conv_s4` defaults): that choice is held to the same 1e-4
    as the measured one, first and last frame of the batch, warped inputs bit-exact.  `strict_fp32`: the same with split_f16 = 0
    (the fp32-MFMA families, i.e. the path a flagged forward is re-run on, under the same rule).  520x1040: level widths 520 /
    260 (multiples of 4, not of 32: partial 32-pixel tiles on the reused shapes) and 130 / 65 / 32 (the generic kernels)."""
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.registry import build_model
    sd = _sd()
    m = build_model(_params(h, w, return_logits='orig', emulate_disk_hop=True, seg_is_label_id=True, per_sample_sentinel=True, **kw))
    m.load_state_dict(sd)
    parts = [synth.make_inputs(b=1, h=h, w=w, seed=60 + i, gap_len=3) for i in range(b)]
    inp = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}
    out = m.predict({k: v.cuda() for k, v in inp.items()}, None)
    assert m.bg.range_reruns == 0 and m.bg.range_status() == 0
    for i in (0, b - 1):
        ref, seg_w, dep_w = oracle_pipeline(sd, parts[i], h, w)
        assert torch.equal(torch.from_numpy(synth.ID2TRAINID)[out['warped_seg'][i:i + 1].cpu().long()].long(), seg_w)
        err = (out['orig_size_logits'][i:i + 1].cpu() - ref['orig_size_logits']).abs().max().item()
        agree = (out['seg'][i:i + 1].cpu().long() == ref['seg']).float().mean().item()
        assert err <= 1e-4, (i, err)
        assert agree >= 0.999, (i, agree)


def test_batch_pinned_kernel_table_makes_logits_batch_invariant():
    """conv_table_batch pins the per-layer kernel choice: the same frame alone and inside a batch of 4 gives bit-identical
    logits (without the pin the B=1 and B=4 rows of the tuned table may pick kernels that differ in the last bits)."""
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.registry import build_model
    h, w = 256, 512
    sd = _sd()
    m = build_model(_params(h, w, return_logits='orig', conv_table_batch=4, per_sample_sentinel=True))
    m.load_state_dict(sd)
    inp = {k: v.cuda() for k, v in synth.make_inputs(b=4, h=h, w=w, seed=9).items()}
    # per-sample sentinel: the reference's batch-global max+1 (pc_transform_model.py:105) would couple the samples
    m4 = m.predict(inp, None)['orig_size_logits']
    one = {k: v[2:3].contiguous() for k, v in inp.items()}
    m1 = m.predict(one, None)['orig_size_logits']
    assert torch.equal(m4[2:3], m1)


def test_predict_on_never_seen_camera_tensors_does_not_synchronise():
    """The reference-shaped loop (export_cityscapes_segmentation_results.py:75-85: batch from the loader -> batch2gpu ->
    predict) makes FRESH camera tensors every batch, so the model's per-tensor inverse cache never hits.  With
    `add_camera_inverses` on the host batch (what export_bg.py does) predict() finds K^-1 / E^-1 in the batch: every
    iteration runs under torch's sync-debug mode 'error' - any device->host read or stream synchronisation inside predict
    would raise - and the result equals the one the model computes from its own (synchronising) inverses, bit for bit."""
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.pc_transform_model import add_camera_inverses
    from panoptic_forecasting_amd.registry import build_model
    h, w = 128, 256
    m = build_model(_params(h, w, return_logits='orig'))
    m.load_state_dict(_sd())
    host = [synth.make_inputs(b=2, h=h, w=w, seed=s_, gap_len=3) for s_ in (31, 32, 33)]
    host = [{k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b_.items()} for b_ in host]
    want = []
    for b_ in host:                                    # the model's own inverses (cache miss: device -> host -> LAPACK)
        o = m.predict({k: v.cuda() for k, v in b_.items()}, None)
        want.append((o['seg'].clone(), o['orig_size_logits'].clone(), o['warped_depth'].clone()))
    m.bg.settle()
    torch.cuda.synchronize()
    outs = []
    torch.cuda.set_sync_debug_mode('error')
    try:
        for b_ in host:
            dev_batch = {k: v.cuda(non_blocking=True) for k, v in add_camera_inverses(b_).items()}   # new tensors every batch
            outs.append(m.predict(dev_batch, None))    # enqueues only
    finally:
        torch.cuda.set_sync_debug_mode('default')
    for o, (seg, logit, wd) in zip(outs, want):
        assert torch.equal(o['seg'], seg) and torch.equal(o['orig_size_logits'], logit)
        assert torch.equal(o['warped_depth'].view(torch.int32), wd.view(torch.int32))
