"""Scope row f2: device hop kernels vs the oracle, and the FILE hop (PNG export -> load) vs the in-register hop."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('n', [0, 3, 4, 1021, 256 * 512 + 2])
@pytest.mark.parametrize('mode', [0, 1, 2])
@pytest.mark.parametrize('i64', [False, True])
def test_hop_export_matches_oracle(n, mode, i64):
    from oracle import hop as oh
    from panoptic_forecasting_amd import hop_io
    g = torch.Generator().manual_seed(n + mode)
    seg = torch.randint(0, 256, (n,), generator=g, dtype=torch.uint8)
    depth = torch.rand(n, generator=g) * 300.0 - 20.0
    if n > 8:
        depth[:8] = torch.tensor([-1.0, -1.5, 0.0, 254.0, 254.001953125, 1e9, -1e9, 0.001953125])   # clamp edges, .5 ties
    if n == 0:
        return   # nothing to launch; the entry returns PF_OK for n == 0 (checked on the host side)
    s_in = seg.long() if i64 else seg
    out_seg, q = hop_io.device_export(s_in.cuda(), depth.cuda(), mode)
    want_q = oh.export_depth_u16(depth)
    assert np.array_equal(hop_io.u16_numpy(q), want_q)
    if mode == 0:
        want = seg.numpy()
    elif mode == 1:
        lut = np.zeros(256, np.uint8)
        lut[:19] = [7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33]
        want = lut[seg.numpy()]
    else:
        lut = oh.id2trainid_lut()
        lut[34:] = 0
        want = lut[seg.numpy()]
    assert np.array_equal(out_seg.cpu().numpy(), want)


@pytest.mark.parametrize('n', [5, 4096, 64 * 128 * 3 + 1])
def test_hop_load_matches_oracle_and_golden(n):
    from oracle import hop as oh
    from panoptic_forecasting_amd import hop_io
    g = torch.Generator().manual_seed(n)
    q = torch.randint(0, 65536, (n,), generator=g).numpy().astype(np.uint16)
    q[:5] = [0, 255, 256, 257, 65535]
    d, m = hop_io.device_load_depth(torch.from_numpy(q.view(np.int16)).cuda(), 0.1, 200.0)
    wd, wm = oh.load_depth(q, 0.1, 200.0)
    assert torch.equal(d.cpu().view(torch.int32), wd.view(torch.int32))
    assert torch.equal(m.cpu(), wm)
    z = np.load(os.path.join(G, 'g2_glue.npz'))       # the committed glue fixture: export side then load side
    _, q2 = hop_io.device_export(None, torch.from_numpy(z['depth_in']).cuda())
    assert np.array_equal(hop_io.u16_numpy(q2), z['depth_u16'])
    d2, m2 = hop_io.device_load_depth(q2, 0.1, 200.0)
    assert np.array_equal(d2.cpu().numpy(), z['depth_dec']) and np.array_equal(m2.cpu().numpy(), z['mask_dec'])


def test_file_hop_equals_in_register_hop(tmp_path):
    """pc_transform x3 -> PNG files (labelIds + u16 depths) -> load -> task bg   ==   task bg_forecast, bit for bit."""
    from panoptic_forecasting_amd import hop_io, synth
    from panoptic_forecasting_amd.registry import build_model
    h, w, b = 128, 256, 2
    with open(os.path.join(G, 'calib_seed1234.json')) as f:
        sd = synth.make_state_dict(seed=1234, calib=json.load(f))
    base = {'no_gpu': False, 'load_model': None, 'load_best_model': False,
            'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])],
                     'min_depth': 0.1, 'max_depth': 200}}
    mp = {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True, 'final_h': h, 'final_w': w}
    inp = {k: v.cuda() for k, v in synth.make_inputs(b=b, h=h, w=w, seed=5, gap_len=3).items()}
    meta = {'city': ['aachen', 'bonn'], 'seq': ['000001', '000002'], 'frame': [19, 49], 'target_frame': [19, 49]}
    dirs = []
    for t in range(3):   # the reference runs the pc_transform export once per input frame (bg_val_short.yaml:12-14)
        m = build_model(dict(base, task='pc_transform', model={'only_this_ind': t, 'is_img': False}))
        preds = m.predict(inp, None)
        d = str(tmp_path / ('frame%d' % t))
        files = hop_io.export_batch(preds, meta, d, no_convert=True, convert_to_trainid=True, save_depth=True,
                                    save_depth_as_png=True)
        assert os.path.basename(files[0]) == 'aachen_000001_000019_gtFine_labelIds.png'
        dirs.append(d)
    samples = []
    for i in range(b):
        city, seq, fr = meta['city'][i], meta['seq'][i], meta['target_frame'][i]
        labels = [os.path.join(d, city, hop_io.LABEL_PNG % (city, seq, fr)) for d in dirs]
        depths = [os.path.join(d, city, hop_io.DEPTH_PNG % (city, seq, fr)) for d in dirs]
        assert hop_io.read_png(depths[0]).dtype == np.uint16
        samples.append(hop_io.load_bg_inputs(labels, depth_pngs=depths, min_depth=0.1, max_depth=200))
    batch = hop_io.collate(samples)
    # the file path hands the network dense int64 labels (stem -> three front-end kernels), the in-register path u8 labels (stem ->
    # conv_front.hip): with the fused front end switched off both run the same kernels on the same bits - equal bit for bit;
    # with the default plan the two front ends agree to the network's tolerance
    for fuse in (0, 1):
        bg = build_model(dict(base, task='bg', model=dict(mp, fuse_front=fuse)))
        bg.load_state_dict(sd)
        two_stage = bg.predict(batch, None)
        fused = build_model(dict(base, task='bg_forecast', model=dict(mp, return_logits=True, fuse_front=fuse)))
        fused.load_state_dict(sd)
        one = fused.predict(inp, None)
        if fuse == 0:
            assert torch.equal(two_stage['seg'].long(), one['seg'].long())
            assert torch.equal(two_stage['orig_size_logits'], one['orig_size_logits'])
        else:
            assert (two_stage['seg'].long() == one['seg'].long()).float().mean().item() >= 0.9999
            assert (two_stage['orig_size_logits'] - one['orig_size_logits']).abs().max().item() <= 1e-4


def test_export_driver_with_reference_flags(tmp_path):
    """export_bg.py = scripts/bg/run_export_bg_val.sh with the python path changed: --config_file / --load_model (with the
    config.yaml stored next to the checkpoint) / --no_convert / --export_name / --working_dir, on synthetic samples; the
    files it writes equal a direct model.predict + the reference's naming (export_cityscapes_segmentation_results.py:65-107)."""
    import json
    import yaml
    from panoptic_forecasting_amd import export_bg, synth
    from panoptic_forecasting_amd.registry import build_model
    h, w = 64, 128
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'calib_seed1234.json')) as f:
        sd = synth.make_state_dict(seed=1234, calib=json.load(f))
    ck = tmp_path / 'ck'
    ck.mkdir()
    stored = {'task': 'bg', 'data': {'num_classes': 11, 'depth_norm_params': None, 'min_depth': 0.1, 'max_depth': 200},
              'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True}}
    (ck / 'config.yaml').write_text(yaml.dump(stored))
    torch.save(sd, str(ck / 'bg_model.pt'))
    cfg = tmp_path / 'bg_val.yaml'
    cfg.write_text(yaml.dump({'task': 'bg', 'data': {'data_splits': ['val']}, 'model': {'final_w': w, 'final_h': h},
                              'training': {'batch_size': 2, 'num_data_workers': 0}}))
    written = export_bg.main(['--config_file', str(cfg), '--load_model', str(ck / 'bg_model.pt'), '--no_convert',
                              '--export_name', 'exported_predictions_short_trainids', '--working_dir', str(tmp_path),
                              '--synthetic', '3'])
    assert len(written) == 3
    params = dict(stored, no_gpu=False, load_model=None, load_best_model=False)
    params['model'] = dict(stored['model'], final_w=w, final_h=h)
    m = build_model(params)
    m.load_state_dict(sd)
    for i, path in enumerate(written):
        assert path == str(tmp_path / 'exported_predictions_short_trainids' / 'val' / 'synth' /
                           ('synth_%06d_000019_gtFine_labelIds.png' % i))
        inp = {k: v.cuda() for k, v in synth.make_bg_inputs(b=1, h=h, w=w, seed=i).items()}
        want = m.predict(inp, None)['seg'][0].cpu().numpy().astype(np.uint8)      # --no_convert: trainIds as predicted
        assert np.array_equal(np.array(Image.open(path)), want)
    # the default loop keeps three batches in flight (three model replicas on three streams, files of batch k written after batch k + 2 was
    # enqueued); the plain serial loop and a deeper pipeline write the same bytes
    for depth in (1, 2):
        again = export_bg.main(['--config_file', str(cfg), '--load_model', str(ck / 'bg_model.pt'), '--no_convert',
                                '--export_name', 'depth%d' % depth, '--working_dir', str(tmp_path), '--synthetic', '3',
                                '--pipeline_depth', str(depth)])
        assert len(again) == 3
        for a, b in zip(written, again):
            assert os.path.basename(a) == os.path.basename(b)
            with open(a, 'rb') as fa, open(b, 'rb') as fb:
                assert fa.read() == fb.read()
