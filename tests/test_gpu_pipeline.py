"""The whole chain on the device at a small size: 3 frames -> warp/splat -> HarDNet -> bg label map -> exported PNG ->
(read back as the fg stage's background) -> instance merge -> panoptic PNG + annotations -> PQ accumulators -> gather.
Every stage is checked against its oracle elsewhere; this test checks that the stages compose the way the reference's
scripts chain them (run_export_bg_val.sh -> run_export_fg_val_panoptics.sh -> run_fg_eval_panoptic.sh)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


def test_forecast_export_merge_encode_pq(tmp_path):
    from oracle import panoptic as op
    from panoptic_forecasting_amd import dist as pfdist
    from panoptic_forecasting_amd import hop_io, panoptic as pp, pq, synth
    from panoptic_forecasting_amd.registry import build_model
    h, w, b = 128, 256, 2
    with open(os.path.join(G, 'calib_seed1234.json')) as f:
        sd = synth.make_state_dict(seed=1234, calib=json.load(f))
    params = {'task': 'bg_forecast', 'no_gpu': False, 'load_model': None, 'load_best_model': False,
              'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])],
                       'min_depth': 0.1, 'max_depth': 200},
              'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True, 'final_h': h, 'final_w': w}}
    model = build_model(params)
    model.load_state_dict(sd)
    inp = {k: v.cuda() for k, v in synth.make_inputs(b=b, h=h, w=w, seed=8, gap_len=3).items()}
    out = model.predict(inp, None)

    # 1. export the bg forecast like export_results does (trainIds kept: --no_convert), read it back as fg background
    meta = {'city': ['ulm', 'ulm'], 'seq': ['000001', '000002'], 'frame': [19, 19], 'target_frame': [19, 19]}
    files = hop_io.export_batch({'seg': out['seg'], 'depth': out['warped_depth'][:, 2]}, meta, str(tmp_path / 'bg'), no_convert=True)
    background = torch.from_numpy(np.stack([hop_io.read_png(f) for f in files])).cuda()
    assert torch.equal(background, out['seg'])

    # 2. paste forecast instances (synthetic fg head outputs), depth sorted
    g = torch.Generator().manual_seed(2)
    counts = [5, 3]
    n = sum(counts)
    logits = torch.randn(n, 28, 28, generator=g) + 1.0
    boxes = torch.stack([torch.rand(n, generator=g) * w, torch.rand(n, generator=g) * h,
                         20 + torch.rand(n, generator=g) * 60, 15 + torch.rand(n, generator=g) * 40], 1)
    depths = 5 + torch.rand(n, generator=g) * 40
    classes = torch.randint(0, 8, (n,), generator=g)
    merger = pp.PanopticMerger({'model': {'use_depth_sorting': True}})
    res = merger.predict_panoptic({'masks': logits.cuda(), 'boxes': boxes.cuda(), 'depths': depths.cuda()},
                                  list(classes.cuda().split(counts)), background=background)
    seg = res['seg']
    want = op.merge(list(torch.sigmoid(logits).split(counts)), list(boxes.split(counts)), list(depths.split(counts)),
                    list(classes.split(counts)), h, w, background=background.cpu().long())
    assert torch.equal(seg.cpu(), want)
    assert (seg >= 11000).any() and (seg < 11).any()

    # 3. panoptic PNG + annotation JSON; the PNG decodes back to the converted ids
    ann = pp.export_panoptic(seg, meta, str(tmp_path), 'pan_val')
    path = pp.write_annotations(ann, str(tmp_path), 'pan_val')
    rec = json.load(open(path))['annotations']
    assert len(rec) == b and rec[0]['file_name'] == 'ulm_000001_000019_pred_panoptic.png'
    from PIL import Image
    ids0 = pp.decode_png(np.array(Image.open(os.path.join(str(tmp_path), 'pan_val', rec[0]['file_name']))))
    _, want_ids, present = op.encode(seg[0].cpu(), convert=True)
    assert np.array_equal(ids0, want_ids)
    assert sorted(s['id'] for s in rec[0]['segments_info']) == [i for i in present if i != 0]

    # 4. PQ of the merged map against itself and against a perturbed copy; accumulators gather like the sharded bench's
    acc = pq.pq_accumulate_panoptic(seg, seg)
    assert abs(pq.pq_from_acc(pfdist.gather_accumulators(acc).sum(0))['pq'] - 100.0) < 1e-9
    worse = seg.clone()
    worse[:, : h // 2] = 255
    assert pq.pq_from_acc(pq.pq_accumulate_panoptic(worse, seg))['pq'] < 100.0


def test_bench_two_ranks_on_one_gpu_prints_the_multi_rank_line():
    """`bench.py --gpus 2` end to end on a 1-GPU box: PF_BENCH_SHARE_GPU=1 lets both ranks use the one device over gloo (RCCL refuses
    two ranks per device), so the N > 1 code path really runs on the GPU - the launcher, the barrier-bracketed timed region with the
    max over ranks, the sharded PQ gather, per-rank times and device identities - and prints ONE line with n_gpus = 2.  Not a
    measurement (the line names gloo and two identical devices); the 8-GPU run is the driver's."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e['PF_BENCH_SHARE_GPU'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '16',
                        '--profile-steps', '1'], capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['value'] > 0
    assert len(d['per_rank_ms']) == 2 and len(d['devices']) == 2 and d['pq_gather_check']['ranks_gathered'] == 2
    assert 'gloo' in d['config']['backend'] and d['config']['world'] == 2
    assert d['cpu_baseline'] is None and d['by_batch'] is None          # rank 0 at N = 1 only
    assert d['range_overflow'] is False
