"""Host logic of scope rows f2/f3 that needs no GPU: file names, flag logic, PNG round trips, missing-file fill, PQ."""
import os

import numpy as np
import torch

from panoptic_forecasting_amd import hop_io, pq


def test_names_keys_and_export_flag_logic():
    assert hop_io.LABEL_PNG % ('aachen', '000001', 19) == 'aachen_000001_000019_gtFine_labelIds.png'
    assert hop_io.DEPTH_PNG % ('ulm', '000033', 7) == 'ulm_000033_000007_depths.png'
    assert hop_io.h5_key('bonn', '000004', 19, (9 - 3) / 3) == 'bonn/000004/000019/2'     # float start_frame, %d
    assert hop_io.seg_mode() == hop_io.SEG_TRAINID_TO_ID
    assert hop_io.seg_mode(no_convert=True) == hop_io.SEG_AS_IS
    assert hop_io.seg_mode(no_convert=True, convert_to_trainid=True) == hop_io.SEG_ID_TO_TRAINID
    assert hop_io.seg_mode(convert_to_trainid=True) == hop_io.SEG_TRAINID_TO_ID            # the elif is never reached
    assert hop_io.seg_mode(is_img=True) == hop_io.SEG_AS_IS


def test_png_round_trips(tmp_path):
    g = np.random.default_rng(0)
    lab = g.integers(0, 256, (32, 48), dtype=np.uint8)
    q = g.integers(0, 65536, (32, 48), dtype=np.uint16)
    img = g.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    for name, a in (('l.png', lab), ('d.png', q), ('i.png', img)):
        p = str(tmp_path / name)
        hop_io.write_png(p, a)
        back = hop_io.read_png(p)
        assert back.dtype == a.dtype and np.array_equal(back, a)


def test_fill_missing(tmp_path):
    gt = tmp_path / 'gtFine' / 'val'
    (gt / 'ulm').mkdir(parents=True)
    (gt / 'bonn').mkdir(parents=True)
    for c, n in (('ulm', 'ulm_000001_000019_gtFine_labelIds.png'), ('ulm', 'ulm_000002_000019_gtFine_labelIds.png'),
                 ('bonn', 'bonn_000001_000019_gtFine_labelIds.png')):
        hop_io.write_png(str(gt / c / n), np.zeros((4, 8), np.uint8))
    out = tmp_path / 'out'
    (out / 'ulm').mkdir(parents=True)
    hop_io.write_png(str(out / 'ulm' / 'ulm_000001_000019_gtFine_labelIds.png'), np.ones((4, 8), np.uint8))
    bgdir = tmp_path / 'bg'
    (bgdir / 'ulm').mkdir(parents=True)
    hop_io.write_png(str(bgdir / 'ulm' / 'ulm_000002_000019_gtFine_labelIds.png'), np.full((4, 8), 2, np.uint8))
    n = hop_io.fill_missing(str(out), str(gt), background_dir=str(bgdir), shape=(4, 8))
    assert n == 2
    assert (hop_io.read_png(str(out / 'ulm' / 'ulm_000002_000019_gtFine_labelIds.png')) == 11).all()   # trainId 2 -> id 11
    assert (hop_io.read_png(str(out / 'bonn' / 'bonn_000001_000019_gtFine_labelIds.png')) == 0).all()
    assert (hop_io.read_png(str(out / 'ulm' / 'ulm_000001_000019_gtFine_labelIds.png')) == 1).all()    # untouched


def test_panoptic_pq_matching():
    g = torch.zeros(1, 8, 8, dtype=torch.long)
    g[0, :4] = 3
    g[0, 4:, :4] = 11000
    g[0, 4:, 4:] = 11001
    g[0, 0, 0] = 255
    assert abs(pq.pq_from_acc(pq.pq_accumulate_panoptic(g, g))['pq'] - 100.0) < 1e-12
    p = g.clone()
    p[0, 4:, 4:] = 11005          # same category, other instance id: still a match
    p[0, 4:, :2] = 2              # half of instance 11000 lost: IoU exactly 0.5 is not a match
    acc = pq.pq_accumulate_panoptic(p, g)
    assert acc[11].tolist() == [1.0, 1.0, 1.0, 1.0] and acc[2].tolist() == [0.0, 0.0, 1.0, 0.0]
    assert acc[3].tolist() == [1.0, 1.0, 0.0, 0.0]
    # a prediction lying mostly on void is not a false positive
    g2 = torch.full((1, 4, 4), 255, dtype=torch.long)
    p2 = torch.full((1, 4, 4), 12000, dtype=torch.long)
    assert pq.pq_accumulate_panoptic(p2, g2).sum() == 0
    # sharded accumulation adds exactly
    a = pq.pq_accumulate_panoptic(torch.cat([p, g]), torch.cat([g, g]))
    b = pq.pq_accumulate_panoptic(p, g) + pq.pq_accumulate_panoptic(g, g)
    assert torch.equal(a, b)
