"""HIP warp/splat (through the C ABI) vs the golden fixtures and the C oracle — bit-exact."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), 'golden')
FILES = sorted(glob.glob(os.path.join(G, 'g1_*x*.npz')))


def _model(ind, is_img):
    from panoptic_forecasting_amd.pc_transform_model import PCTransformModel
    return PCTransformModel({'model': {'only_this_ind': ind, 'is_img': is_img}})


def _inputs(z, img, dev):
    d = {'intrinsics': z['K'], 'extrinsics': z['E'], 'target_T': z['T'], 'depth': z['depth'],
         'depth_mask': z['mask'], 'seg': z['img'] if img else z['seg'],
         'intrinsics_inv': z['Kinv'], 'extrinsics_inv': z['Einv']}
    return {k: torch.from_numpy(v).to(dev) for k, v in d.items()}


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f) for f in FILES])
@pytest.mark.parametrize('ind', [None, 0, 1, 2])
@pytest.mark.parametrize('is_img', [False, True])
def test_hip_matches_reference_fixture(path, ind, is_img):
    z = np.load(path)
    out = _model(ind, is_img).predict(_inputs(z, is_img, 'cuda'), None)
    tag = '%s_%d' % ('all' if ind is None else str(ind), int(is_img))
    assert np.array_equal(out['seg'].cpu().numpy(), z['seg_' + tag])
    assert np.array_equal(out['depth'].cpu().numpy().view(np.uint32), z['depth_bits_' + tag])
    if not is_img:
        assert np.array_equal(out['result2d'].cpu().numpy(), z['result2d_' + tag].astype(np.int64))


@pytest.mark.parametrize('cfg', [dict(h=256, w=512, b=2, gap_len=3, depth_mode='scene'),
                                 dict(h=256, w=512, b=1, gap_len=9, depth_mode='uniform', predicted=True),
                                 dict(h=200, w=333, b=3, gap_len=3, depth_mode='uniform'),
                                 dict(h=128, w=256, b=1, identity=True, depth_mode='uniform')])
@pytest.mark.parametrize('ind', [None, 1])
def test_hip_matches_oracle(cfg, ind):
    from oracle import warp_splat as oracle
    from panoptic_forecasting_amd import synth
    inp = synth.make_inputs(seed=11, **cfg)
    ref = oracle.predict(inp, only_this_ind=ind, debug=False)
    out = _model(ind, False).predict({k: v.cuda() for k, v in inp.items()}, None)
    assert torch.equal(out['result2d'].cpu(), ref['result2d'])
    assert torch.equal(out['seg'].cpu(), ref['seg'])
    assert torch.equal(out['depth'].cpu().view(torch.int32), ref['depth'].view(torch.int32))


def test_per_frame_mode_equals_three_single_calls():
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.pc_transform_model import WarpSplat
    inp = {k: v.cuda() for k, v in synth.make_inputs(b=2, h=128, w=256, seed=3, depth_mode='uniform').items()}
    ws = WarpSplat()
    seg, dep, _ = ws(inp['depth'], inp['depth_mask'], inp['seg'], inp['intrinsics'], inp['extrinsics'],
                     inp['target_T'], per_frame=True)
    for t in range(3):
        o = _model(t, False).predict(inp, None)
        assert torch.equal(seg[:, t], o['seg'])
        assert torch.equal(dep[:, t].view(torch.int32), o['depth'].view(torch.int32))


@pytest.mark.parametrize('term', [dict(gap_len=3), dict(gap_len=9, predicted=True)], ids=['short', 'mid_predicted'])
@pytest.mark.parametrize('ind', [None, 0, 2])
def test_full_size_bit_exact_vs_oracle(term, ind):
    """1024x2048 (BASELINE configs[1] / configs[2]) against the C oracle, bit for bit: result2d, seg, depth.  The oracle
    needs about half a second per splat at this size."""
    from oracle import warp_splat as oracle
    from panoptic_forecasting_amd import synth
    inp = synth.make_inputs(b=1, seed=5, **term)
    ref = oracle.predict(inp, only_this_ind=ind)
    out = _model(ind, False).predict({k: v.cuda() for k, v in inp.items()}, None)
    assert torch.equal(out['result2d'].cpu(), ref['result2d'])
    assert torch.equal(out['seg'].cpu(), ref['seg'])
    assert torch.equal(out['depth'].cpu().view(torch.int32), ref['depth'].view(torch.int32))


def test_full_size_per_frame_batch_vs_oracle():
    """The launch shape bench.py times: per-frame z-buffers for a batch of frames in one call (B=4 here)."""
    from oracle import warp_splat as oracle
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.pc_transform_model import WarpSplat
    parts = [synth.make_inputs(b=1, seed=70 + i) for i in range(4)]
    inp = {k: torch.cat([p[k] for p in parts], 0).cuda() for k in parts[0]}
    seg, dep, _ = WarpSplat()(inp['depth'], inp['depth_mask'], inp['seg'], inp['intrinsics'], inp['extrinsics'],
                              inp['target_T'], per_frame=True, want_result2d=False, per_sample_sentinel=True)
    for i in (0, 3):
        for t in range(3):
            ref = oracle.predict(parts[i], only_this_ind=t)
            assert torch.equal(seg[i, t].cpu(), ref['seg'][0])
            assert torch.equal(dep[i, t].cpu().view(torch.int32), ref['depth'][0].view(torch.int32))


def test_full_size_properties():
    """1024x2048 size-independent properties: determinism, every output depth is either -1, the sentinel, or the z of
    a valid source, identity warp reproduces the input exactly where the mask holds."""
    from panoptic_forecasting_amd import synth
    inp = {k: v.cuda() for k, v in synth.make_inputs(b=1, seed=0).items()}
    m = _model(None, False)
    a = m.predict(inp, None)
    b = m.predict(inp, None)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    d = a['depth']
    holes = d == -1
    assert 0 < holes.float().mean() < 0.9
    assert (a['seg'][holes] == 0).all()
    assert (d[~holes] > 0).all()
    ident = {k: v.cuda() for k, v in synth.make_inputs(b=1, h=512, w=1024, seed=1, identity=True,
                                                       mask_p=1.0, depth_mode='uniform').items()}
    o = _model(2, False).predict(ident, None)
    # identity ego-motion: every pixel lands within one pixel of itself; the nearest of <=4 candidates wins
    assert (o['depth'] > 0).float().mean() > 0.99


@pytest.mark.parametrize('case', ['non_affine_camera', 'skewed_camera', 'first_column_negative_focal', 'huge_depth', 'list_overflow'])
def test_paths_beside_the_fast_one_match_oracle(case):
    """bin_kernel takes the packed shortcut chain only for affine cameras and finite intermediates, and the raster kernel
    reads per-tile lists unless one overflowed.  The other branches against the C oracle, bit for bit:
      non_affine_camera  last rows of K / E that are not (0,0,1) / (0,0,0,1): every point on the full scalar chain;
      skewed_camera      K with a skew term (K[0][1] != 0, so K^-1[0][1] != 0 too): the shortcut chain skips the products with
                         the exactly-zero entries of a pinhole camera and is taken only when they ARE zero - this camera runs the
                         full chain;
      first_column_negative_focal  fx < 0 (a mirrored image): K^-1[0][0] * u is -0 in pixel column 0, the one place where the
                         dropped `0 +` of the dot products changes an intermediate (the sign of a zero) - outputs must not move;
      huge_depth         finite depths around 1e37..3e38 scattered in the maps: intermediates overflow to inf (and inf - inf =
                         NaN in later sums), so those lanes leave the shortcut; both sides clamp the resulting coordinates
                         with fmaxf/fminf before the integer conversion, which is defined for NaN and inf;
      list_overflow      depth planes that alternate between 2 m and 80 m per pixel at a large ego translation: every
                         source tile's box covers a wide band, destination lists exceed their 64 entries and the kernel falls
                         back to testing all boxes."""
    from oracle import warp_splat as oracle
    from panoptic_forecasting_amd import synth
    if case == 'list_overflow':
        inp = synth.make_inputs(b=1, h=512, w=1024, seed=5, gap_len=9, depth_mode='uniform')
        g = torch.Generator().manual_seed(1)
        near = torch.rand(inp['depth'].shape, generator=g) < 0.5
        inp['depth'] = torch.where(near, torch.full_like(inp['depth'], 2.0), torch.full_like(inp['depth'], 80.0))
        inp['target_T'][:, :, 0, 3] += 1.5          # metres sideways: 1700 px of parallax at 2 m, 40 px at 80 m
    else:
        inp = synth.make_inputs(b=2, h=128, w=256, seed=21, gap_len=3, depth_mode='uniform')
    if case == 'non_affine_camera':
        inp['intrinsics'][:, 2, 0] = 1e-5
        inp['extrinsics'][:, 3, 2] = 1e-3
    if case == 'skewed_camera':
        inp['intrinsics'][:, 0, 1] = 3.5
    if case == 'first_column_negative_focal':
        inp['intrinsics'][:, 0, 0] *= -1.0
    if case == 'huge_depth':
        g = torch.Generator().manual_seed(2)
        pick = torch.rand(inp['depth'].shape, generator=g) < 0.02
        big = torch.empty(inp['depth'].shape).uniform_(1e37, 3e38, generator=g)
        inp['depth'] = torch.where(pick, big, inp['depth'])
    for ind in (None, 1):
        ref = oracle.predict(inp, only_this_ind=ind)
        out = _model(ind, False).predict({k: v.cuda() for k, v in inp.items()}, None)
        assert torch.equal(out['result2d'].cpu(), ref['result2d']), (case, ind)
        assert torch.equal(out['seg'].cpu(), ref['seg']), (case, ind)
        assert torch.equal(out['depth'].cpu().view(torch.int32), ref['depth'].view(torch.int32)), (case, ind)


@pytest.mark.gpu
def test_inverse_cache_finds_a_write_behind_the_version_counter_within_its_window():
    """Device camera tensors are keyed on (storage, version); a write that bypasses the counter (`K.data`, a foreign kernel) is not
    seen by the key.  Round 6: the first and then every `verify_every`-th hit compares contents (one synchronisation), so such a
    write is found within the window instead of never; `verify=True` checks every hit; inference tensors are never cached."""
    from panoptic_forecasting_amd.pc_transform_model import InverseCache
    K = torch.tensor([[[2.0, 0.0, 1.0], [0.0, 4.0, 2.0], [0.0, 0.0, 1.0]]], device='cuda')
    eye = torch.eye(3, device='cuda').expand(1, 3, 3)
    c = InverseCache(verify=False, verify_every=3)
    a = c(K)
    assert c(K) is a                                   # hit 0: checked, equal
    K.data.mul_(2.0)                                   # version counter untouched
    assert c(K) is a and c(K) is a                     # hits 1, 2: inside the window (documented)
    fresh = c(K)                                       # hit 3: compared, found stale, re-inverted
    assert fresh is not a and torch.allclose(fresh @ K, eye, atol=1e-6)
    strict = InverseCache(verify=True)
    s0 = strict(K)
    K.data.mul_(0.5)
    s1 = strict(K)
    assert s1 is not s0 and torch.allclose(s1 @ K, eye, atol=1e-6)
    with torch.inference_mode():
        Ki = K.clone()
        assert Ki.is_inference()
        i0 = c(Ki)
        assert torch.allclose(i0 @ Ki, eye, atol=1e-6) and c(Ki) is not i0
