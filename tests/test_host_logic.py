"""Host-side pieces either side of the device path, against the reference's own outputs (fixtures)."""
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import hop
from panoptic_forecasting_amd import ego, hardnet_arch as arch, packing, pq, synth


def test_ego_chain_matches_reference_data_utils():
    z = np.load(os.path.join(GOLDEN, 'g1_ego.npz'))
    assert np.array_equal(ego.camera_extrinsics(synth.CS_EXTRINSIC), z['E'])
    assert np.array_equal(ego.intrinsics_matrix(synth.CS_FX, synth.CS_FY, synth.CS_U0, synth.CS_V0), z['K'])
    for (speed, yaw, dt), want in zip(z['step_cases'], z['steps']):
        assert np.array_equal(ego.now_T_prev(speed, yaw, dt), want)
    # cumulative product order of pc_transform_dataset.py:219-231 (target 19, inputs [10,13,16])
    steps = [ego.now_T_prev(10.0 + 0.1 * k, 0.02 - 0.001 * k, 1 / 17.0 + 1e-4 * k) for k in range(1, 30)]
    cum = ego.cumulative_target_T(steps[:19])
    assert np.array_equal(cum[np.array([10, 13, 16])], z['chain'])


def test_hop_restatement_matches_fixture():
    z = np.load(os.path.join(GOLDEN, 'g2_glue.npz'))
    q = hop.export_depth_u16(torch.from_numpy(z['depth_in']))
    assert np.array_equal(q, z['depth_u16'])
    d, m = hop.load_depth(q)
    assert np.array_equal(d.numpy().view(np.uint32), z['depth_dec'].view(np.uint32))
    assert np.array_equal(m.numpy(), z['mask_dec'])
    oh = hop.onehot(torch.from_numpy(z['seg']))
    assert np.array_equal(oh.numpy().astype(np.uint8), z['onehot'])


def test_label_table_is_consistent():
    lut = hop.id2trainid_lut()
    assert np.array_equal(lut, synth.ID2TRAINID)
    assert np.array_equal(lut[synth.TRAINID2ID], np.arange(19))
    assert (lut[:34] == 255).sum() == 34 - 19


def test_bn_fold_equals_conv_then_bn():
    g = torch.Generator().manual_seed(5)
    sd = {'p.conv.weight': torch.randn(6, 5, 3, 3, generator=g), 'p.norm.weight': torch.rand(6, generator=g) + 0.5,
          'p.norm.bias': torch.randn(6, generator=g), 'p.norm.running_mean': torch.randn(6, generator=g),
          'p.norm.running_var': torch.rand(6, generator=g) + 0.5}
    w, b = packing.fold_conv_bn(sd, 'p')
    x = torch.randn(2, 5, 9, 11, generator=g)
    ref = torch.nn.functional.batch_norm(torch.nn.functional.conv2d(x, sd['p.conv.weight'], padding=1),
                                         sd['p.norm.running_mean'], sd['p.norm.running_var'], sd['p.norm.weight'],
                                         sd['p.norm.bias'], training=False, eps=1e-5)
    got = torch.nn.functional.conv2d(x, w, b, padding=1)
    assert (got - ref).abs().max() < 1e-5


def test_blob_roundtrip_header_and_tables():
    import struct
    sd = synth.make_state_dict(seed=3)
    blob = packing.pack_blob(sd, 36, 11)
    magic, ver, n_t, n_o, in_ch, n_cls, _, t_off, o_off, w_off, total = struct.unpack_from('<8sIIIIII4Q', blob, 0)
    spec = arch.Spec(36, 11)
    assert magic == packing.MAGIC and ver == packing.VERSION and total == len(blob)
    assert (n_t, n_o, in_ch, n_cls) == (len(spec.tensors), len(spec.ops), 36, 11)
    n_w = sum(o.cout * o.cin * o.k * o.k + o.cout for o in spec.conv_ops())
    assert len(blob) - w_off == 4 * n_w
    assert len(spec.conv_ops()) == 70 and n_w - sum(o.cout for o in spec.conv_ops()) + 0 > 4_000_000


def test_pq_identity_and_known_case():
    gt = torch.zeros(1, 4, 8, dtype=torch.long)
    gt[:, :, 4:] = 1
    gt[:, 0, 0] = 255                      # void pixel is ignored
    acc = pq.pq_accumulate(gt.clone().clamp(max=1), gt, 3)
    r = pq.pq_from_acc(acc)
    assert abs(r['pq'] - 100.0) < 1e-9 and r['n_classes'] == 2
    pred = gt.clone().clamp(max=1)
    pred[:, :, 4:6] = 0                    # class 1 loses half its area -> IoU 0.5 (not > 0.5): FP+FN; class 0 IoU=15/23
    r2 = pq.pq_from_acc(pq.pq_accumulate(pred, gt, 3))
    want0 = (15.0 / 23.0) / 1.0            # TP for class 0 (IoU>0.5)
    want1 = 0.0
    assert abs(r2['pq'] - 100.0 * (want0 + want1) / 2) < 1e-9


def test_pq_accumulators_add_across_shards():
    g = torch.Generator().manual_seed(0)
    pred = torch.randint(0, 11, (6, 32, 64), generator=g)
    gt = torch.randint(0, 12, (6, 32, 64), generator=g)
    gt[gt == 11] = 255
    whole = pq.pq_accumulate(pred, gt, 11)
    parts = pq.pq_accumulate(pred[0::2], gt[0::2], 11) + pq.pq_accumulate(pred[1::2], gt[1::2], 11)
    assert torch.equal(whole[:, 1:], parts[:, 1:])                 # integer counts add exactly
    assert (whole[:, 0] - parts[:, 0]).abs().max() < 1e-12


def test_pretrained_checkpoint_path_matches_reference(tmp_path):
    """params['model']['hardnet']['pretrain_path'] (hardnet.py:390-400, bg_model.py:45-48): the 19-class / RGB-stem pickle is
    loaded with its ``module.`` prefix stripped, the stem is averaged over RGB and tiled to the 36 input channels, the
    19-class head is replaced by a fresh one.  g7_pretrained.npz holds what the reference's own BGModel.__init__ made of the
    same synthetic pickle (tests/golden/make_golden_pretrained.py): every tensor but the re-initialised head must agree."""
    from panoptic_forecasting_amd.bg_model import BGModel
    path = str(tmp_path / 'hardnet70_pretrained.pkl')
    torch.save(synth.make_pretrained_checkpoint(seed=77), path)
    params = {'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])]},
              'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True, 'hardnet': {'pretrain_path': path}}}
    sd = BGModel(params).state_dict()
    g = np.load(os.path.join(GOLDEN, 'g7_pretrained.npz'))
    keys = [str(k) for k in g['keys']]
    assert sorted(k for k in sd if not k.startswith('model.finalConv')) == keys
    for k, (s1, s2) in zip(keys, g['sums']):
        v = sd[k].double()
        assert abs(float(v.sum()) - s1) <= 1e-9 * max(1.0, abs(s1)), k
        assert abs(float((v ** 2).sum()) - s2) <= 1e-9 * max(1.0, abs(s2)), k
    assert np.array_equal(sd['model.base.0.conv.weight'].numpy(), g['stem'])
    assert tuple(sd['model.finalConv.weight'].shape) == tuple(g['final_shape'])


def test_split_operand_bound():
    """The operand bound DESIGN.md / conv_mfma.h / bench.py's `dtype` state for the two-term fp16 split
    (hi = fp16_rne(x), mid = fp16_rne(x - hi); numpy's float16 cast is the same round-to-nearest-even as v_cvt_pk_f16_f32):
        |x - hi - mid| <= 2^-23 |x|   for 2^-2 <= |x| <= 65504        |x - hi - mid| <= 2^-25   below
    Every one of the 2^23 significands is tried at the bottom, in the middle and at the top of the range (the bound depends
    on the significand pattern, and on the exponent only through fp16's subnormal floor), plus a log-uniform sample."""
    def err(x):
        hi = x.astype(np.float16)
        r = x - hi.astype(np.float32)                    # exact in fp32
        mid = r.astype(np.float16)
        return np.abs(x.astype(np.float64) - hi.astype(np.float64) - mid.astype(np.float64))

    man = np.arange(1 << 23, dtype=np.uint32)
    worst_rel = 0.0
    for e in (-2, 0, 7, 15):
        x = ((np.uint32(127 + e) << np.uint32(23)) | man).view(np.float32)
        x = x[x <= 65504.0]
        rel = (err(x) / x.astype(np.float64)).max()
        worst_rel = max(worst_rel, rel)
        assert rel <= 2.0 ** -23, (e, rel)
    assert worst_rel > 2.0 ** -24          # the bound is tight to within a factor of two: do not quote a better one
    for e in (-3, -8, -14, -20, -30):      # hi and/or mid in fp16's subnormal range: absolute bound
        x = ((np.uint32(127 + e) << np.uint32(23)) | man[::8]).view(np.float32)
        assert err(x).max() <= 2.0 ** -25, e
    rng = np.random.default_rng(0)
    x = (np.exp2(rng.uniform(-30, 16, 1 << 22)) * rng.choice([-1.0, 1.0], 1 << 22)).astype(np.float32)
    x = x[np.abs(x) <= 65504.0]
    assert (err(x) <= 2.0 ** -23 * np.abs(x.astype(np.float64)) + 2.0 ** -25).all()
    # the round-2 scheme (both terms toward zero) for comparison: 2^-21
    # the largest representable magnitude is 65504 itself; 65520 rounds hi to inf, which is why the kernels guard |x| <= 65504
    with np.errstate(over='ignore'):
        assert np.isinf(np.float32(65520.0).astype(np.float16))
    assert np.float32(65519.0).astype(np.float16) == np.float16(65504.0)


def test_lazy_result_checks_once_on_first_value_access():
    """bg_model.LazyResult (the dict predict() returns): listing keys waits for nothing; the first access of a VALUE settles the
    forward behind it exactly once - through every accessor a caller of the reference's result dict may use."""
    from panoptic_forecasting_amd.bg_model import LazyResult

    class FakeModel:
        def __init__(self):
            self.resolved = []

        def _resolve(self, token):
            self.resolved.append(token)

    import copy
    import pickle
    assert not issubclass(LazyResult, dict)          # CPython's dict fast paths would skip the accessors (dict(r), {**r}, r | x)
    for access in (lambda r: r['seg'], lambda r: r.get('seg'), lambda r: list(r.items()), lambda r: list(r.values()),
                   lambda r: r.copy(), lambda r: r.pop('seg'), lambda r: dict(r), lambda r: {**r}, lambda r: r | {'x': 0},
                   lambda r: {'x': 0} | r, lambda r: {}.update(r), lambda r: pickle.dumps(r), lambda r: copy.copy(r),
                   lambda r: r == {'seg': 1}, lambda r: r.setdefault('seg', 5), lambda r: r.popitem(),
                   lambda r: [v for _, v in r.items()]):
        m = FakeModel()
        r = LazyResult({'seg': 1, 'orig_size_logits': 2}, m, 'token')
        assert sorted(r.keys()) == ['orig_size_logits', 'seg'] and 'seg' in r and len(r) == 2 and sorted(r) == sorted(r.keys())
        assert m.resolved == []                      # nothing waited for yet
        access(r)
        access(LazyResult({'seg': 1, 'orig_size_logits': 2}, m, None))   # a result without a token (stream capture, policy 'ignore') never resolves
        r.get('orig_size_logits')
        assert m.resolved == ['token'], m.resolved   # exactly once
    settled = dict(LazyResult({'seg': 1}, FakeModel(), 'token'))
    assert type(settled) is dict and settled == {'seg': 1}
    assert type(pickle.loads(pickle.dumps(LazyResult({'seg': 1}, FakeModel(), 't')))) is dict


def test_inverse_cache_keys_on_storage_and_version_and_checks_host_tensors_by_content():
    """pc_transform_model.InverseCache: a hit on (storage, version) returns the cached host inverse; an in-place edit bumps the
    version and misses.  HOST tensors are also compared by content on every hit (no stream is involved), so a write that
    bypasses the version counter (`K.data.copy_`, numpy-aliased memory) is caught; for DEVICE tensors that comparison would
    cost a stream synchronisation per predict and stays opt-in (`verify`; exercised on the GPU in test_gpu_warp_splat.py)."""
    import numpy as np
    from panoptic_forecasting_amd.pc_transform_model import InverseCache
    K = torch.tensor([[[2.0, 0.0, 1.0], [0.0, 4.0, 2.0], [0.0, 0.0, 1.0]]])
    eye = torch.eye(3).expand(1, 3, 3)
    c = InverseCache(verify=False)
    a = c(K)
    assert torch.allclose(a @ K, eye) and c(K) is a
    K.mul_(2.0)                                       # in-place: _version changes
    b = c(K)
    assert b is not a and torch.allclose(b @ K, eye, atol=1e-6)
    K.data.copy_(K.data * 0.5)                         # bypasses the version counter: caught by the host content check
    d = c(K)
    assert d is not b and torch.allclose(d @ K, eye, atol=1e-6)
    arr = np.array([[[2.0, 0.0, 1.0], [0.0, 4.0, 2.0], [0.0, 0.0, 1.0]]], np.float32)
    Kn = torch.from_numpy(arr)                         # numpy-aliased camera matrix edited behind torch's back
    first = c(Kn)
    arr *= 4.0
    again = c(Kn)
    assert again is not first and torch.allclose(again @ Kn, eye, atol=1e-6)


def test_add_camera_inverses_gives_the_models_own_bits_and_leaves_device_batches_alone():
    """pc_transform_model.add_camera_inverses (what export_bg.py calls before the batch moves to the device): K^-1 / E^-1 =
    torch.inverse of the HOST tensors = bit for bit what predict() would compute for itself (host_inverse: the reference's
    torch.inverse, pc_transform_model.py:51,71), as new keys of a shallow copy; inverses already in the batch are kept."""
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.pc_transform_model import add_camera_inverses, host_inverse
    inp = synth.make_inputs(b=2, h=16, w=32, seed=3)
    out = add_camera_inverses(inp)
    assert out is not inp and 'intrinsics_inv' not in inp
    assert torch.equal(out['intrinsics_inv'].view(torch.int32), host_inverse(inp['intrinsics']).view(torch.int32))
    assert torch.equal(out['extrinsics_inv'].view(torch.int32), host_inverse(inp['extrinsics']).view(torch.int32))
    # LAPACK returns column-major views: left like that, every predict() would enqueue one copy kernel per matrix
    assert not torch.inverse(inp['intrinsics']).is_contiguous()
    for m in (out['intrinsics_inv'], out['extrinsics_inv'], host_inverse(inp['intrinsics']), host_inverse(inp['extrinsics'])):
        assert m.is_contiguous()
    assert out['depth'] is inp['depth']
    marked = dict(inp, intrinsics_inv=torch.zeros(2, 3, 3))
    assert add_camera_inverses(marked)['intrinsics_inv'] is marked['intrinsics_inv']
    only_seg = {'seg': inp['seg']}
    assert add_camera_inverses(only_seg) is only_seg                # no cameras: untouched


def test_backend_description_single_process():
    from panoptic_forecasting_amd import dist as pfdist
    assert pfdist.backend_description() == 'single process'


def test_tuned_tables_key_each_layer_to_one_resolution():
    """conv_select.cpp::measured_geometry looks a layer (ks, cin, cout) up at the resolution it was MEASURED at: both shape tables
    must hold every layer at exactly one (hout, wout)."""
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'panoptic-forecasting_amd', 'csrc')
    row = re.compile(r'^\s*\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), \{')
    for name in ('conv_s4_tuned.inc', 'conv_tuned.inc'):
        sizes, n = {}, 0
        with open(os.path.join(csrc, name)) as f:
            for line in f:
                m = row.match(line)
                if not m:
                    continue
                ks, cin, cout, h, w, b = map(int, m.groups())
                sizes.setdefault((ks, cin, cout), set()).add((h, w))
                n += 1
        assert n > 50, name
        dup = {k: v for k, v in sizes.items() if len(v) != 1}
        assert not dup, (name, dup)
