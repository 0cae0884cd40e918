#!/usr/bin/env python
"""Generate the golden fixtures by running the REFERENCE itself (build container only).

    python tests/golden/make_golden.py

Imports ``PCTransformModel`` / ``BGModel`` / ``data_utils`` from /root/reference (see
``_ref_import.py`` for the stubbing and the torch_scatter stand-in) and writes small
``.npz``/``.json`` fixtures next to this script.  The fixtures are data only (inputs and
the reference's outputs); no reference source travels.

  g1_*.npz   warp/splat: PCTransformModel.predict for only_this_ind in {0,1,2,None},
             is_img in {False,True}, ego motion in {identity, short (gap 3), mid (gap 9, predicted odom)}
  g1_ego.npz ego chain: reference data_utils matrices the build's ego.py must reproduce
  g2_glue.npz  export/load hop (depth u16 quantisation, decode, clamp) + one-hot
  g3_*.npz   BGModel.predict on synthetic weights (seed 1234 + calibration) at 64x128 and 96x160
  g4_arch.json  per-conv (name, cin, cout, k, stride, out H, W) table at 1024x2048 from the reference modules
  calib_seed1234.json  finalConv calibration constants (reference forward, 256x512, input seed 1)
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import  # noqa: E402
from panoptic_forecasting_amd import ego, synth  # noqa: E402

PCTransformModel, BGModel, data_utils = _ref_import.install()
torch.set_grad_enabled(False)


def bits(x):
    return x.detach().cpu().numpy().view(np.uint32) if x.dtype == torch.float32 else x.numpy()


def gen_g1():
    for (h, w) in [(48, 96), (64, 128)]:
        for motion in ['identity', 'short', 'mid']:
            inp = synth.make_inputs(b=2, t=3, h=h, w=w, seed=h + len(motion),
                                    gap_len=9 if motion == 'mid' else 3, identity=(motion == 'identity'),
                                    depth_mode='uniform' if motion != 'short' else 'scene',
                                    predicted=(motion == 'mid'))
            # make the two samples differ in camera / motion a little (per-sample matrices are exercised)
            inp['intrinsics'][1, 0, 2] += 3.25
            if motion != 'identity':
                inp['target_T'][1] = torch.from_numpy(
                    synth.target_T(gap_len=9 if motion == 'mid' else 3, speed=7.0, yaw_rate=-0.05)).float()
            out = {'K': inp['intrinsics'].numpy(), 'E': inp['extrinsics'].numpy(),
                   'T': inp['target_T'].numpy(), 'depth': inp['depth'].numpy(),
                   'mask': inp['depth_mask'].numpy(), 'seg': inp['seg'].numpy(),
                   'Kinv': torch.inverse(inp['intrinsics']).numpy(),
                   'Einv': torch.inverse(inp['extrinsics']).numpy()}
            g = torch.Generator().manual_seed(7)
            img = torch.randint(0, 256, (2, 3, h, w, 3), generator=g, dtype=torch.uint8)
            out['img'] = img.numpy()
            for ind in [None, 0, 1, 2]:
                for is_img in [False, True]:
                    model = PCTransformModel({'model': {'only_this_ind': ind, 'is_img': is_img}})
                    feed = {k: v.clone() for k, v in inp.items()}
                    if is_img:
                        feed['seg'] = img.clone()
                    res = model.predict(feed, None)
                    tag = '%s_%d' % ('all' if ind is None else str(ind), int(is_img))
                    out['seg_' + tag] = res['seg'].numpy()
                    out['depth_bits_' + tag] = res['depth'].numpy().view(np.uint32)
                    if not is_img:
                        r2d = res['result2d'].numpy()
                        assert r2d.min() >= 0 and r2d.max() < 32768
                        out['result2d_' + tag] = r2d.astype(np.int16)
            path = os.path.join(HERE, 'g1_%s_%dx%d.npz' % (motion, h, w))
            np.savez_compressed(path, **out)
            print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def gen_ego():
    """Reference data_utils outputs for the host-side chain (float64)."""
    cam = {'extrinsic': dict(synth.CS_EXTRINSIC), 'intrinsic': {}}
    E_ref = data_utils.cityscapes_camera2extrinsics(cam)
    K_ref = data_utils.build_intrinsics_mat([synth.CS_FX, synth.CS_FY, synth.CS_U0, synth.CS_V0])
    cases = [(10.0, 0.02, 1 / 17.0), (7.0, -0.05, 0.06), (3.0, 0.0001, 0.0588), (0.0, 0.0, 0.05),
             (12.5, 0.3, 0.059)]
    steps = np.stack([data_utils.get_vehicle_now_T_prev(*c)[0] for c in cases])
    # cumulative chain exactly as pc_transform_dataset.py:219-231 for target 19, inputs [10,13,16]
    ego_T = [data_utils.get_vehicle_now_T_prev(10.0 + 0.1 * k, 0.02 - 0.001 * k, 1 / 17.0 + 1e-4 * k)[0]
             for k in range(1, 30)]
    cum = []
    cur = np.eye(4)
    cum.append(cur)
    for fr in range(19 - 1, -1, -1):
        cur = cur @ ego_T[fr]
        cum.append(cur)
    cum.reverse()
    cum = np.stack(cum)[np.array([10, 13, 16])]
    np.savez_compressed(os.path.join(HERE, 'g1_ego.npz'), E=E_ref, K=K_ref,
                        step_cases=np.array(cases), steps=steps, chain=cum)
    assert np.array_equal(ego.camera_extrinsics(synth.CS_EXTRINSIC), E_ref)
    print('wrote g1_ego.npz')


def gen_g2():
    """Export hop (export_cityscapes_segmentation_results.py:119-124) and load hop (bg_dataset.py:224-230)."""
    g = torch.Generator().manual_seed(3)
    d = torch.rand(16, 32, generator=g) * 300.0 - 20.0     # includes <0, >254 (clamp) and holes
    d[0, :8] = -1.0
    q = ((d + 1).clamp(0, 255) * 256).round().numpy().astype(np.uint16)   # export side
    x = torch.from_numpy(q.astype(np.float32)).float()
    dec = x / 256.0 - 1                                                    # load side
    m = dec > 0
    dec[~m] = -1
    dec[m & (dec > 200)] = 200
    dec[m & (dec < 0.1)] = 0.1
    seg = torch.tensor([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 18, 255] * 37)[:16 * 32].view(1, 1, 16, 32)
    model = BGModel({'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])]},
                     'model': {'num_inputs': 1, 'use_depth_inps': True, 'convert2onehot': True}})
    onehot = model._inp2onehot(seg.clone())
    np.savez_compressed(os.path.join(HERE, 'g2_glue.npz'), depth_in=d.numpy(), depth_u16=q,
                        depth_dec=dec.numpy(), mask_dec=m.numpy(), seg=seg.numpy(),
                        onehot=onehot.numpy().astype(np.uint8))
    print('wrote g2_glue.npz')


def ref_bg_model(sd, h, w, final=True):
    params = {'data': {'num_classes': 11,
                       'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])]},
              'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True}}
    if final:
        params['model'].update(final_w=w, final_h=h)
    m = BGModel(params)
    missing = m.load_state_dict(sd, strict=True)
    m.eval()
    return m


def gen_calib():
    sd = synth.make_state_dict(seed=1234)
    m = ref_bg_model(sd, 256, 512)
    inp = synth.make_bg_inputs(b=1, h=256, w=512, seed=1)
    res = m.predict({k: v.clone() for k, v in inp.items()}, None)
    lo = res['orig_size_logits'].double()
    calib = {'mean': lo.mean(dim=(0, 2, 3)).tolist(), 'std': lo.std(dim=(0, 2, 3)).tolist(),
             'note': 'class-logit mean/std of the uncalibrated seed-1234 net on make_bg_inputs(h=256,w=512,seed=1), '
                     'reference BGModel forward, torch %s CPU' % torch.__version__}
    with open(os.path.join(HERE, 'calib_seed1234.json'), 'w') as f:
        json.dump(calib, f, indent=1)
    print('wrote calib_seed1234.json', calib['mean'][:3], calib['std'][:3])
    return calib


def gen_g3(calib):
    sd = synth.make_state_dict(seed=1234, calib=calib)
    for (h, w) in [(64, 128), (96, 160)]:
        m = ref_bg_model(sd, h, w)
        inp = synth.make_bg_inputs(b=2, h=h, w=w, seed=5)
        # per-stage checksums for bisecting: hook every leaf conv-layer output
        stages = {}
        def hook(name):
            def f(mod, i, o):
                stages[name] = np.array([float(o.double().sum()), float(o.double().abs().sum())])
            return f
        hs = []
        for name, mod in m.model.named_modules():
            if hasattr(mod, 'conv') and hasattr(mod, 'norm'):
                hs.append(mod.register_forward_hook(hook(name)))
        res = m.predict({k: v.clone() for k, v in inp.items()}, None)
        for hk in hs:
            hk.remove()
        hist = torch.bincount(res['seg'].flatten(), minlength=11).numpy()
        print('g3 %dx%d class histogram' % (h, w), hist)
        np.savez_compressed(os.path.join(HERE, 'g3_%dx%d.npz' % (h, w)),
                            seg_in=inp['seg'].numpy().astype(np.uint8), depth=inp['depth'].numpy(),
                            mask=inp['depth_mask'].numpy(),
                            orig_size_logits=res['orig_size_logits'].numpy(),
                            logits=res['logits'].numpy().astype(np.float32),
                            seg=res['seg'].numpy().astype(np.uint8),
                            stage_names=np.array(list(stages.keys())),
                            stage_sums=np.stack(list(stages.values())))
        print('wrote g3_%dx%d.npz' % (h, w))


def gen_g4():
    sd = synth.make_state_dict(seed=1234)
    m = ref_bg_model(sd, 1024, 2048)
    rows = []
    def hook(name):
        def f(mod, i, o):
            wt = mod.weight
            rows.append({'name': name, 'cin': wt.shape[1], 'cout': wt.shape[0], 'k': wt.shape[2],
                         'stride': mod.stride[0], 'oh': o.shape[2], 'ow': o.shape[3]})
        return f
    hs = [mod.register_forward_hook(hook(name)) for name, mod in m.model.named_modules()
          if isinstance(mod, torch.nn.Conv2d)]
    x = torch.zeros(1, 36, 128, 256)
    m.model(x)
    for r in rows:   # scale the probe's spatial dims to 1024x2048 (all dims divide exactly)
        r['oh'] *= 8
        r['ow'] *= 8
    keys = sorted(m.state_dict().keys())
    with open(os.path.join(HERE, 'g4_arch.json'), 'w') as f:
        json.dump({'convs': rows, 'state_dict_keys': keys,
                   'shapes': {k: list(v.shape) for k, v in m.state_dict().items()}}, f)
    print('wrote g4_arch.json', len(rows), 'convs', len(keys), 'keys')


if __name__ == '__main__':
    gen_ego()
    gen_g1()
    gen_g2()
    calib = gen_calib()
    gen_g3(calib)
    gen_g4()
