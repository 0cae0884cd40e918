"""Synthetic Cityscapes-shaped inputs for tests and ``bench.py`` (no dataset on the box).

Shapes, camera constants and the ego-motion recipe follow SURVEY.md §8(d):
Cityscapes intrinsics, extrinsics with pitch 0.038 rad and t=(1.7, 0.1, 1.22),
unicycle motion at 10 m/s / 0.02 rad/s / dt=1/17 s; input frames [10,13,16]
(short term, gap 3) or [4,7,10] (mid term, gap 9) forecasting frame 19 — the
index arithmetic of reference ``pc_transform_dataset.py:83,94``.
"""
import numpy as np
import torch

from . import ego

CS_FX, CS_FY, CS_U0, CS_V0 = 2262.52, 2265.3017905988554, 1096.98, 513.137
CS_EXTRINSIC = dict(yaw=0.0, pitch=0.038, roll=0.0, x=1.7, y=0.1, z=1.22)
CS_H, CS_W = 1024, 2048

# Cityscapes label table (public dataset constants; cityscapesscripts.helpers.labels):
# id -> trainId.  Used by the export hop (reference
# experiments/export_cityscapes_segmentation_results.py:34-38) and by BGDataset.
ID2TRAINID = np.full(256, 0, dtype=np.uint8)       # ids outside the table map to 0 (zeros_like init)
ID2TRAINID[:34] = 255
for _id, _tid in {7: 0, 8: 1, 11: 2, 12: 3, 13: 4, 17: 5, 19: 6, 20: 7, 21: 8, 22: 9, 23: 10,
                  24: 11, 25: 12, 26: 13, 27: 14, 28: 15, 31: 16, 32: 17, 33: 18}.items():
    ID2TRAINID[_id] = _tid
TRAINID2ID = np.array([7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33],
                      dtype=np.uint8)


def camera(h=CS_H, w=CS_W):
    """Intrinsics scaled with the image size so small test images see the same field of view."""
    sx, sy = w / CS_W, h / CS_H
    K = ego.intrinsics_matrix(CS_FX * sx, CS_FY * sy, CS_U0 * sx, CS_V0 * sy)
    E = ego.camera_extrinsics(CS_EXTRINSIC)
    return K, E


def target_T(gap_len=3, speed=10.0, yaw_rate=0.02, dt=1.0 / 17.0, predicted=False, rng=None):
    """[3,4,4] float64 transforms taking each input frame's vehicle pose to frame 19."""
    target = 19
    input_inds = np.array([0, 3, 6]) + target - (6 + gap_len)
    n = 30
    times = np.arange(n) * dt
    speeds = np.full(n, speed)
    yaws = np.full(n, yaw_rate)
    if rng is not None:  # mild per-frame variation
        speeds = speeds + rng.normal(0, 0.3, n)
        yaws = yaws + rng.normal(0, 0.002, n)
    if not predicted:
        return ego.target_T_measured(speeds, yaws, times, input_inds, target)
    preds = np.stack([np.full(18, speed * 0.98), np.full(18, yaw_rate * 1.05)], axis=1)
    return ego.target_T_predicted(speeds, yaws, times, preds, input_inds, target, gap_len)


def _scene_depth(g, t, h, w, K):
    """Plane-ish depth: ground plane below the horizon, blocky facades above, 1 % jitter."""
    v = torch.arange(h, dtype=torch.float32).view(1, h, 1)
    fy, v0 = float(K[1, 1]), float(K[1, 2])
    ground = (fy * 1.22) / (v - v0).clamp(min=1e-3)
    bh, bw = max(h // 16, 1), max(w // 32, 1)
    blocks = torch.rand(t, bh, bw, generator=g) * 72.0 + 10.0
    facade = torch.nn.functional.interpolate(blocks[None], size=(h, w), mode='nearest')[0]
    d = torch.minimum(ground.expand(t, h, w), facade)
    d = d * (1.0 + 0.01 * (torch.rand(t, h, w, generator=g) - 0.5))
    return d.clamp(2.0, 82.0)


def make_inputs(b=1, t=3, h=CS_H, w=CS_W, seed=0, gap_len=3, identity=False, depth_mode='scene',
                mask_p=0.9, trainids=False, predicted=False, device='cpu'):
    """Batch dict ``inputs`` for PCTransformModel.predict (reference pc_transform_dataset.py:320-334).

    seg is u8 label ids (0..33) unless ``trainids`` (then LUT-mapped), depth f32, depth_mask bool.
    """
    g = torch.Generator().manual_seed(1000 + seed)
    K, E = camera(h, w)
    if identity:
        T = np.stack([np.eye(4)] * t)
    else:
        T = target_T(gap_len=gap_len, predicted=predicted)[:t]
    segs, depths = [], []
    for _ in range(b):
        lab = torch.randint(0, 34, (t, max(h // 16, 1), max(w // 16, 1)), generator=g, dtype=torch.uint8)
        seg = torch.nn.functional.interpolate(lab[None].float(), size=(h, w), mode='nearest')[0].to(torch.uint8)
        flip = torch.rand(t, h, w, generator=g) < 0.02
        noise = torch.randint(0, 34, (t, h, w), generator=g, dtype=torch.uint8)
        seg = torch.where(flip, noise, seg)
        if depth_mode == 'uniform':
            d = torch.rand(t, h, w, generator=g) * 80.0 + 2.0
        else:
            d = _scene_depth(g, t, h, w, K)
        segs.append(seg)
        depths.append(d)
    seg = torch.stack(segs)
    if trainids:
        seg = torch.from_numpy(ID2TRAINID)[seg.long()]
    depth = torch.stack(depths).float()
    mask = torch.rand(b, t, h, w, generator=g) < mask_p
    out = {
        'intrinsics': torch.from_numpy(K).float().unsqueeze(0).repeat(b, 1, 1),
        'extrinsics': torch.from_numpy(E).float().unsqueeze(0).repeat(b, 1, 1),
        'target_T': torch.from_numpy(T).float().unsqueeze(0).repeat(b, 1, 1, 1),
        'depth': depth,
        'depth_mask': mask,
        'seg': seg,
    }
    return {k: v.to(device) for k, v in out.items()}


def make_bg_inputs(b=1, t=3, h=CS_H, w=CS_W, seed=0, device='cpu', num_classes=11):
    """Batch dict ``inputs`` for BGModel.predict (reference bg_dataset.py:203-230): blocky label maps.

    seg int64 with trainIds 0..18 and 255 (things/void -> all-zero one-hot), depth f32 in
    [0.1,200] or -1 where masked, depth_mask bool.
    """
    g = torch.Generator().manual_seed(2000 + seed)
    lab = torch.randint(0, 20, (b, t, max(h // 16, 1), max(w // 16, 1)), generator=g)
    lab[lab == 19] = 255
    seg = torch.nn.functional.interpolate(lab.float(), size=(h, w), mode='nearest').long()
    K, _ = camera(h, w)
    d = torch.stack([_scene_depth(g, t, h, w, K) for _ in range(b)])
    mask = torch.rand(b, t, h, w, generator=g) < 0.9
    d = torch.where(mask, d, torch.full_like(d, -1.0))
    out = {'seg': seg, 'depth': d.float(), 'depth_mask': mask}
    return {k: v.to(device) for k, v in out.items()}


# --------------------------------------------------------------------------
# Synthetic ("random-init") weights.  The reference's default init collapses the
# argmax to one class, so tests and the bench use this recipe:
# He-normal convs, randomised BN statistics, and a finalConv rescale that makes
# every class logit zero-mean / unit-std on a calibration input.  numpy's PCG64
# streams are bit-reproducible across machines (the GPU box regenerates exactly
# the weights the golden fixtures were made with).
# --------------------------------------------------------------------------
def make_state_dict(seed=1234, in_ch=36, n_cls=11, calib=None, depth_norm=(20.0, 15.0)):
    """state_dict with the reference checkpoint's key set (418 tensors for the bg net).

    ``calib``: optional {'mean': [n_cls], 'std': [n_cls]} of the un-calibrated class logits;
    applied as w_c /= std_c, b_c = -mean_c/std_c.
    """
    from . import hardnet_arch as arch
    rng = np.random.Generator(np.random.PCG64(seed))
    spec = arch.Spec(in_ch, n_cls)
    sd = {}
    sd['depth_mean'] = torch.tensor([depth_norm[0]], dtype=torch.float32)
    sd['depth_std'] = torch.tensor([depth_norm[1]], dtype=torch.float32)
    for op in spec.conv_ops():
        fan_in = op.cin * op.k * op.k
        w = rng.standard_normal((op.cout, op.cin, op.k, op.k)) * np.sqrt(2.0 / fan_in)
        p = 'model.' + op.name
        if op.bn:
            sd[p + '.conv.weight'] = torch.from_numpy(w.astype(np.float32))
            sd[p + '.norm.weight'] = torch.from_numpy(rng.uniform(0.5, 1.5, op.cout).astype(np.float32))
            sd[p + '.norm.bias'] = torch.from_numpy((rng.standard_normal(op.cout) * 0.1).astype(np.float32))
            sd[p + '.norm.running_mean'] = torch.from_numpy((rng.standard_normal(op.cout) * 0.1).astype(np.float32))
            sd[p + '.norm.running_var'] = torch.from_numpy(rng.uniform(0.5, 1.5, op.cout).astype(np.float32))
            sd[p + '.norm.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)
        else:
            w = w.astype(np.float32)
            b = np.zeros(op.cout, dtype=np.float32)
            if calib is not None:
                std = np.asarray(calib['std'], dtype=np.float64)
                mean = np.asarray(calib['mean'], dtype=np.float64)
                w = (w / std.reshape(-1, 1, 1, 1)).astype(np.float32)
                b = (-mean / std).astype(np.float32)
            sd[p + '.weight'] = torch.from_numpy(w)
            sd[p + '.bias'] = torch.from_numpy(b)
    return sd


def make_pretrained_checkpoint(seed=77):
    """A synthetic FC-HarDNet "pretrain_path" pickle payload: the 19-class, 3-channel-stem network the reference's
    ``build_hardnet`` loads (hardnet.py:390-400: ``torch.load(path)['model_state']`` with ``module.``-prefixed keys)."""
    sd = make_state_dict(seed=seed, in_ch=3, n_cls=19)
    out = {}
    for k, v in sd.items():
        if k.startswith('model.'):
            out['module.' + k[len('model.'):]] = v
    return {'model_state': out}
