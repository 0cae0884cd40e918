"""Host-side camera / ego-motion math feeding the warp (float64 numpy, tiny).

Mirrors what the reference computes before the hot path is entered:

* camera JSON -> intrinsics / extrinsics
  (reference ``data/data_utils.py:74-78,109-114,170-214``),
* unicycle ego step ``now_T_prev`` (``data/data_utils.py:117-165``),
* the cumulative ``target_T`` chain for the three input frames
  (``data/datasets/pc_transform_dataset.py:126-186`` predicted-odometry branch and
  ``:219-231`` measured-odometry branch).

All matrices are float64 here and cast to float32 exactly once when they are
handed to the device path, as the reference does with ``.float()``.
"""
import numpy as np

ANGLE_RAD_EPS = 0.000175  # data_utils.py:139 (~0.01 deg): below this the step is a straight line


def affine(R=None, t=None):
    T = np.identity(4, dtype=np.float64)
    if R is not None:
        T[:3, :3] = R
    if t is not None:
        T[:3, 3] = t
    return T


def flu_T_rdf():
    """RDF camera axes (x right, y down, z front) -> FLU axes (x front, y left, z up)."""
    return affine(R=np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float64))


def vehicle_T_camera_flu(extrinsic):
    """``extrinsic``: dict with yaw/pitch/roll/x/y/z (Cityscapes camera.json)."""
    sy, cy = np.sin(extrinsic['yaw']), np.cos(extrinsic['yaw'])
    sp, cp = np.sin(extrinsic['pitch']), np.cos(extrinsic['pitch'])
    sr, cr = np.sin(extrinsic['roll']), np.cos(extrinsic['roll'])
    R = np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                  [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                  [-sp, cp * sr, cp * cr]], dtype=np.float64)
    t = np.array([extrinsic['x'], extrinsic['y'], extrinsic['z']], dtype=np.float64)
    return affine(R, t)


def camera_extrinsics(extrinsic):
    """vehicle(FLU)_T_camera(RDF) — the ``extrinsics`` input of the warp."""
    return vehicle_T_camera_flu(extrinsic) @ flu_T_rdf()


def intrinsics_matrix(fx, fy, u0, v0):
    K = np.eye(3)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, u0, v0
    return K


def now_T_prev(speed, yaw_rate, dt):
    """Unicycle step: pose of the previous vehicle frame expressed in the current one."""
    if abs(yaw_rate) < ANGLE_RAD_EPS:
        x, y, theta = dt * speed, 0.0, 0.0
    else:
        r = speed / yaw_rate
        wt = yaw_rate * dt
        x, y, theta = r * np.sin(wt), r - r * np.cos(wt), wt
    c, s = np.cos(theta), np.sin(theta)
    prev_T_now = affine(np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64),
                        np.array([x, y, 0.0]))
    return np.linalg.inv(prev_T_now)


def cumulative_target_T(steps):
    """``steps[k]`` = now_T_prev taking frame k to frame k+1 (k = 0 .. n-1).

    Returns ``cum[k]`` (k = 0 .. n) = transform from frame k to the last frame n:
    ``cum[n] = I``, ``cum[k] = steps[n-1] @ ... @ steps[k]`` — the product order
    of ``pc_transform_dataset.py:172-178,221-226``.
    """
    cur = np.eye(4)
    out = [cur]
    for k in range(len(steps) - 1, -1, -1):
        cur = cur @ steps[k]
        out.append(cur)
    out.reverse()
    return np.stack(out)


def target_T_measured(speeds, yaw_rates, times, input_inds, target):
    """Measured-odometry branch (``pc_transform_dataset.py:103-125,219-231``).

    ``speeds/yaw_rates/times`` are per-frame lists for the 30-frame snippet;
    the step into frame k uses the odometry *of frame k* and ``times[k]-times[k-1]``.
    """
    steps = [now_T_prev(speeds[k], yaw_rates[k], times[k] - times[k - 1])
             for k in range(1, len(times))]
    cum = cumulative_target_T(steps[:target])
    return cum[np.asarray(input_inds)]


def target_T_predicted(speeds, yaw_rates, times, odom_preds, input_inds, target, gap_len):
    """Predicted-odometry branch (``pc_transform_dataset.py:156-186``).

    Past steps (between the first and last *input* frame) use measured
    odometry; the ``gap_len`` future steps use rows of the odometry forecaster's
    output ``odom_preds[:gap_len] = [speed, yaw_rate]`` with ``dt = mean(past dt)``.
    The three matrices returned are entries [0, 3, 6] of the cumulative chain.
    """
    input_inds = np.asarray(input_inds)
    first, start = int(input_inds[0]), int(input_inds[-1])
    past_times = np.asarray(times[first:start + 1], dtype=np.float64)
    sp = list(speeds[first + 1:start + 1]) + list(np.asarray(odom_preds)[:gap_len, 0])
    yr = list(yaw_rates[first + 1:start + 1]) + list(np.asarray(odom_preds)[:gap_len, 1])
    dts = past_times[1:] - past_times[:-1]
    dts = list(dts) + [np.mean(dts)] * gap_len
    steps = [now_T_prev(sp[k], yr[k], dts[k]) for k in range(len(dts))]
    cum = cumulative_target_T(steps)
    if len(cum) != target - first + 1:
        raise ValueError('ego chain length %d != expected %d' % (len(cum), target - first + 1))
    return cum[np.array([0, 3, 6])]
