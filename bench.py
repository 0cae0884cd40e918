#!/usr/bin/env python
"""Headline benchmark: bg forecast frames/sec @1024x2048, 3 inputs -> Δt=3 (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch of synthetic, HBM-resident inputs on every rank:
    3 x (unproject + ego warp + z-buffered splat)  ->  on-the-fly disk-hop emulation  ->  FC-HarDNet-70
    -> bilinear upsample + argmax   (task ``bg_forecast`` of panoptic-forecasting_amd)
Ranks are independent (sequences shard by batch); the only collective is the end-of-run all-gather of
the PQ accumulators.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--replays R] [--no-graph] [--no-cpu-baseline] [--no-legs]

A step visits the rank's resident synthetic batch R times (``--replays``, default 2: 2 x 128 = 256 forecast frames per GPU per
step, one hipGraph), so that the driver's 20 steps time about two seconds instead of a quarter of one; ``config`` states it.
The 128 resident frames run as 4 sub-batches of 32 on 4 HIP streams (round 5: +1.9 % over 64 / 4 in a same-box sweep, 192 and 256
frames give no more: profiles/r05_experiments.md), staggered (``--stagger 1``): the warp/splat of sub-batch
i + 1 starts when that of sub-batch i is done, so the vector-ALU-bound front of one sub-batch runs beside the memory-bound
convolutions of another (+3 % over starting them together: profiles/r03_experiments.md).

``--gpus N`` with N > 1 and no launcher environment (WORLD_SIZE unset) re-executes this file under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`` — one rank per GPU over
RCCL; started by an external launcher it reads RANK/LOCAL_RANK/WORLD_SIZE from the environment.  It refuses to run
when the ranks that joined differ from ``--gpus`` or the node has fewer GPUs than ranks.

At N=1 the same line also carries (driver-timed, same process):
    by_batch   frames/s of ONE stream at B = 1, 2, 4 frames per step (SURVEY.md 8d Config 2; the reference loop batches 2): latency
    by_batch_pipelined  the reference-shaped loop with batches in flight (export_bg.py --pipeline_depth N): B = 1, 2 per batch, N = 2
               and 3 model replicas on as many streams, eager, fresh host-inverse camera tensors every batch
    fresh_cameras  eager B = 2 / 16 with NEW camera tensors every step: inverses taken on the host batch vs read back by the model
    other_resolution  512x1024 at B = 16, one stream: the heuristic (untuned) kernel choice, frames/s + dominant-kernel fraction
    train_step  one bg training step at batch 8 of 800x800 (configs/bg/bg_train.yaml): ms, dominant kernel + its fraction of the fp32 matrix peak
    fp32_only  the headline workload with every convolution on the fp32 MFMA (no two-term fp16 operands) + its parity
    roofline   dominant kernel (live hipEvent timing) + ``step``: whole-step algorithmic bytes / kernel time, per stage
    cpu_baseline  the oracle pipeline on the host cores: at the best torch thread count of a short sweep (`value`) and on all of
                  them (`value_all_cores`)
    range_overflow  false = no forward inside a captured graph of the timed region met an activation outside the range of the
                  fp16-pair path (PF_STATUS_RANGE / PF_STATUS_RANGE_LOW, include/pfhip.h; read from the workspaces' sticky status
                  words after the timed region); a true here would invalidate the run and bench.py exits non-zero.  Eager
                  forwards (--no-graph) that were flagged have been re-run on fp32 MFMA (`range_reruns`): valid, not an error;
                  `range_status_sticky` carries the raw word either way
The driver keeps the standard keys plus the scalars of `config`, `roofline` and `cpu_baseline` (strings cut at 120 characters, nested
objects dropped), so every number a reader needs is ALSO a scalar there: roofline.step_frac, roofline.fp32_only_value / _frac /
_kernel / _max_abs_dlogit, roofline.train_step_ms, roofline.other_resolution_value / _frac, roofline.parity_max_abs_dlogit,
config.by_batch, config.by_batch_pipelined.  The line is kept under 6 KB (--verbose adds the per-stage breakdown and the legs' details).
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from panoptic_forecasting_amd import dist as pfdist  # noqa: E402
from panoptic_forecasting_amd import synth  # noqa: E402

H, W, T = 1024, 2048, 3
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, dense
PEAK_HBM_GBPS = 8000.0          # HBM3E spec (6.3 TB/s achievable)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 (2495 measured; tools/ubench/f16_split.hip: same rate)
CPU_BUDGET_S = 20.0             # stop starting new baseline frames after this much CPU time
SUB_BATCH = 32                  # frames per concurrent sub-batch of the headline workload (the shape tables hold rows measured at B = 32)
# test hook (tests/test_gpu_pipeline.py): PF_BENCH_SHARE_GPU=1 lets the ranks of --gpus N share the GPUs that exist (gloo instead of
# RCCL, which refuses two ranks per device) so that the N > 1 line - per-rank times, gather, backend - is exercised on a 1-GPU box.
# Never a measurement: the line says so (config.backend = gloo, identical `devices`)
SHARE_GPU = os.environ.get('PF_BENCH_SHARE_GPU') == '1'


def build_model(params):
    """The registry prints like the reference's does (models/__init__.py:18); stdout carries only the JSON line here."""
    import contextlib
    from panoptic_forecasting_amd.registry import build_model as _build_model
    with contextlib.redirect_stdout(sys.stderr):
        return _build_model(params)


def calibrated_state_dict():
    with open(os.path.join(ROOT, 'tests', 'golden', 'calib_seed1234.json')) as f:
        calib = json.load(f)
    return synth.make_state_dict(seed=1234, calib=calib)


def model_params(**model_kw):
    p = {'task': 'bg_forecast', 'no_gpu': False, 'load_model': None, 'load_best_model': False,
         'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])],
                  'min_depth': 0.1, 'max_depth': 200},
         'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True,
                   'final_h': H, 'final_w': W, 'emulate_disk_hop': True, 'seg_is_label_id': True,
                   # every frame gets the sentinel the reference gives it at batch size 1 (outputs independent of how the
                   # frames are batched and sharded; the reference's batch-global max couples the samples of a call)
                   'per_sample_sentinel': True}}
    # on_range_overflow keeps its default ('rerun'): predict() enqueues and never waits; eager runs check every forward's
    # status words lazily (pinned copy + event), captured graphs OR them into the workspace's sticky word, which is read
    # once after the timed region (range_overflow in the JSON line)
    p['model'].update(model_kw)
    return p


TERM = {'short': dict(gap_len=3, predicted=False), 'mid': dict(gap_len=9, predicted=True)}   # configs[1] / configs[2]


def make_batch(b, seed0, device, term='short'):
    parts = [synth.make_inputs(b=1, t=T, h=H, w=W, seed=seed0 + i, **TERM[term]) for i in range(b)]
    inp = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}
    # the batch dict is exactly the reference's (no pre-computed camera inverses): the model inverts K and E on the host
    # (LAPACK, like the reference) once per distinct camera tensor and afterwards hits its cache without touching the
    # stream (pc_transform_model.InverseCache) - the warm-up steps pay it, the timed region and the captured graph do not
    return {k: v.to(device) for k, v in inp.items()}


def source_sha():
    """Hash of the device sources the loaded library was built from: profiles/pmc_latest.json records it, and a PMC
    file measured on other kernels is refused (roofline.traffic = null) instead of being quoted."""
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, 'panoptic-forecasting_amd', 'csrc', '*'))):
        if p.endswith(('.hip', '.h', '.cpp', '.inc', 'Makefile')):
            h.update(os.path.basename(p).encode())
            with open(p, 'rb') as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_label):
    """(HBM bytes per launch of `kernel_label`, note) from the committed PMC passes (profiles/pmc_latest.json: FETCH_SIZE
    + WRITE_SIZE collected in separate rocprofv3 --pmc runs of this same command and calibrated on a known-size copy,
    tools/profile_gpu.sh + tools/profile_summarise.py)."""
    path = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
    if not os.path.exists(path):
        return None, 'profiles/pmc_latest.json missing'
    with open(path) as f:
        pmc = json.load(f)
    if pmc.get('source_sha') != source_sha():
        return None, 'profiles/pmc_latest.json was measured on other kernel sources (sha %s, now %s): refused' % (
            pmc.get('source_sha'), source_sha())
    # labels are the rocprofv3 symbol + optional " +res"/" +pool"/" lowres-half" annotations of the launch
    k = pmc.get('kernels', {}).get(kernel_label.split(')')[0] + ')')
    if not k or k.get('hbm_bytes_per_launch') is None:
        return None, 'kernel not in profiles/pmc_latest.json'
    return k['hbm_bytes_per_launch'], 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, calibrated), profiles/pmc_latest.json'


def cpu_model_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _oracle_splats(seed, term):
    """The three per-frame warp/splats of one forecast frame + the hop quantisation, on the C oracle (3 host threads:
    the C call releases the GIL)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import warp_splat as ow
    inp = synth.make_inputs(b=1, t=T, h=H, w=W, seed=seed, **TERM[term])

    def one(t):
        o = ow.predict(inp, only_this_ind=t)
        tid = torch.from_numpy(synth.ID2TRAINID)[o['seg'].long()]
        q = ((o['depth'] + 1).clamp(0, 255) * 256).round().numpy().astype(np.uint16)
        d = torch.from_numpy(q.astype(np.float32)) / 256.0 - 1
        m = d > 0
        d[~m] = -1
        d[m & (d > 200)] = 200
        d[m & (d < 0.1)] = 0.1
        return tid, d
    with ThreadPoolExecutor(T) as ex:
        res = list(ex.map(one, range(T)))
    return torch.stack([r[0] for r in res], 1).long(), torch.stack([r[1] for r in res], 1)


def _cpu_sample(sd, n_frames, term, budget_s, seed0=0):
    """frames/s of the oracle pipeline at the CURRENT torch thread count: the three splats of a frame run on three
    threads (scalar C, the scatter itself is sequential like pytorch_scatter's CPU loop) and are prefetched one frame ahead
    while torch-CPU runs the network of the current frame."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import hardnet_ref
    last = None
    pre = ThreadPoolExecutor(1)
    t0 = time.perf_counter()
    nxt = pre.submit(_oracle_splats, seed0, term)
    done = 0
    for f in range(n_frames):
        seg, dep = nxt.result()
        more = f + 1 < n_frames and time.perf_counter() - t0 <= budget_s
        if more:
            nxt = pre.submit(_oracle_splats, seed0 + f + 1, term)
        last = hardnet_ref.bg_predict(sd, {'seg': seg, 'depth': dep, 'depth_mask': dep > 0}, final_size=(H, W))
        done = f + 1
        if not more:
            break
    dt = time.perf_counter() - t0
    pre.shutdown()
    # the generation of synthetic inputs is inside the loop but is <3 % of it
    return done / dt, dt, last, done


def cpu_baseline(sd, n_frames, term='short'):
    """The oracle (CPU port of the reference path) timed on this box's host cores.  torch's intra-op pool is swept over
    {32, 64, 128, os.cpu_count()} threads on one warm network forward each (all 256 hardware threads of a 2-socket box
    thrash: round 2 measured 2.1 s per frame there against 1.25 s on 8 cores); the sample is timed at the best setting
    (`value`) and, when that is not all cores, again on all of them (`value_all_cores`, what BASELINE.md asks for).
    The last frame of the sample (seed n_done - 1) is the parity reference of the GPU path."""
    from oracle import hardnet_ref
    from oracle import warp_splat as ow
    ncpu = os.cpu_count() or 1
    ow.lib()
    seg, dep = _oracle_splats(0, term)
    x = {'seg': seg, 'depth': dep, 'depth_mask': dep > 0}
    sweep = {}
    # (the all-cores setting is timed by the second sample below, not here: the first forward after growing torch's pool to
    #  256 threads took 72 s on the 2 x EPYC 9575F box)
    for n in sorted({min(n, ncpu) for n in ((32, 64, 128, ncpu) if ncpu <= 128 else (32, 64, 128))}):
        torch.set_num_threads(n)
        if not sweep:
            hardnet_ref.bg_predict(sd, x, final_size=(H, W))       # warm-up (allocator, oneDNN primitives)
        t0 = time.perf_counter()
        hardnet_ref.bg_predict(sd, x, final_size=(H, W))
        sweep[n] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    fps, secs, last, done = _cpu_sample(sd, n_frames, term, CPU_BUDGET_S * (1.0 if best == ncpu else 0.6))
    res = {'value': fps, 'unit': 'frames/s', 'cores': min(ncpu, best + T), 'kind': 'port', 'cpu_model': cpu_model_name(),
           'threads_best': best, 'host_threads': ncpu, 'value_all_cores': fps,
           'network_s_per_frame_by_threads': {str(k): round(v, 3) for k, v in sweep.items()},
           'sample': '%d forecast frames @%dx%d in %.1f s: per frame 3 C-oracle splats on 3 threads (prefetched one frame '
                     'ahead) + torch-CPU HarDNet with torch.set_num_threads(%d) = the fastest of a {32, 64, 128, %d}-thread '
                     'sweep on one warm frame' % (done, H, W, secs, best, ncpu)}
    if best != ncpu:
        torch.set_num_threads(ncpu)
        hardnet_ref.bg_predict(sd, x, final_size=(H, W))           # warm-up at this pool size, untimed
        fps_all, secs_all, _, done_all = _cpu_sample(sd, n_frames, term, CPU_BUDGET_S * 0.4, seed0=100)
        res['value_all_cores'] = fps_all
        res['sample'] += '; value_all_cores: %d frames in %.1f s with torch.set_num_threads(%d = os.cpu_count())' % (
            done_all, secs_all, ncpu)
        torch.set_num_threads(best)
    return res, last, done


class Workload:
    """B forecast frames per step as S concurrent sub-batches (own model object, workspaces and HIP stream each) inside
    one captured hipGraph.  The low-resolution layers of one sub-batch (small grids, latency-bound) and its memory-bound
    splat/stem kernels overlap the matrix-bound high-resolution layers of another."""

    def __init__(self, sd, B, S, dev, seed0, term, use_graph=True, stagger=False, free_run_ms=None, passes=1, batch=None, **model_kw):
        self.stagger = stagger
        # passes: how many times step() visits the resident batch.  With `stagger` they form ONE software pipeline (the
        # warp/splat of sub-batch i + 1 starts when that of sub-batch i is done, across pass boundaries too), captured
        # in one hipGraph: the streams join once per step, not once per pass
        self.passes = max(1, passes)
        # free_run_ms (experiment, --free-run MS): one hipGraph PER sub-batch, each replayed on its own stream with no join
        # between steps; sub-batch i starts i * MS late (a spin kernel inside the timed region), so that the streams run
        # out of phase instead of in lockstep.  K steps still enqueue K forwards of every sub-batch; the clock stops when
        # the last stream has drained
        self.free_run_ms = free_run_ms if (free_run_ms is not None and use_graph and S > 1) else None
        if B % S:
            raise SystemExit('--batch must be a multiple of --streams')
        self.B, self.S, self.use_graph = B, S, use_graph
        self.batch = batch if batch is not None else make_batch(B, seed0=seed0, device=dev, term=term)
        self.models = []
        for _ in range(S):
            m = build_model(model_params(**model_kw))
            m.load_state_dict(sd)
            m.eval()
            self.models.append(m)
        sub = B // S
        self.subs = [{k: v[i * sub:(i + 1) * sub].contiguous() for k, v in self.batch.items()} for i in range(S)]
        self.side = [torch.cuda.Stream() for _ in range(S - 1)]
        self.out = self.step()          # builds the plans, sizes the workspaces
        torch.cuda.synchronize()
        self.run = self.step
        if self.free_run_ms is not None:
            self.graphs = []
            streams = [torch.cuda.current_stream()] + self.side
            for i in range(S):
                st = torch.cuda.Stream()
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    for _ in range(2):
                        self.models[i].predict(self.subs[i], None)
                torch.cuda.current_stream().wait_stream(st)
                torch.cuda.synchronize()
                self.models[i].bg.settle()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.out[i] = self.models[i].predict(self.subs[i], None)
                self.graphs.append(g)
            # spin-kernel cycles per millisecond
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.cuda._sleep(10000000)
            e1.record()
            torch.cuda.synchronize()
            self.cyc_per_ms = 10000000 / e0.elapsed_time(e1)
            self.streams = streams
            self.run = None
            return
        if use_graph:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    self.step()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            for m in self.models:
                m.bg.settle()       # the eager warm-up forwards are checked (and their tensors released) before the capture
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self.step()
            self.run = self.graph.replay

    def step(self):
        cur = torch.cuda.current_stream()
        outs = [None] * self.S
        if self.stagger and self.S > 1:
            # software pipeline: sub-batch i + 1 starts when the warp/splat of sub-batch i has finished, so that its
            # vector-ALU-bound warp/splat + stem run beside the memory-bound convolutions of sub-batch i instead of beside
            # another warp/splat (two identical chains started together stay in lockstep: tools/graph_timeline.py).  The
            # chain continues across the passes of a step: sub-batch 0 of pass p + 1 follows the splat of the last
            # sub-batch of pass p (and, on its own stream, its own network of pass p)
            streams = [cur] + self.side
            for st in self.side:
                st.wait_stream(cur)
            prev = None
            for _ in range(self.passes):
                for i in range(self.S):
                    st = streams[i]
                    if prev is not None:
                        st.wait_event(prev)
                    ev = torch.cuda.Event()
                    self.models[i].after_splat = (lambda e=ev, q=st: e.record(q))
                    with torch.cuda.stream(st):
                        outs[i] = self.models[i].predict(self.subs[i], None)
                    self.models[i].after_splat = None
                    prev = ev
            for i in range(1, self.S):
                cur.wait_stream(self.side[i - 1])
            return outs
        for _ in range(self.passes):
            for i in range(1, self.S):
                self.side[i - 1].wait_stream(cur)
                with torch.cuda.stream(self.side[i - 1]):
                    outs[i] = self.models[i].predict(self.subs[i], None)
            outs[0] = self.models[0].predict(self.subs[0], None)
            for i in range(1, self.S):
                cur.wait_stream(self.side[i - 1])
        return outs

    def timed_free_run(self, steps, warmup, dev, barrier):
        cur = torch.cuda.current_stream()

        def go(n, offset):
            for i in range(1, self.S):
                self.streams[i].wait_stream(cur)
                if offset:
                    with torch.cuda.stream(self.streams[i]):
                        torch.cuda._sleep(int(i * self.free_run_ms * self.cyc_per_ms))
            for _ in range(n):
                for i in range(self.S):
                    with torch.cuda.stream(self.streams[i]):
                        self.graphs[i].replay()
            for i in range(1, self.S):
                cur.wait_stream(self.streams[i])
        go(warmup, False)
        if barrier and pfdist.is_dist():
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        go(steps, True)
        torch.cuda.synchronize()
        self.local_s = time.perf_counter() - t0
        if barrier and pfdist.is_dist():
            torch.distributed.barrier()
        return pfdist.max_over_ranks(time.perf_counter() - t0, dev) if barrier else time.perf_counter() - t0

    def timed(self, steps, warmup, dev, barrier=False):
        """Seconds for `steps` steps (max over ranks when `barrier`), each step = self.passes passes over the resident batch
        (one graph replay).  self.local_s keeps this rank's own time between the two synchronisations."""
        if self.free_run_ms is not None:
            return self.timed_free_run(steps * self.passes, warmup * self.passes, dev, barrier)
        for _ in range(warmup):
            self.run()
        if barrier and pfdist.is_dist():
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.run()
        torch.cuda.synchronize()
        self.local_s = time.perf_counter() - t0
        if barrier and pfdist.is_dist():
            torch.distributed.barrier()
        return pfdist.max_over_ranks(time.perf_counter() - t0, dev) if barrier else time.perf_counter() - t0

    def range_overflow(self):
        """OR of the sticky status words of the workspaces (include/pfhip.h: every forward ORs its final status into the
        sticky word, only the host clears it): non-zero = some forward since the last call met values outside the range of the
        fp16-pair path (PF_STATUS_RANGE = 1: |x| > 65504; PF_STATUS_RANGE_LOW = 2: a tensor of tiny values).  Eager forwards
        that were flagged have been re-run on the fp32 MFMA by then (counted in range_reruns)."""
        st = 0
        for m in self.models:
            m.bg.settle()
            st |= m.bg.range_status_sticky(clear=True)
        return st

    def outputs(self):
        return {k: torch.cat([o[k] for o in self.out]) for k in self.out[0]}


def fresh_cameras_leg(sd, dev, term):
    """The reference-shaped loop makes NEW camera tensors every batch (loader -> batch2gpu -> predict,
    export_cityscapes_segmentation_results.py:75-85), so the model's per-tensor inverse cache - which the replayed headline
    hits - never does.  Eager predicts on one stream at B = 2 (the reference's batch size) and B = 16, the big tensors
    resident, the camera tensors rebuilt from host memory every step, three ways:
      cached          the same device tensors every step (what every other leg of this file times)
      host_inverses   K^-1 / E^-1 taken on the host batch before the move (pc_transform_model.add_camera_inverses, what
                      export_bg.py does): five small H2D copies per step, nothing read back, no stream synchronisation
      device_cameras  fresh device tensors without inverses (an unmodified caller): the model reads K and E back, inverts with
                      LAPACK and uploads - two stream synchronisations per predict"""
    from panoptic_forecasting_amd.pc_transform_model import add_camera_inverses
    cam_keys = ('intrinsics', 'extrinsics', 'target_T')
    res = {}
    for b, steps in ((2, 240), (16, 48)):
        m = build_model(model_params())
        m.load_state_dict(sd)
        m.eval()
        host = {k: torch.cat([synth.make_inputs(b=1, t=T, h=H, w=W, seed=i, **TERM[term])[k] for i in range(b)], 0)
                for k in ('depth', 'depth_mask', 'seg') + cam_keys}
        big = {k: host[k].to(dev) for k in ('depth', 'depth_mask', 'seg')}
        cams_host = {k: host[k].pin_memory() for k in cam_keys}
        cams_dev = {k: v.to(dev) for k, v in cams_host.items()}

        def step(mode):
            if mode == 'cached':
                cams = cams_dev
            elif mode == 'host_inverses':
                fresh = add_camera_inverses({k: v.clone() for k, v in cams_host.items()})
                cams = {k: v.pin_memory().to(dev, non_blocking=True) for k, v in fresh.items()}
            else:
                cams = {k: v.clone().to(dev, non_blocking=True) for k, v in cams_host.items()}
            return m.predict(dict(big, **cams), None)
        leg = {}
        for mode in ('cached', 'host_inverses', 'device_cameras'):
            for _ in range(5):
                step(mode)
            m.bg.settle()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step(mode)
            m.bg.settle()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            leg[mode] = {'value': b * steps / dt, 'unit': 'frames/s', 'ms_per_step': 1e3 * dt / steps}
        leg['steps'], leg['launch'] = steps, 'eager, one stream'
        res[str(b)] = leg
        del m, big, cams_dev
        torch.cuda.empty_cache()
    return res


def pipelined_leg(sd, dev, term, steps_by_b=((1, 480), (2, 360)), depth=2):
    """the reference-shaped loop (export_cityscapes_segmentation_results.py:75-85: loader -> batch2gpu -> predict
    -> write) is a STREAM of batches, so it can be double-buffered without touching predict's contract: two model replicas on two
    HIP streams, batch k + 1 enqueued while batch k runs (what export_bg.py --pipeline_depth 2 does).  Eager launches, the big
    tensors resident, the camera tensors rebuilt from host memory with host-side inverses every batch (add_camera_inverses) - no
    graph, no cache hit, no synchronisation inside the loop.  by_batch stays the latency figure of ONE forward at a time."""
    from panoptic_forecasting_amd.pc_transform_model import add_camera_inverses
    cam_keys = ('intrinsics', 'extrinsics', 'target_T')
    res = {}
    for b, steps in steps_by_b:
        models = []
        for _ in range(depth):
            m = build_model(model_params())
            m.load_state_dict(sd)
            m.eval()
            models.append(m)
        host = {k: torch.cat([synth.make_inputs(b=1, t=T, h=H, w=W, seed=i, **TERM[term])[k] for i in range(b)], 0)
                for k in ('depth', 'depth_mask', 'seg') + cam_keys}
        big = {k: host[k].to(dev) for k in ('depth', 'depth_mask', 'seg')}
        cams_host = {k: host[k].pin_memory() for k in cam_keys}
        streams = [torch.cuda.Stream() for _ in range(depth)]
        outs = [None] * depth

        def run(n):
            cur = torch.cuda.current_stream()
            for st in streams:
                st.wait_stream(cur)
            for k in range(n):
                i = k % depth
                fresh = add_camera_inverses({kk: v.clone() for kk, v in cams_host.items()})
                with torch.cuda.stream(streams[i]):
                    cams = {kk: v.pin_memory().to(dev, non_blocking=True) for kk, v in fresh.items()}
                    outs[i] = models[i].predict(dict(big, **cams), None)
            for st in streams:
                cur.wait_stream(st)
        run(10)
        for m in models:
            m.bg.settle()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        for m in models:
            m.bg.settle()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[str(b)] = {'value': b * steps / dt, 'unit': 'frames/s', 'ms_per_batch': 1e3 * dt / steps, 'steps': steps,
                       'streams': depth, 'launch': 'eager', 'cameras': 'fresh, host inverses',
                       'reruns': sum(m.bg.range_reruns for m in models)}
        del models, big, outs
        torch.cuda.empty_cache()
    return res


def train_step_in_fresh_process():
    """`train_step_leg` in a process of its own (python bench.py --train-step-only).  Not for cleanliness: by this point this
    process has created a dozen HIP streams (the sub-batch streams of every leg, torch's stream pool, capture streams), and
    the runtime multiplexes streams onto a handful of hardware queues - the training plan's weight-gradient stream then shares
    a queue with the stream it forks from and joins, every cross-stream edge becomes a barrier packet in that one queue, and
    the step measures 27-28 ms instead of 14.8 (same kernels, same box; a process that trains has two streams)."""
    env = dict(os.environ)
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--train-step-only'], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, timeout=900)
        lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if out.returncode != 0 or not lines:
            return {'error': 'train-step process failed (rc %d): %s' % (out.returncode, out.stderr[-400:])}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {'error': 'train-step process timed out'}


def other_resolution_leg(sd, dev, term, steps):
    """A second resolution, 512x1024 at B = 16 on one stream (hipGraph replay): no row of the measured kernel tables
    (csrc/conv_s4_tuned.inc, conv_tuned.inc: keyed on the 1024x2048 network's exact shapes) was measured here, so every layer
    runs what conv_select.cpp's rule for unmeasured sizes picks (the row of the same layer whose measured launch had as many
    pixels, else the heuristics) - frames/s plus the dominant kernel's roofline fraction of that choice.  Parity at this size: tests/test_gpu_bg_forecast.py::test_second_resolution_heuristic_kernel_choice_vs_oracle."""
    h, w, b = 512, 1024, 16
    parts = [synth.make_inputs(b=1, t=T, h=h, w=w, seed=i, **TERM[term]) for i in range(b)]
    batch = {k: torch.cat([p[k] for p in parts], 0).to(dev) for k in parts[0]}
    m = build_model(model_params(final_h=h, final_w=w))
    m.load_state_dict(sd)
    m.eval()
    for _ in range(3):
        m.predict(batch, None)
    m.bg.settle()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        m.predict(batch, None)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sticky = m.bg.range_status_sticky(clear=True)
    recs = profile_records(lambda: m.predict(batch, None), 2)
    m.bg.settle()
    r = roofline_of(recs, 2, b, 'one batch of %d frames at %dx%d, eager, hipEvents' % (b, h, w))
    res = {'value': b * steps / dt, 'unit': 'frames/s', 'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'frames_per_step': b,
           'size': [h, w], 'streams': 1, 'launch': 'hipGraph replay', 'kernel_choice': 'unmeasured size: rows of the same layer at the batch whose measured launch had as many pixels (here B = 4), else heuristics',
           'range_status_sticky': int(sticky),
           'roofline': {k: r[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'kernel', 'avg_launch_us', 'launches_per_step',
                                          'kernel_ms_per_step')},
           'roofline_step_frac': r['step']['frac']}
    del m, batch, g
    torch.cuda.empty_cache()
    return res, sticky


def train_step_leg(sd, steps=10):
    """Scope row f4, driver-timed: one bg training step (forward with batch statistics + loss + backward + clipped SGD,
    training/train.py:185-222) on the reference's configuration - batch 8 of 800x800 crops, 3 input frames
    (configs/bg/bg_train.yaml:25,48) - with the library's defaults (eager launches, weight gradients on the plan's own stream).
    The same configuration is parity-tested against the oracle in tests/test_gpu_train.py::test_timed_configuration_800x800_vs_oracle."""
    from panoptic_forecasting_amd import lib as pflib
    from panoptic_forecasting_amd.bg_train import BGTrainer
    b, size = 8, 800
    params = {'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])]},
              'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True},
              'training': {'lr': 2e-3, 'mom': 0.9, 'wd': 1e-4, 'clip_grad_norm': 5.0}}
    tr = BGTrainer(params)
    tr.load_state_dict(sd)
    inp = {k: v.cuda() for k, v in synth.make_bg_inputs(b=b, h=size, w=size, seed=1).items()}
    inp['seg'] = inp['seg'].to(torch.uint8)
    lab = {'seg': torch.randint(0, 11, (b, size, size), dtype=torch.uint8, device='cuda')}
    for _ in range(3):
        tr.train_step(inp, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = tr.train_step(inp, lab)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    stats = tr.path_stats()
    recs = profile_records(lambda: tr.train_step(inp, lab), 1)
    dom = max(recs, key=lambda r: r['ms'])
    tf = dom['flops'] / max(dom['ms'], 1e-9) / 1e9
    res = {'ms_per_step': ms, 'samples_per_s': b / ms * 1e3, 'batch': b, 'size': size, 'steps': steps, 'dtype': 'f32',
           'launch': 'eager, weight gradients on their own stream' if tr.side_stream else 'eager, one stream',
           'loss': float(out['loss']), 'workspace_GB': tr._ws.numel() / 1e9,
           'kernel_ms_sum': sum(r['ms'] for r in recs), 'launches': sum(r['launches'] for r in recs), 'path_stats': stats,
           'dominant_kernel': {'kernel': dom['label'], 'ms_per_step': dom['ms'], 'launches': dom['launches'], 'TFLOPs': tf,
                               'peak': PEAK_FP32_MFMA_TFLOPS, 'frac': tf / PEAK_FP32_MFMA_TFLOPS, 'bound': 'mfma',
                               'peak_note': 'fp32 MFMA (= fp32 vector) peak; algorithmic flops of the launches / their hipEvent time'}}
    del tr, inp, lab
    torch.cuda.empty_cache()
    return res


STAGES = (('splat', ('bin_kernel', 'raster_kernel', 'splat')), ('stem', ('stem',)), ('head', ('head',)),
          ('convs', ('conv_',)))


def stage_of(label):
    for name, keys in STAGES:
        if any(k in label for k in keys):
            return name
    return 'other'


def roofline_of(recs, n_steps, frames, measured_on):
    """Dominant kernel + whole-step accounting from the live hipEvent records [{label, launches, ms, flops, bytes}]."""
    tot = sum(r['ms'] for r in recs)
    dom = max(recs, key=lambda r: r['ms'])
    per_launch_ms = dom['ms'] / dom['launches']
    gbs = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9
    if dom['flops'] > 0:
        achieved = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
        if 'conv_split' in dom['label'] or 'conv_s4' in dom['label']:
            # every algorithmic fp32 MAC is 3 fp16 MFMA MACs (hi*hi + hi*mid + mid*hi): the matrix ceiling of this
            # scheme, in algorithmic flops, is the dense 16-bit peak / 3
            peak, note = PEAK_BF16_MFMA_TFLOPS / 3.0, 'dense fp16/bf16 MFMA peak 2500 TFLOP/s / 3 products per fp32 MAC'
        else:
            peak, note = PEAK_FP32_MFMA_TFLOPS, 'fp32 MFMA (= fp32 vector) peak'
        mf, hf = achieved / peak, gbs / PEAK_HBM_GBPS
        # both fractions are reported; `bound` names the larger one
        if mf >= hf:
            roofline = {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': mf, 'peak_note': note,
                        'hbm_frac': hf, 'hbm_GBps_algorithmic': gbs}
        else:
            roofline = {'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': hf,
                        'mfma_frac': mf, 'mfma_TFLOPs_algorithmic': achieved, 'mfma_peak': peak, 'peak_note': note}
    else:
        roofline = {'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBPS}
    conv = [r for r in recs if r['flops'] > 0]
    conv_ms = sum(r['ms'] for r in conv)
    traffic, tnote = pmc_traffic(dom['label'])
    roofline.update({'traffic': traffic, 'traffic_note': tnote, 'kernel': dom['label'],
                     'algorithmic_bytes_per_launch': dom['bytes'] / dom['launches'],
                     'launches_per_step': dom['launches'] // n_steps,
                     'avg_launch_us': per_launch_ms * 1e3, 'share_of_step': dom['ms'] / tot,
                     'all_conv_tflops': sum(r['flops'] for r in conv) / (conv_ms * 1e-3) / 1e12 if conv_ms else None,
                     'kernel_ms_per_step': tot / n_steps, 'measured_on': measured_on})
    # whole step: sum of the algorithmic bytes of every launch / sum of kernel time, against the HBM peak
    stages = {}
    for r in recs:
        st = stages.setdefault(stage_of(r['label']), {'ms': 0.0, 'bytes': 0.0, 'flops': 0.0})
        st['ms'] += r['ms'] / n_steps
        st['bytes'] += r['bytes'] / n_steps
        st['flops'] += r['flops'] / n_steps
    for st in stages.values():
        st['GBps'] = st['bytes'] / (st['ms'] * 1e-3) / 1e9 if st['ms'] > 0 else None
        st['hbm_frac'] = st['GBps'] / PEAK_HBM_GBPS if st['GBps'] else None
        st['MB_per_frame'] = st.pop('bytes') / frames / 1e6
        st['GFLOP_per_frame'] = st.pop('flops') / frames / 1e9
    tot_bytes = sum(r['bytes'] for r in recs) / n_steps
    roofline['step'] = {'bound': 'hbm', 'algorithmic_MB_per_frame': tot_bytes / frames / 1e6,
                        'kernel_ms': tot / n_steps, 'achieved': tot_bytes / (tot / n_steps * 1e-3) / 1e9,
                        'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                        'frac': tot_bytes / (tot / n_steps * 1e-3) / 1e9 / PEAK_HBM_GBPS, 'stages': stages}
    return roofline


def profile_records(fn, n_steps):
    from panoptic_forecasting_amd import lib as pflib
    pflib.profile(True)
    for _ in range(n_steps):
        fn()
    torch.cuda.synchronize()
    recs = pflib.profile_results()
    pflib.profile(False)
    return recs


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=128, help='resident forecast frames per GPU, run as --streams concurrent '
                    'sub-batches (the reference export loop batches 2; throughput saturates at 128 frames in flight as four '
                    'staggered sub-batches of 32)')
    ap.add_argument('--replays', type=int, default=2, help='passes over the resident batch per step (one hipGraph): a step is '
                    '--replays x --batch forecast frames per GPU (20 steps then time about a second)')
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a hipGraph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-legs', action='store_true', help='skip the by_batch / fp32_only legs (N=1 only)')
    ap.add_argument('--cpu-frames', type=int, default=12)
    ap.add_argument('--profile-steps', type=int, default=3)
    ap.add_argument('--fp32-mfma-only', action='store_true', help='headline itself without the split kernels '
                    "(model param split_f16=0): every convolution on v_mfma_f32_16x16x4_f32 / the fp32 VALU")
    ap.add_argument('--streams', type=int, default=0, help='sub-batches run concurrently on this many HIP streams '
                    '(0 = one per 16 frames of the batch)')
    ap.add_argument('--stagger', type=int, default=1, help='1: sub-batch i + 1 of a step starts when the warp/splat of '
                    'sub-batch i is done (software pipeline inside the step) instead of all sub-batches starting together')
    ap.add_argument('--free-run', type=float, default=None, metavar='MS', help='experiment: one hipGraph per sub-batch on its '
                    'own stream, no join between steps, sub-batch i starts i * MS late (streams out of phase)')
    ap.add_argument('--term', choices=['short', 'mid'], default='short',
                    help="short = BASELINE configs[1] (dt=3, the headline); mid = configs[2] (dt=9, predicted odometry)")
    ap.add_argument('--train-step-only', action='store_true', help='print the train_step object alone (the N=1 line runs this in a fresh process)')
    ap.add_argument('--verbose', action='store_true', help='keep the per-stage breakdown and every detail of the legs in the line (> 6 KB)')
    ap.add_argument('--dry-run', action='store_true', help='rendezvous + the sharded metric exchange only (gloo, no GPU '
                    'work): checks that --gpus N really starts N ranks')
    return ap.parse_args(argv)


def relaunch_under_torchrun(args):
    """python bench.py --gpus N (N > 1) from a plain shell: become the launcher of N ranks."""
    if not args.dry_run and torch.cuda.device_count() < args.gpus and not SHARE_GPU:
        raise SystemExit('bench.py: --gpus %d but this node has %d GPU(s)' % (args.gpus, torch.cuda.device_count()))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC (RCCL across processes needs it on this pool)
    env.setdefault('OMP_NUM_THREADS', '4')
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse_args()
    if args.train_step_only:
        torch.cuda.set_device(0)
        print(json.dumps(train_step_leg(calibrated_state_dict())))
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        relaunch_under_torchrun(args)
    rank, world, local = pfdist.init_distributed_mode(backend='gloo' if (args.dry_run or SHARE_GPU) else None)
    if SHARE_GPU and not args.dry_run:
        local = local % max(1, torch.cuda.device_count())
    joined = torch.distributed.get_world_size() if pfdist.is_dist() else 1
    if world != args.gpus or joined != args.gpus:
        raise SystemExit('bench.py: --gpus %d but %d rank(s) joined (WORLD_SIZE=%d)' % (args.gpus, joined, world))

    if args.dry_run:
        from panoptic_forecasting_amd import pq as pfpq
        acc = torch.zeros(11, 4, dtype=torch.float64)
        acc[:, 1] = rank + 1           # every rank contributes a different count
        allacc = pfdist.gather_accumulators(acc)
        ok = allacc.shape[0] == world and float(allacc[:, 0, 1].sum()) == world * (world + 1) / 2
        per_rank = pfdist.gather_objects({'rank': rank, 'ms_per_step': 0.0, 'device': 'cpu:%d' % local})
        ok = ok and [r['rank'] for r in per_rank] == list(range(world)) and len({r['device'] for r in per_rank}) == world
        if rank == 0:
            print(json.dumps({'metric': 'forecast frames/sec @1024x2048, 3-in->dt=3 bg', 'value': None, 'unit': 'frames/s',
                              'n_gpus': joined, 'dry_run': True, 'backend': 'gloo', 'gather_ok': bool(ok),
                              'per_rank_ms': [r['ms_per_step'] for r in per_rank], 'devices': [r['device'] for r in per_rank],
                              'sharding': 'batch over %d rank(s), no data-path collective' % world}))
        if pfdist.is_dist():
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        if not ok:
            raise SystemExit(1)
        return

    if torch.cuda.device_count() < (local + 1):
        raise SystemExit('bench.py: rank %d (local %d) has no GPU: %d visible' % (rank, local, torch.cuda.device_count()))
    from panoptic_forecasting_amd import lib as pflib
    from panoptic_forecasting_amd import pq as pfpq
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    L = pflib.load()   # fails loudly if libpfhip.so is missing
    for kv in filter(None, os.environ.get('PF_OPTS', '').split(',')):    # A/B runs: PF_OPTS=fuse_upsample=1,...
        k, v = kv.split('=')
        pflib.check(L.pf_set_option(k.encode(), int(v)), 'pf_set_option')

    sd = calibrated_state_dict()
    B = args.batch
    S = max(1, min(args.streams, B)) if args.streams > 0 else max(1, B // SUB_BATCH)
    use_graph = not args.no_graph
    head_kw = {'split_f16': 0} if args.fp32_mfma_only else {}
    R = max(1, args.replays)
    wl = Workload(sd, B, S, dev, seed0=rank * B, term=args.term, use_graph=use_graph, stagger=bool(args.stagger),
                  free_run_ms=args.free_run, passes=R, **head_kw)
    elapsed = wl.timed(args.steps, args.warmup, dev, barrier=True)
    frames = world * B * R * args.steps
    value = frames / elapsed
    sticky = wl.range_overflow()
    reruns = sum(m.bg.range_reruns for m in wl.models)
    # a flag raised inside a captured graph cannot have been acted on (nothing waits or re-runs under replay): the number is
    # invalid.  An EAGER forward that was flagged has already been re-run on the fp32 matrix instructions into the same
    # outputs (the 'rerun' default, counted in range_reruns): valid results - at the re-run's cost, which the line then shows
    overflow = sticky if use_graph else 0
    # every rank's own step time and device identity: a straggler, or two ranks on one device, shows in the N > 1 line
    ident = pfdist.device_identity(local)
    per_rank = pfdist.gather_objects({'rank': rank, 'ms_per_step': 1e3 * wl.local_s / args.steps, 'device': ident})
    if len({r['device'] for r in per_rank}) != len(per_rank) and not SHARE_GPU:
        raise SystemExit('bench.py: %d ranks share devices: %s' % (len(per_rank), [r['device'] for r in per_rank]))

    # ---- sharded metric exchange: PQ accumulators of this rank's forecasts vs a synthetic ground truth
    out = wl.outputs()   # the sub-batch outputs are concatenated outside the timed region
    gt = torch.from_numpy(synth.ID2TRAINID).to(dev)[wl.batch['seg'][:, T - 1].long()].long()
    acc = pfpq.pq_accumulate(out['seg'].long(), gt, 11)
    allacc = pfdist.gather_accumulators(acc)
    if int(allacc.shape[0]) != args.gpus:
        raise SystemExit('bench.py: the PQ all-gather returned %d rank(s), --gpus %d' % (int(allacc.shape[0]), args.gpus))
    pq_synth = pfpq.pq_from_acc(allacc.sum(0))['pq']

    # ---- per-kernel timing pass (eager, hipEvents on the launch stream) -> roofline of the dominant kernel + whole step
    roofline = None
    if rank == 0:
        # one sub-batch alone on the launch stream: kernels of concurrent streams share the chip, which would inflate
        # the per-launch durations the roofline fraction is computed from
        recs = profile_records(lambda: wl.models[0].predict(wl.subs[0], None), args.profile_steps)
        roofline = roofline_of(recs, args.profile_steps, B // S,
                               'one sub-batch of %d frames alone on the launch stream (eager, hipEvents)' % (B // S))
        if os.environ.get('PF_BENCH_KERNELS'):
            for r in sorted(recs, key=lambda r: -r['ms']):
                print('# %-70s n=%3d %8.3f ms  %7.2f TF/s %8.1f GB/s' % (
                    r['label'][:70], r['launches'] // args.profile_steps, r['ms'] / args.profile_steps,
                    r['flops'] / max(r['ms'], 1e-9) / 1e9, r['bytes'] / max(r['ms'], 1e-9) / 1e6), file=sys.stderr)

    cpu = parity = by_batch = by_batch_pipelined = fp32_only = fresh_cameras = train_step = other_resolution = None
    single = rank == 0 and world == 1
    ref = None
    n_done = 0
    if single and not args.no_cpu_baseline:
        cpu, ref, n_done = cpu_baseline(sd, args.cpu_frames, args.term)

    def parity_of(sub, model_kw):
        """Full-size parity of the LAST cpu frame (seed n_done-1) against the HIP path, computed inside the SAME
        16-frame sub-batch the timed region runs (the per-layer kernel choice depends on the batch size)."""
        idx = n_done - 1
        if idx >= sub['depth'].shape[0]:
            sub, idx = make_batch(1, seed0=idx, device=dev, term=args.term), 0
        lm = build_model(model_params(return_logits='orig', **model_kw))
        lm.load_state_dict(sd)
        lm.eval()
        res = lm.predict(sub, None)
        got = res['seg'][idx:idx + 1].long().cpu()
        agree = float((got == ref['seg']).float().mean())
        pq_ref = pfpq.pq_from_acc(pfpq.pq_accumulate(got, ref['seg'], 11))['pq']
        dlogit = float((res['orig_size_logits'][idx:idx + 1].cpu() - ref['orig_size_logits']).abs().max())
        return {'argmax_agreement_vs_oracle': agree, 'pq_vs_oracle_as_gt': pq_ref, 'max_abs_dlogit_vs_oracle': dlogit,
                'logit_tolerance': 1e-3, 'checked_in_batch_of': int(sub['depth'].shape[0])}

    sub0 = wl.subs[0]
    if ref is not None:
        parity = parity_of(sub0, head_kw)

    if single and not args.no_legs:
        shared_batch = wl.batch          # the fp32-only leg runs the same resident frames (no second synthetic generation)
        del wl
        torch.cuda.empty_cache()
        # SURVEY.md 8d Config 2: B in {1, 2, 4} frames per step on ONE stream (B=1 is the latency configuration)
        by_batch = {}
        leg_steps = max(args.steps, 30) * 16        # 480 steps: the B = 1 leg times about half a second
        for b in (1, 2, 4):
            leg = Workload(sd, b, 1, dev, seed0=0, term=args.term, use_graph=use_graph, **head_kw)
            dt = leg.timed(leg_steps, args.warmup, dev)
            by_batch[str(b)] = {'value': b * leg_steps / dt, 'unit': 'frames/s', 'ms_per_step': 1e3 * dt / leg_steps,
                                'steps': leg_steps, 'streams': 1}
            st = leg.range_overflow()
            sticky |= st
            overflow |= st if use_graph else 0
            reruns += sum(m.bg.range_reruns for m in leg.models)
            del leg
            torch.cuda.empty_cache()
        by_batch_pipelined = pipelined_leg(sd, dev, args.term)
        deeper = pipelined_leg(sd, dev, args.term, depth=3)          # export_bg.py --pipeline_depth 3 (its default)
        for k, v in deeper.items():
            by_batch_pipelined[k + '_depth3'] = v
        reruns += sum(v['reruns'] for v in by_batch_pipelined.values())
        fresh_cameras = fresh_cameras_leg(sd, dev, args.term)
        if not args.fp32_mfma_only:
            other_resolution, st = other_resolution_leg(sd, dev, args.term, max(args.steps, 30) * 4)
            sticky |= st
            overflow |= st
        if not args.fp32_mfma_only:
            # fp32-instruction configuration: no two-term fp16 operands anywhere (fp32 MFMA / fp32 VALU only)
            leg = Workload(sd, B, S, dev, seed0=0, term=args.term, use_graph=use_graph, stagger=bool(args.stagger), passes=R,
                           batch=shared_batch, split_f16=0)
            dt = leg.timed(args.steps, args.warmup, dev)
            fp32_only = {'value': B * R * args.steps / dt, 'unit': 'frames/s', 'ms_per_step': 1e3 * dt / args.steps,
                         'frames_per_gpu_per_step': B * R, 'streams': S, 'dtype': 'f32'}
            recs = profile_records(lambda: leg.models[0].predict(leg.subs[0], None), args.profile_steps)
            r32 = roofline_of(recs, args.profile_steps, B // S, 'one sub-batch, eager, hipEvents')
            fp32_only['roofline'] = {k: r32[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'kernel', 'avg_launch_us',
                                                         'kernel_ms_per_step')}
            fp32_only['roofline_step_frac'] = r32['step']['frac']
            if ref is not None:
                p32 = parity_of(leg.subs[0], {'split_f16': 0})
                fp32_only['max_abs_dlogit'] = p32['max_abs_dlogit_vs_oracle']
                fp32_only['argmax_agreement_vs_oracle'] = p32['argmax_agreement_vs_oracle']
            del leg
            torch.cuda.empty_cache()
        del shared_batch
        torch.cuda.empty_cache()
        train_step = train_step_in_fresh_process()

    if rank == 0:
        r3 = lambda x: None if x is None else round(float(x), 3)
        launch = (('hipGraph per sub-batch, free-running streams (+%g ms)' % args.free_run) if (args.free_run is not None and use_graph and S > 1)
                  else (('hipGraph replay' if use_graph else 'eager') + (', sub-batches staggered' if (args.stagger and S > 1) else '')))
        config = {'workload': 'configs[%d] bg %s-term forecast 3-in dt=%d @1024x2048: 3 warp/splats + HarDNet-70 + upsample/argmax' % (
                      (1, 'short', 3) if args.term == 'short' else (2, 'mid', 9)),
                  'frames_per_gpu_per_step': B * R, 'resident_frames_per_gpu': B, 'passes_per_step': R, 'streams': S,
                  'sub_batch': B // S, 'launch': launch, 'weights': 'random-init, calibrated (tests/golden/calib_seed1234.json)',
                  'model': "registry.build_model(task 'bg_forecast'), default options (on_range_overflow='rerun', per-sample sentinel)",
                  'sharding': 'batch over %d rank(s), no data-path collective' % world, 'world': joined,
                  'device': torch.cuda.get_device_name(local), 'backend': pfdist.backend_description()}
        if roofline is not None:
            # scalars the driver's record keeps (nested objects are dropped there): see the module docstring
            roofline['step_frac'] = roofline['step']['frac']
            roofline['step_kernel_ms'] = roofline['step']['kernel_ms']
            for name, st in roofline['step']['stages'].items():
                roofline['stage_ms_' + name] = r3(st['ms'])
            if parity is not None:
                roofline['parity_max_abs_dlogit'] = parity['max_abs_dlogit_vs_oracle']
                roofline['parity_argmax_agreement'] = parity['argmax_agreement_vs_oracle']
            if fp32_only is not None:
                roofline.update({'fp32_only_value': r3(fp32_only['value']), 'fp32_only_frac': fp32_only['roofline']['frac'],
                                 'fp32_only_bound': fp32_only['roofline']['bound'], 'fp32_only_kernel': fp32_only['roofline']['kernel'][:80],
                                 'fp32_only_step_frac': fp32_only['roofline_step_frac'],
                                 'fp32_only_max_abs_dlogit': fp32_only.get('max_abs_dlogit')})
            if train_step is not None and 'ms_per_step' in train_step:
                roofline.update({'train_step_ms': r3(train_step['ms_per_step']), 'train_step_dominant_frac': train_step['dominant_kernel']['frac'],
                                 'train_step_dominant_kernel': train_step['dominant_kernel']['kernel'][:60]})
            if other_resolution is not None:
                roofline.update({'other_resolution_value': r3(other_resolution['value']), 'other_resolution_frac': other_resolution['roofline']['frac'],
                                 'other_resolution_kernel': other_resolution['roofline']['kernel'][:60]})
            if not args.verbose:
                for k in ('step', 'peak_note', 'traffic_note', 'measured_on'):
                    roofline.pop(k, None)
        if by_batch is not None:
            config['by_batch'] = ', '.join('B%s %.0f' % (k, v['value']) for k, v in by_batch.items()) + ' frames/s (1 stream, graph)'
        if by_batch_pipelined is not None:
            config['by_batch_pipelined'] = ', '.join('B%s %.0f' % (k.replace('_depth3', ' (3 streams)'), v['value']) for k, v in by_batch_pipelined.items()) + \
                ' frames/s (2 streams unless said, eager, fresh cameras)'
        if fresh_cameras is not None:
            config['fresh_cameras'] = '; '.join('B%s ' % k + ' / '.join('%.0f' % v[m_]['value'] for m_ in ('cached', 'host_inverses', 'device_cameras'))
                                                for k, v in fresh_cameras.items()) + ' (cached / host inverses / device cameras)'
        if cpu is not None and not args.verbose:
            cpu = {k: v for k, v in cpu.items() if k != 'network_s_per_frame_by_threads'}
            cpu['sample'] = cpu['sample'][:118]

        def slim(leg, keep):
            return None if leg is None else {k: leg[k] for k in keep if k in leg}
        if not args.verbose:
            if by_batch is not None:
                by_batch = {k: r3(v['value']) for k, v in by_batch.items()}
            if by_batch_pipelined is not None:
                by_batch_pipelined = {k: r3(v['value']) for k, v in by_batch_pipelined.items()}
            if fresh_cameras is not None:
                fresh_cameras = {k: {m_: r3(v[m_]['value']) for m_ in ('cached', 'host_inverses', 'device_cameras')} for k, v in fresh_cameras.items()}
            if fp32_only is not None:
                fp32_only = dict(slim(fp32_only, ('value', 'unit', 'ms_per_step', 'dtype', 'max_abs_dlogit', 'argmax_agreement_vs_oracle', 'roofline_step_frac')),
                                 frac=fp32_only['roofline']['frac'], bound=fp32_only['roofline']['bound'], kernel=fp32_only['roofline']['kernel'][:60])
            if other_resolution is not None:
                other_resolution = dict(slim(other_resolution, ('value', 'unit', 'ms_per_step', 'frames_per_step', 'size', 'roofline_step_frac')),
                                        frac=other_resolution['roofline']['frac'], kernel=other_resolution['roofline']['kernel'][:60])
            if train_step is not None and 'ms_per_step' in train_step:
                train_step = dict(slim(train_step, ('ms_per_step', 'samples_per_s', 'batch', 'size', 'dtype', 'loss', 'kernel_ms_sum', 'launches')),
                                  dominant_kernel=train_step['dominant_kernel']['kernel'][:60], dominant_frac=train_step['dominant_kernel']['frac'],
                                  dominant_TFLOPs=r3(train_step['dominant_kernel']['TFLOPs']))
        line = {'metric': 'forecast frames/sec @1024x2048, 3-in->dt=%d bg' % (3 if args.term == 'short' else 9), 'value': value, 'unit': 'frames/s',
                'n_gpus': joined, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                # (the operand bound and its proof: DESIGN.md 4; the strict-fp32-instruction number: roofline.fp32_only_value)
                'dtype': 'f32' if args.fp32_mfma_only else 'f32 as 2 fp16 terms/operand, 3 MFMA products, f32 accumulate (strict f32: fp32_only)',
                'data': 'synthetic',
                'range_overflow': bool(overflow), 'range_status_sticky': int(sticky), 'range_reruns': int(reruns),
                'config': config,
                'per_rank_ms': [r3(r['ms_per_step']) for r in per_rank],
                'devices': [r['device'] for r in per_rank],
                'roofline': roofline, 'cpu_baseline': cpu, 'parity': parity, 'by_batch': by_batch, 'by_batch_pipelined': by_batch_pipelined,
                'fresh_cameras': fresh_cameras, 'other_resolution': other_resolution, 'fp32_only': fp32_only, 'train_step': train_step,
                'pq_gather_check': {'pq_vs_last_input_labels': pq_synth, 'ranks_gathered': int(allacc.shape[0])}}
        print(json.dumps(line))
    if pfdist.is_dist():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if overflow:
        raise SystemExit('bench.py: range status %d was raised inside the timed region (1: the fp16-pair path met |x| > 65504, '
                         '2: a tensor of tiny values); the number above is not a valid measurement' % overflow)


if __name__ == '__main__':
    main()
