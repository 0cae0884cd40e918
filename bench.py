#!/usr/bin/env python
"""Headline benchmark: bg forecast frames/sec @1024x2048, 3 inputs -> Δt=3 (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch of synthetic, HBM-resident inputs on every rank:
    3 x (unproject + ego warp + z-buffered splat)  ->  on-the-fly disk-hop emulation  ->  FC-HarDNet-70
    -> bilinear upsample + argmax   (task ``bg_forecast`` of panoptic-forecasting_amd)
Ranks are independent (sequences shard by batch); the only collective is the end-of-run all-gather of
the PQ accumulators.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-graph] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from panoptic_forecasting_amd import dist as pfdist  # noqa: E402
from panoptic_forecasting_amd import lib as pflib  # noqa: E402
from panoptic_forecasting_amd import pq as pfpq  # noqa: E402
from panoptic_forecasting_amd import synth  # noqa: E402
from panoptic_forecasting_amd.pc_transform_model import host_inverse  # noqa: E402
from panoptic_forecasting_amd.registry import build_model as _build_model  # noqa: E402


def build_model(params):
    """The registry prints like the reference's does (models/__init__.py:18); stdout carries only the JSON line here."""
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        return _build_model(params)

H, W, T = 1024, 2048, 3
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, dense
PEAK_HBM_GBPS = 8000.0          # HBM3E spec (6.3 TB/s achievable)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 (2495 measured)
CPU_THREADS = 32                # torch-CPU threads for the baseline leg (more oversubscribes these small convs)
CPU_BUDGET_S = 20.0             # stop starting new baseline frames after this much CPU time


def calibrated_state_dict():
    with open(os.path.join(ROOT, 'tests', 'golden', 'calib_seed1234.json')) as f:
        calib = json.load(f)
    return synth.make_state_dict(seed=1234, calib=calib)


def model_params():
    return {'task': 'bg_forecast', 'no_gpu': False, 'load_model': None, 'load_best_model': False,
            'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])],
                     'min_depth': 0.1, 'max_depth': 200},
            'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True,
                      'final_h': H, 'final_w': W, 'emulate_disk_hop': True, 'seg_is_label_id': True}}


TERM = {'short': dict(gap_len=3, predicted=False), 'mid': dict(gap_len=9, predicted=True)}   # configs[1] / configs[2]


def make_batch(b, seed0, device, term='short'):
    parts = [synth.make_inputs(b=1, t=T, h=H, w=W, seed=seed0 + i, **TERM[term]) for i in range(b)]
    inp = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}
    # camera inverses are per-sequence constants prepared with the inputs (host LAPACK)
    inp['intrinsics_inv'] = host_inverse(inp['intrinsics'])
    inp['extrinsics_inv'] = host_inverse(inp['extrinsics'])
    return {k: v.to(device) for k, v in inp.items()}


def pmc_traffic(kernel_label):
    """HBM bytes per launch of `kernel_label` from the committed PMC passes (profiles/pmc_latest.json: FETCH_SIZE +
    WRITE_SIZE collected in separate rocprofv3 --pmc runs of this same command and calibrated on a known-size copy,
    tools/profile_gpu.sh + tools/profile_summarise.py), or None when that kernel was not profiled."""
    path = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
    if not os.path.exists(path):
        return None
    with open(path) as f:
        # labels are the rocprofv3 symbol + optional " +res"/" +pool"/" lowres-half" annotations of the launch
        k = json.load(f).get('kernels', {}).get(kernel_label.split(')')[0] + ')')
    return k.get('hbm_bytes_per_launch') if k else None


def cpu_baseline(sd, n_frames, term='short'):
    """The oracle (CPU port of the reference path) timed on this box's host cores."""
    import numpy as np
    from oracle import hardnet_ref
    from oracle import warp_splat as ow
    torch.set_num_threads(min(os.cpu_count(), CPU_THREADS))
    ow.lib()
    last = None
    t0 = time.perf_counter()
    for f in range(n_frames):
        if f > 0 and time.perf_counter() - t0 > CPU_BUDGET_S:
            n_frames = f
            break
        inp = synth.make_inputs(b=1, t=T, h=H, w=W, seed=f, **TERM[term])
        segs, deps = [], []
        for t in range(T):
            o = ow.predict(inp, only_this_ind=t)
            tid = torch.from_numpy(synth.ID2TRAINID)[o['seg'].long()]
            q = ((o['depth'] + 1).clamp(0, 255) * 256).round().numpy().astype(np.uint16)
            d = torch.from_numpy(q.astype(np.float32)) / 256.0 - 1
            m = d > 0
            d[~m] = -1
            d[m & (d > 200)] = 200
            d[m & (d < 0.1)] = 0.1
            segs.append(tid)
            deps.append(d)
        seg, dep = torch.stack(segs, 1).long(), torch.stack(deps, 1)
        last = hardnet_ref.bg_predict(sd, {'seg': seg, 'depth': dep, 'depth_mask': dep > 0}, final_size=(H, W))
    dt = time.perf_counter() - t0
    # the generation of synthetic inputs is inside the loop but is <3 % of it
    return n_frames / dt, dt, last, n_frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='forecast frames per GPU per step (the reference export loop '
                    'batches 2; throughput saturates around 32-48 frames in flight as two or three concurrent sub-batches of 16)')
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a hipGraph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-frames', type=int, default=12)
    ap.add_argument('--profile-steps', type=int, default=3)
    ap.add_argument('--fp32-mfma-only', action='store_true', help='disable the bf16-split 3x3 kernels (pf_set_option split_bf16=0): '
                    'every convolution on v_mfma_f32_16x16x4_f32 / the fp32 VALU')
    ap.add_argument('--streams', type=int, default=0, help='sub-batches run concurrently on this many HIP streams '
                    '(0 = one per 16 frames of the batch)')
    ap.add_argument('--term', choices=['short', 'mid'], default='short',
                    help="short = BASELINE configs[1] (dt=3, the headline); mid = configs[2] (dt=9, predicted odometry)")
    args = ap.parse_args()

    rank, world, local = pfdist.init_distributed_mode()
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    L = pflib.load()   # fails loudly if libpfhip.so is missing
    for kv in filter(None, os.environ.get('PF_OPTS', '').split(',')):    # A/B runs: PF_OPTS=fuse_upsample=1,...
        k, v = kv.split('=')
        pflib.check(L.pf_set_option(k.encode(), int(v)), 'pf_set_option')

    if args.fp32_mfma_only:
        pflib.check(L.pf_set_option(b'split_bf16', 0), 'pf_set_option')
    sd = calibrated_state_dict()
    model = build_model(model_params())
    model.load_state_dict(sd)
    model.eval()
    B = args.batch
    batch = make_batch(B, seed0=rank * B, device=dev, term=args.term)

    # --streams S > 1: the per-rank batch is cut into S sub-batches, each with its own model object (own workspaces,
    # weights re-packed per plan) on its own HIP stream inside the captured step; the low-resolution layers of one
    # sub-batch (small grids, latency-bound) and its memory-bound splat/stem kernels then overlap the matrix-bound
    # high-resolution layers of another.
    S = max(1, min(args.streams, B)) if args.streams > 0 else max(1, B // 16)
    if S > 1:
        if B % S:
            raise SystemExit('--batch must be a multiple of --streams')
        models = [model] + [build_model(model_params()) for _ in range(S - 1)]
        for m in models[1:]:
            m.load_state_dict(sd)
            m.eval()
        sub = B // S
        subs = [{k: v[i * sub:(i + 1) * sub].contiguous() for k, v in batch.items()} for i in range(S)]
        side = [torch.cuda.Stream() for _ in range(S - 1)]

    def step():
        if S == 1:
            return model.predict(batch, None)
        cur = torch.cuda.current_stream()
        outs = [None] * S
        for i in range(1, S):
            side[i - 1].wait_stream(cur)
            with torch.cuda.stream(side[i - 1]):
                outs[i] = models[i].predict(subs[i], None)
        outs[0] = models[0].predict(subs[0], None)
        for i in range(1, S):
            cur.wait_stream(side[i - 1])
        return outs

    out = step()          # builds the plan, sizes the workspaces
    torch.cuda.synchronize()
    use_graph = not args.no_graph
    graph = None
    if use_graph:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = step()
        run = graph.replay
    else:
        run = step

    for _ in range(args.warmup):
        run()
    if pfdist.is_dist():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if pfdist.is_dist():
        torch.distributed.barrier()
    elapsed = pfdist.max_over_ranks(time.perf_counter() - t0, dev)
    frames = world * B * args.steps
    value = frames / elapsed

    if S > 1:   # the sub-batch outputs are concatenated outside the timed region
        out = {k: torch.cat([o[k] for o in out]) for k in out[0]}
    # ---- sharded metric exchange: PQ accumulators of this rank's forecasts vs a synthetic ground truth
    gt = torch.from_numpy(synth.ID2TRAINID).to(dev)[batch['seg'][:, T - 1].long()].long()
    acc = pfpq.pq_accumulate(out['seg'].long(), gt, 11)
    allacc = pfdist.gather_accumulators(acc)
    pq_synth = pfpq.pq_from_acc(allacc.sum(0))['pq']

    # ---- per-kernel timing pass (eager, hipEvents on the launch stream) -> roofline of the dominant kernel
    roofline = None
    if rank == 0:
        # one sub-batch alone on the launch stream: kernels of concurrent streams share the chip, which would inflate
        # the per-launch durations the roofline fraction is computed from
        prof_step = (lambda: models[0].predict(subs[0], None)) if S > 1 else step
        pflib.profile(True)
        for _ in range(args.profile_steps):
            prof_step()
        torch.cuda.synchronize()
        recs = pflib.profile_results()
        pflib.profile(False)
        tot = sum(r['ms'] for r in recs)
        dom = max(recs, key=lambda r: r['ms'])
        per_launch_ms = dom['ms'] / dom['launches']
        gbs = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9
        if dom['flops'] > 0:
            achieved = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
            if 'conv_split' in dom['label']:
                # every algorithmic fp32 MAC is 3 bf16 MFMA MACs (hi*hi + hi*mid + mid*hi): the matrix ceiling of this
                # scheme, in algorithmic flops, is the dense bf16 peak / 3
                peak, note = PEAK_BF16_MFMA_TFLOPS / 3.0, 'dense bf16 MFMA peak 2500 TFLOP/s / 3 products per fp32 MAC'
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
        roofline.update({'traffic': pmc_traffic(dom['label']), 'kernel': dom['label'], 'launches_per_step': dom['launches'] // args.profile_steps,
                         'avg_launch_us': per_launch_ms * 1e3, 'share_of_step': dom['ms'] / tot,
                         'all_conv_tflops': sum(r['flops'] for r in conv) / (conv_ms * 1e-3) / 1e12 if conv_ms else None,
                         'kernel_ms_per_step': tot / args.profile_steps,
                         'measured_on': 'one sub-batch of %d frames alone on the launch stream (eager, hipEvents)' % (B // S)})
        if os.environ.get('PF_BENCH_KERNELS'):
            for r in sorted(recs, key=lambda r: -r['ms']):
                print('# %-70s n=%3d %8.3f ms  %7.2f TF/s %8.1f GB/s' % (
                    r['label'][:70], r['launches'] // args.profile_steps, r['ms'] / args.profile_steps,
                    r['flops'] / max(r['ms'], 1e-9) / 1e9, r['bytes'] / max(r['ms'], 1e-9) / 1e6), file=sys.stderr)

    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        fps, secs, ref, n_done = cpu_baseline(sd, args.cpu_frames, args.term)
        cpu = {'value': fps, 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
               'sample': '%d forecast frames @%dx%d (3 C-oracle splats on 1 thread + torch-CPU HarDNet on %d threads each), %.1f s'
                         % (n_done, H, W, torch.get_num_threads(), secs)}
        # full-size parity of the LAST cpu frame (seed cpu_frames-1) against the HIP path
        chk = make_batch(1, seed0=n_done - 1, device=dev, term=args.term)
        pl = model_params()
        pl['model']['return_logits'] = True
        lmodel = build_model(pl)
        lmodel.load_state_dict(sd)
        lmodel.eval()
        res = lmodel.predict(chk, None)
        got = res['seg'].long().cpu()
        agree = float((got == ref['seg']).float().mean())
        pq_ref = pfpq.pq_from_acc(pfpq.pq_accumulate(got, ref['seg'], 11))['pq']
        dlogit = float((res['orig_size_logits'].cpu() - ref['orig_size_logits']).abs().max())
        parity = {'argmax_agreement_vs_oracle': agree, 'pq_vs_oracle_as_gt': pq_ref, 'max_abs_dlogit_vs_oracle': dlogit,
                  'logit_tolerance': 1e-3}

    if rank == 0:
        line = {'metric': 'forecast frames/sec @1024x2048, 3-in->dt=%d bg' % (3 if args.term == 'short' else 9), 'value': value, 'unit': 'frames/s',
                'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                'dtype': 'f32' if args.fp32_mfma_only else 'f32 (storage, accumulation, 1x1/strided/low-res convs: fp32 MFMA; tuned 3x3 layers: '
                         'operands split into bf16 hi+mid, 3 products on the bf16 MFMA, fp32 accumulate)', 'data': 'synthetic',
                'config': {'workload': ('configs[1]: bg short-term forecast, 3 frames in, dt=3' if args.term == 'short' else
                                        'configs[2]: bg mid-term forecast, 3 frames in, dt=9, predicted-odometry ego chain') +
                                       ', 1024x2048, random-init calibrated weights; step = 3 warp/splats + HarDNet + upsample/argmax',
                           'frames_per_gpu_per_step': B, 'streams': S, 'launch': 'hipGraph replay' if use_graph else 'eager',
                           'sharding': 'batch over %d rank(s), no data-path collective' % world},
                'roofline': roofline, 'cpu_baseline': cpu, 'parity': parity,
                'pq_gather_check': {'pq_vs_last_input_labels': pq_synth,
                                    'note': 'random-init weights: value is meaningless, it exercises the sharded PQ all-gather'}}
        print(json.dumps(line))
    if pfdist.is_dist():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
