#!/usr/bin/env python
"""Timing of one bg training step (scope row f4) on the configuration of configs/bg/bg_train.yaml: batch 8, 800x800 crops,
3 input frames; per-kernel hipEvent times of pf_train_forward_backward + pf_sgd_step.

    python tools/bench_train.py [--batch 8] [--size 800] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panoptic_forecasting_amd import lib as pflib, synth  # noqa: E402
from panoptic_forecasting_amd.bg_train import BGTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--size', type=int, default=800)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--graph', action='store_true', help='training.use_hip_graph: replay a captured step (one stream) instead of eager launches')
    ap.add_argument('--no-side-stream', action='store_true', help='training.weight_gradient_stream = False: weight gradients on the caller\'s stream')
    ap.add_argument('--autotune', action='store_true', help='training.autotune: measured conv shapes')
    ap.add_argument('--lib', default='', help='another build of libpfhip.so (A/B runs)')
    ap.add_argument('--out', default='', help='also write the JSON line here')
    ap.add_argument('--layers', action='store_true', help='one row per weight-gradient launch (use with --no-side-stream: alone on the chip)')
    a = ap.parse_args()
    if a.lib:
        pflib.LIB_PATH = os.path.abspath(a.lib)
    for kv in filter(None, os.environ.get('PF_OPTS', '').split(',')):    # A/B runs: PF_OPTS=train_blocked_sum=0,...
        k, v = kv.split('=')
        pflib.check(pflib.load().pf_set_option(k.encode(), int(v)), 'pf_set_option')
    params = {'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])]},
              'model': {'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True},
              'training': {'lr': 2e-3, 'mom': 0.9, 'wd': 1e-4, 'clip_grad_norm': 5.0, 'use_hip_graph': a.graph, 'weight_gradient_stream': not (a.graph or a.no_side_stream), 'autotune': a.autotune}}
    tr = BGTrainer(params)
    tr.load_state_dict(synth.make_state_dict(seed=1234))
    inp = {k: v.cuda() for k, v in synth.make_bg_inputs(b=a.batch, h=a.size, w=a.size, seed=1).items()}
    inp['seg'] = inp['seg'].to(torch.uint8)
    lab = {'seg': torch.randint(0, 11, (a.batch, a.size, a.size), dtype=torch.uint8, device='cuda')}
    for _ in range(3):          # the first step of a configuration runs eagerly, the second captures the hipGraph
        tr.train_step(inp, lab)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        out = tr.train_step(inp, lab)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / a.steps * 1e3
    graph = tr.use_graph
    tr.use_graph = False        # the per-kernel records need eager launches (hipEvents around every kernel)
    pflib.profile(True)
    tr.train_step(inp, lab)
    torch.cuda.synchronize()
    recs = pflib.profile_results()
    pflib.profile(False)
    tr.use_graph = graph
    if a.layers:
        tot = 0.0
        for r in recs:
            if '@' in r['label'] and 'wgrad' in r['label'] and 'reduce' not in r['label']:
                tot += r['ms']
                print('%-60s %8.1f us %7.1f TF/s %7.0f GB/s' % (r['label'][-60:], r['ms'] * 1e3, r['flops'] / max(r['ms'], 1e-9) / 1e9, r['bytes'] / max(r['ms'], 1e-9) / 1e6))
        print('weight-gradient kernels: %.3f ms' % tot)
        recs = [dict(r, label=r['label'].split(' @')[0]) for r in recs]
        agg = {}
        for r in recs:
            g = agg.setdefault(r['label'], dict(r, ms=0.0, launches=0, flops=0.0, bytes=0.0))
            for k in ('ms', 'launches', 'flops', 'bytes'):
                g[k] += r[k]
        recs = list(agg.values())
    top = sorted(recs, key=lambda r: -r['ms'])
    line = {'ms_per_step': ms, 'samples_per_s': a.batch / ms * 1e3, 'batch': a.batch, 'size': a.size, 'loss': float(out['loss']),
            'launch': 'hipGraph replay of forward + loss + backward, eager SGD step' if graph else ('eager, weight gradients on their own stream' if tr.side_stream else 'eager, one stream'), 'steps': a.steps,
            'weight_gradient_stream': tr.side_stream, 'autotune': tr.autotune, 'workspace_GB': tr._ws.numel() / 1e9, 'kernel_ms_sum': sum(r['ms'] for r in recs),
            'profiled_kernels': {r['label'][:70]: {'ms': round(r['ms'], 3), 'launches': r['launches'],
                                                   'TFLOPs': round(r['flops'] / max(r['ms'], 1e-9) / 1e9, 1),
                                                   'GBps': round(r['bytes'] / max(r['ms'], 1e-9) / 1e6, 0)} for r in top}}
    print(json.dumps(line))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
        with open(a.out, 'w') as f:
            json.dump(line, f, indent=1)


if __name__ == '__main__':
    main()
