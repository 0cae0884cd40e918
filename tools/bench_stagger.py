#!/usr/bin/env python
"""A/B of the step schedule (bench.py Workload): sub-batches started together vs software-pipelined (--stagger).
    python tools/bench_stagger.py "32,2,0 32,2,1 32,4,1 48,3,1 64,4,1"      # frames,streams,stagger triples
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main(spec):
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    sd = bench.calibrated_state_dict()
    for rep in range(2):
        for item in spec.split():
            B, S, stg = (int(x) for x in item.split(','))
            wl = bench.Workload(sd, B, S, dev, seed0=0, term='short', use_graph=True, stagger=bool(stg))
            el = wl.timed(20, 5, dev)
            print('B=%d S=%d stagger=%d  %.1f frames/s  %.3f ms/step' % (B, S, stg, B * 20 / el, el / 20 * 1e3), flush=True)
            del wl
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '32,2,0 32,2,1 32,4,0 32,4,1 48,3,1 64,4,1')
