#!/usr/bin/env python
"""Serial timeline of a one-stream graph replay from a rocprofv3 kernel trace (csv): each launch's duration and the idle
gap between the previous kernel's end and its start, averaged over the replayed steps. The low-batch legs (B = 1, 2) are a
dependent chain of ~85 launches, so step time = sum(durations) + sum(gaps).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlb1 -o t -- \
        python bench.py --batch 1 --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-legs --profile-steps 1
    python tools/serial_timeline.py gpurun_out/tlb1/*/t_kernel_trace.csv [steps]
"""
import csv
import sys
from collections import defaultdict


def short(n):
    return n.replace('void pf::', '').replace('pf::', '').replace('(pf::ConvArgs)', '')[:58]


def main(path, steps=20, skip_tail=2):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    bins = [i for i, r in enumerate(rows) if 'bin_kernel' in r[2]]
    # the trace ends with bench.py's eager per-kernel pass: drop the last `skip_tail` forwards
    bins = bins[:len(bins) - skip_tail]
    if len(bins) < steps + 1:
        print('only %d forwards in the trace' % len(bins))
        return
    per = []
    for a, b in zip(bins[-steps - 1:-1], bins[-steps:]):
        per.append(rows[a:b])
    n = min(len(p) for p in per)
    if any(len(p) != n for p in per):
        print('launch counts differ between steps: %s' % sorted(set(len(p) for p in per)))
    dur = defaultdict(float)
    gap = defaultdict(float)
    name = {}
    for p in per:
        prev_end = None
        for i, (s, e, k) in enumerate(p[:n]):
            dur[i] += (e - s) / 1e3
            if prev_end is not None:
                gap[i] += (s - prev_end) / 1e3
            prev_end = max(prev_end or e, e)
            name[i] = k
    span = sum((p[n - 1][1] - p[0][0]) / 1e3 for p in per) / len(per)
    step = sum((b[0][0] - a[0][0]) / 1e3 for a, b in zip(per[:-1], per[1:])) / max(len(per) - 1, 1)
    td = sum(dur.values()) / len(per)
    tg = sum(gap.values()) / len(per)
    print('%d launches per forward; first start -> last end %.1f us; start -> next start %.1f us' % (n, span, step))
    print('sum of durations %.1f us, sum of gaps %.1f us (mean gap %.2f us)' % (td, tg, tg / max(n - 1, 1)))
    for i in range(n):
        print('%3d %-58s %7.2f us  gap before %6.2f us' % (i, short(name[i]), dur[i] / len(per), gap[i] / len(per)))


if __name__ == '__main__':
    main(sys.argv[1], *[int(v) for v in sys.argv[2:]])
