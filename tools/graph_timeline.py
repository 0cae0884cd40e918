#!/usr/bin/env python
"""Timeline of the replayed hipGraph from a rocprofv3 kernel trace (csv): how much of a step the chip is busy, how much of
that with two kernels at once, where the idle gaps are and which kernels stretch when they share the chip.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-legs
    python tools/graph_timeline.py gpurun_out/tl/*/t_kernel_trace.csv [steps_to_analyse [sub_batches_per_step [skip_last_sub_batches]]]
The trace ends with the eager per-kernel timing pass of bench.py (--profile-steps sub-batch forwards): skip those.
"""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace('void pf::', '').replace('pf::', '')
    return n[:60]


def main(path, last_steps=4, per_step=2, skip_last=0):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    # a step of the default bench = 2 sub-batches: find the last `last_steps` steps by counting bin_kernel launches (one per sub-batch)
    bins = [i for i, r in enumerate(rows) if 'bin_kernel' in r[2]]
    need = last_steps * per_step
    if len(bins) < need + per_step:
        print('not enough steps in the trace (%d bin_kernel launches)' % len(bins))
        return
    if len(bins) < need + per_step + skip_last:
        print('not enough steps in the trace (%d bin_kernel launches)' % len(bins))
        return
    first = bins[-need - skip_last]
    # the memset before the first bin_kernel belongs to the step too; good enough to start at the bin kernel
    sel = rows[first:bins[-skip_last]] if skip_last else rows[first:]
    # the selection ends with the last head kernel of the replayed steps (what follows is the metric exchange / the memset of
    # the next forward)
    heads = [i for i, r in enumerate(sel) if 'head_' in r[2]]
    if heads:
        sel = sel[:heads[-1] + 1]
    t0, t1 = sel[0][0], max(r[1] for r in sel)
    span = t1 - t0
    ev = []
    for s, e, n in sel:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    conc = defaultdict(int)
    level, last = 0, t0
    for t, d in ev:
        conc[level] += t - last
        last = t
        level += d
    total_k = sum(e - s for s, e, _ in sel)
    print('steps analysed: %d   span %.3f ms per step   sum of kernel durations %.3f ms per step' %
          (last_steps, span / last_steps / 1e6, total_k / last_steps / 1e6))
    for lv in sorted(conc):
        print('  %d kernel(s) running: %6.2f %% of the span' % (lv, 100.0 * conc[lv] / span))
    # idle gaps
    gaps = []
    cur_end, cur_name = sel[0][1], sel[0][2]
    for s, e, n in sel[1:]:
        if s > cur_end:
            gaps.append((s - cur_end, cur_name, n))
        if e > cur_end:
            cur_end, cur_name = e, n
    gaps.sort(reverse=True)
    print('idle gaps: %d, %.1f us per step in total; largest:' % (len(gaps), sum(g[0] for g in gaps) / last_steps / 1e3))
    for g, a, b in gaps[:8]:
        print('  %7.1f us  after %-50s before %s' % (g / 1e3, short(a), short(b)))
    # per-kernel durations in this (concurrent) run
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        agg[n][0] += 1
        agg[n][1] += e - s
    print('per kernel (concurrent run): calls per step, avg us, total us per step')
    for n, (c, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        print('  %-62s %6.1f %9.1f %9.1f' % (short(n), c / last_steps, tot / c / 1e3, tot / last_steps / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], *[int(a) for a in sys.argv[2:5]])
