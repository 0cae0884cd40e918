#!/usr/bin/env python3
"""Per-phase / per-queue summary of one training step from a rocprofv3 --kernel-trace CSV of tools/bench_train.py.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/<tag> -o t -- python tools/bench_train.py --steps 3
    python tools/train_timeline.py gpurun_out/<tag>/t_kernel_trace.csv
"""
import collections
import csv
import statistics
import sys


def main(path, top=14):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    rows.sort(key=lambda r: r['s'])
    starts = [i for i, r in enumerate(rows) if 'onehot_dense' in r['Kernel_Name']]
    i0, i1 = starts[-3], starts[-2]          # a timed step (the last one is the per-kernel profiled step)
    step = rows[i0:i1]
    t0, t1 = step[0]['s'], rows[i1]['s']
    print('step %.3f ms, %d kernels' % ((t1 - t0) / 1e6, len(step)))
    q = collections.Counter(r['Queue_Id'] for r in step)
    mainq = max(q, key=q.get)
    for qid in q:
        ks = [r for r in step if r['Queue_Id'] == qid]
        print('queue %s: %d kernels, busy %.3f ms, first start %.3f, last end %.3f' % (
            qid, len(ks), sum(r['e'] - r['s'] for r in ks) / 1e6, (ks[0]['s'] - t0) / 1e6, (ks[-1]['e'] - t0) / 1e6))
    ks = [r for r in step if r['Queue_Id'] == mainq]
    gaps = [ks[i + 1]['s'] - ks[i]['e'] for i in range(len(ks) - 1)]
    print('main queue gaps: %.3f ms in total, median %.2f us' % (sum(g for g in gaps if g > 0) / 1e6, statistics.median(gaps) / 1e3))
    ice = [i for i, r in enumerate(ks) if 'ce_fwd_bwd' in r['Kernel_Name']][0]
    print('forward : span %.3f ms, busy %.3f, %d kernels' % ((ks[ice]['e'] - t0) / 1e6, sum(r['e'] - r['s'] for r in ks[:ice + 1]) / 1e6, ice + 1))
    print('backward: span %.3f ms, busy %.3f, %d kernels' % ((ks[-1]['e'] - ks[ice]['e']) / 1e6, sum(r['e'] - r['s'] for r in ks[ice + 1:]) / 1e6, len(ks) - ice - 1))

    def agg(lst):
        c = collections.defaultdict(lambda: [0, 0])
        for r in lst:
            n = r['Kernel_Name'].split('(')[0][-58:]
            c[n][0] += r['e'] - r['s']
            c[n][1] += 1
        return sorted(c.items(), key=lambda kv: -kv[1][0])
    for title, lst in (('forward', ks[:ice + 1]), ('backward, caller\'s stream', ks[ice + 1:]), ('weight-gradient stream', [r for r in step if r['Queue_Id'] != mainq])):
        print('--- ' + title)
        for n, (t, k) in agg(lst)[:top]:
            print('  %-60s %7.3f ms %4d' % (n, t / 1e6, k))


if __name__ == '__main__':
    main(sys.argv[1])
