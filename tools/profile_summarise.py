#!/usr/bin/env python
"""Turn gpurun_out/<tag>/ (tools/profile_gpu.sh) into the committed summaries under profiles/.

    python tools/profile_summarise.py <tag> <round-prefix> [--no-latest]     e.g.  r1c r01_c

The hash of the kernel sources the numbers were measured on comes from gpurun_out/<tag>/source_sha.txt (written on the GPU
box by tools/profile_gpu.sh); without that file the current tree is hashed.  --no-latest: do not refresh pmc_latest.json
(side legs such as --fp32-mfma-only).

Writes profiles/<prefix>_kernel_stats.txt (per-kernel calls/avg ns from the kernel trace) and
profiles/<prefix>_pmc.json (per-kernel HBM traffic per launch from FETCH_SIZE / WRITE_SIZE, raw and
calibrated against the known-byte copy kernel, plus MFMA busy fraction).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAL_BYTES = 4 * 64 * 1024 * 1024


def find(d, pat):
    r = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return r[0] if r else None


def kernel_stats(trace_dir):
    path = find(trace_dir, '*kernel_trace.csv')
    agg = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            agg[row['Kernel_Name']].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
    return {k: (len(v), sum(v), sum(v) / len(v), min(v), max(v)) for k, v in agg.items()}


def counters(d):
    path = find(d, '*counter_collection.csv')
    agg = defaultdict(lambda: defaultdict(list))
    if not path:
        return agg
    with open(path) as f:
        for row in csv.DictReader(f):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
    return agg


def main(tag, prefix, latest=True):
    src = os.path.join(ROOT, 'gpurun_out', tag)
    dst = os.path.join(ROOT, 'profiles')
    ks = kernel_stats(os.path.join(src, 'trace'))
    total = sum(v[1] for v in ks.values())
    bench_line = ''
    try:
        bench_line = [l for l in open(os.path.join(src, 'bench_trace.log')) if l.startswith('{')][-1].strip()
    except Exception:
        pass
    with open(os.path.join(dst, prefix + '_kernel_stats.txt'), 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats -- %s\n' % (open(os.path.join(src, 'bench_cmd.txt')).read().strip() if os.path.exists(os.path.join(src, 'bench_cmd.txt')) else 'python bench.py --batch 16 --streams 1 --steps 10 --warmup 3 --replays 1 --no-cpu-baseline --no-legs --no-graph --profile-steps 1'))
        f.write('# bench line of that run: %s\n' % bench_line)
        f.write('%-90s %7s %12s %11s %11s %11s %6s\n' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', '%'))
        for k, (c, tot, avg, mn, mx) in sorted(ks.items(), key=lambda kv: -kv[1][1]):
            f.write('%-90s %7d %12d %11.0f %11d %11d %6.2f\n' % (k[:90], c, tot, avg, mn, mx, 100.0 * tot / total))
    sys.path.insert(0, ROOT)
    import bench as _bench
    sha_file = os.path.join(src, 'source_sha.txt')
    sha = open(sha_file).read().strip() if os.path.exists(sha_file) else _bench.source_sha()
    out = {'source_sha': sha,
           'note': 'per-launch averages; FETCH_SIZE/WRITE_SIZE are reported by rocprofv3 in KiB-like units and are '
                   'calibrated here on a 256 MiB copy kernel run under the same counter (factor = known bytes / raw)',
           'kernels': {}}
    cal = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        cc = counters(os.path.join(src, 'cal_' + c))
        vals = [v for k, d in cc.items() if 'elementwise' in k for v in d.get(c, [])]
        vals = [v for v in vals if v > 0]
        if vals:
            big = max(vals)
            vals = [v for v in vals if v > 0.5 * big]       # the five 256 MiB launches
            cal[c] = {'raw_per_launch': sum(vals) / len(vals), 'factor_bytes_per_unit': CAL_BYTES / (sum(vals) / len(vals))}
    out['calibration'] = cal
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        cc = counters(os.path.join(src, 'pmc_' + c))
        for k, d in cc.items():
            v = d.get(c, [])
            if not v:
                continue
            e = out['kernels'].setdefault(k, {})
            raw = sum(v) / len(v)
            e[c + '_raw'] = raw
            if c in cal:
                e[c + '_bytes'] = raw * cal[c]['factor_bytes_per_unit']
    for sub in ('pmc_mfma', 'pmc_sq'):
        cc = counters(os.path.join(src, sub))
        for k, d in cc.items():
            e = out['kernels'].setdefault(k, {})
            for name, v in d.items():
                e[name] = sum(v) / len(v)
    for k, d in list(out['kernels'].items()):
        e = d
        if e.get('SQ_BUSY_CYCLES') and 'SQ_VALU_MFMA_BUSY_CYCLES' in e:
            e['mfma_busy_over_sq_busy'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / e['SQ_BUSY_CYCLES']
    for k, e in out['kernels'].items():
        if k in ks:
            e['avg_ns'] = ks[k][2]
            e['calls'] = ks[k][0]
        if 'FETCH_SIZE_bytes' in e and 'WRITE_SIZE_bytes' in e:
            e['hbm_bytes_per_launch'] = e['FETCH_SIZE_bytes'] + e['WRITE_SIZE_bytes']
    for k, e in out['kernels'].items():
        # MFMA pipe utilisation: busy cycles summed over the 1024 SIMDs / (per-XCD active cycles summed over 8 XCDs / 8)
        if e.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in e:
            e['mfma_util'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * e['GRBM_GUI_ACTIVE'] / 8.0)
    for name in ((prefix + '_pmc.json', 'pmc_latest.json') if latest else (prefix + '_pmc.json',)):
        with open(os.path.join(dst, name), 'w') as f:
            json.dump(out, f, indent=1, sort_keys=True)
    print('wrote', prefix + '_kernel_stats.txt', prefix + '_pmc.json')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], latest='--no-latest' not in sys.argv[3:])
