#!/usr/bin/env python
"""Level-consistent conv_s4 rows from a tools/tune_s4.py log.

The tuner picks a winner per LAYER, but the packed-pair layout is a property of TENSORS: one fp32-source reader pins every tensor it
shares with its neighbours to fp32, and the whole level falls back (profiles/r05_experiments.md).  This script decides per LEVEL
(encoder / decoder block of one resolution): if the level as a whole is faster with every layer on its best conv_s4 shape than on
the automatic choice, ALL its layers get kind-5 rows (their best shapes), else the log's per-layer rows stand.

    python tools/level_rows.py gpurun_out/r5_tune_s4_b1_pass2.txt 1 [--margin 0.03]   -> rows on stdout for the levels that flip
"""
import collections
import re
import sys


def shape_row(name):
    m = re.match(r'k<(\d+), (\d+), (\d+), (\d+)>', name)
    if m:
        nt, tw, th, ks = map(int, m.groups())
        p1 = {(32, 8, 1): 0, (64, 8, 1): 1, (32, 16, 1): 2, (32, 8, 2): 3, (32, 8, 4): 4}[(tw, th, ks)]
        return nt, p1
    m = re.match(r'1x1_k<(\d+), (\d+)>', name)
    return int(m.group(1)), 0


def main():
    path, batch = sys.argv[1], int(sys.argv[2])
    margin = float(sys.argv[sys.argv.index('--margin') + 1]) if '--margin' in sys.argv else 0.03
    levels = collections.OrderedDict()
    for line in open(path):
        m = re.match(r'^(\d+[ab]?) (\S+) (\d+)->(\d+) (\d+)x(\d+)\s+auto\s+([\d.]+) us (\S+?)[<( ].*?\|\s*(.*)$', line)
        if not m:
            continue
        idx, name, cin, cout, h, w, auto, kern, rest = m.groups()
        cands = re.findall(r'(k<[^>]*>|1x1_k<[^>]*>)\s+([\d.]+)', rest)
        if not cands:
            continue
        best = min(cands, key=lambda c: float(c[1]))
        n = int(re.match(r'\d+', idx).group())
        ks = 1 if ('conv1x1' in name or 'finalConv' in name or re.match(r'base\.(5|8|11|14|17)$', name)) else 3
        key = ('enc' if n < 45 else 'dec', int(h), int(w))
        levels.setdefault(key, []).append(dict(idx=idx, name=name, ks=ks, cin=int(cin), cout=int(cout), h=int(h), w=int(w),
                                               auto=float(auto), kern=kern, best=best[0], best_us=float(best[1])))
    for key, items in levels.items():
        a, s = sum(i['auto'] for i in items), sum(i['best_us'] for i in items)
        mixed = any('s4' not in i['kern'] for i in items)
        if not mixed or s >= (1.0 - margin) * a:
            continue
        print('    // B=%d, %s %dx%d as a whole on conv_s4 (tools/level_rows.py): %.1f -> %.1f us' % (batch, key[0], key[1], key[2], a, s))
        seen = set()
        for i in items:
            k = (i['ks'], i['cin'], i['cout'], i['h'], i['w'])
            if k in seen:
                continue
            seen.add(k)
            nt, p1 = shape_row(i['best'])
            print('    {%d, %d, %d, %d, %d, %d, {5, %d, %d, 0}},   // %s: auto %.1f (%s) -> %.1f us' % (
                i['ks'], i['cin'], i['cout'], i['h'], i['w'], batch, nt, p1, i['name'], i['auto'], i['kern'].replace('conv_', '').replace('_kernel', ''), i['best_us']))


if __name__ == '__main__':
    main()
