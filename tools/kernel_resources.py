#!/usr/bin/env python
"""Register / LDS / scratch usage of every kernel of one csrc file, from the compiler's own remarks (no GPU needed).

    python tools/kernel_resources.py conv_s4.hip [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'panoptic-forecasting_amd', 'csrc')


def demangle(names):
    try:
        out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
        return out[:len(names)]
    except OSError:
        return names


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Rpass-analysis=kernel-resource-usage']
    if src.startswith(('warp_splat', 'panoptic_merge', 'hop_kernels')):
        cmd.append('-ffp-contract=off')
    err = subprocess.run(cmd + extra + ['-c', src, '-o', '/dev/null'], cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split('\n'):
        m = re.search(r'remark: (?:Function Name|\s*)(.*?): *(\S+) \[-Rpass', line)
        m = re.search(r'remark:\s+(.*?):\s+(\S+)\s+\[-Rpass-analysis', line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == 'Function Name':
            cur = {'name': val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    names = demangle([r['name'] for r in rows])
    for r, n in zip(rows, names):
        print('%-72s vgpr %3s agpr %3s sgpr %3s spill_v %3s scratch %5s occ %s lds %s' % (
            n[:72], r.get('VGPRs'), r.get('AGPRs'), r.get('SGPRs'), r.get('VGPRs Spill'), r.get('ScratchSize [bytes/lane]'),
            r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]')))


if __name__ == '__main__':
    main()
