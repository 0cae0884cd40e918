#!/usr/bin/env python
"""Which kernels wait for their own stores?  On gfx950 stores count in vmcnt, so an `s_waitcnt vmcnt(N)` after the first
global/buffer store of a kernel makes the wave sit through store round trips (1-2 us under load): typically hipcc
protecting a register it cannot prove loaded (a load issued before a loop that contains inline-asm waits), re-inserted in
every iteration of an epilogue loop.  Static count per kernel, from the compiler's own assembly (no GPU needed).

    python tools/asm_store_waits.py conv_s4.hip [conv_dma.hip ...]
"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'panoptic-forecasting_amd', 'csrc')


def main():
    for src in sys.argv[1:]:
        cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only', '-o', '-']
        if src.startswith(('warp_splat', 'panoptic_merge', 'hop_kernels')):
            cmd.append('-ffp-contract=off')
        asm = subprocess.run(cmd + [src], cwd=CSRC, capture_output=True, text=True).stdout
        name, stores, waits, rows = None, 0, [], []
        for line in asm.split('\n'):
            m = re.match(r'^(_Z\w+):', line)
            if m:
                name, stores, waits = m.group(1), 0, []
                continue
            if name is None:
                continue
            if re.search(r"\b(global|buffer|flat)_store", line):
                stores += 1
            m = re.search(r's_waitcnt.*vmcnt\((\d+)\)', line)
            if m and stores:
                waits.append((int(m.group(1)), stores))
            if 's_endpgm' in line:
                if waits:
                    rows.append((name, stores, waits))
                name = None
        names = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
        for (n, stores, waits), dn in zip(rows, names):
            print('%-34s %-80s stores %3d  waits after a store: %s' % (src, dn[:80], stores, ' '.join('vmcnt(%d)@%d' % w for w in waits[:12])))


if __name__ == '__main__':
    main()
