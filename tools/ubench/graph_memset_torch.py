#!/usr/bin/env python3
"""The same question as graph_memset.hip inside a torch process (torch.cuda.graph capture + replay, torch's bundled HIP runtime):
does a captured hipMemsetAsync node run on every replay?  buf is cleared by the captured memset, then incremented by a torch kernel:
1 after every replay = the memset ran."""
import ctypes
import os
import sys

import torch

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def case(name, nbytes, pre, post, side=False):
    buf = torch.zeros(nbytes // 4, dtype=torch.int32, device='cuda')
    other = torch.zeros(1 << 18, dtype=torch.float32, device='cuda')
    st2 = torch.cuda.Stream()

    def body():
        for _ in range(pre):
            other.mul_(1.0001)
        if side:
            st2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st2):
                other.add_(1.0)
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        buf.add_(1)
        for _ in range(post):
            other.mul_(1.0001)
        if side:
            torch.cuda.current_stream().wait_stream(st2)

    body()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    out = []
    for rep in range(5):
        g.replay()
        torch.cuda.synchronize()
        out.append('%d/%d' % (int(buf[0]), int(buf[-1])))
    print('%-60s %s' % (name, '  '.join(out)))


case('chain: memset 4 KB -> add', 4096, 0, 0)
case('chain: 3 k -> memset 1 MB -> add -> 3 k', 1 << 20, 3, 3)
case('chain: 40 k -> memset 64 MB -> add -> 40 k', 64 << 20, 40, 40)
case('chain: 8 k -> memset 1 GB -> add -> 300 k', 1 << 30, 8, 300)
case('fork : 3 k -> memset 1 MB -> add -> 3 k', 1 << 20, 3, 3, side=True)
