import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from test_gpu_conv import _pair_net, _pair_ref
from helpers import MiniNet
from panoptic_forecasting_amd import lib as pflib
L = pflib.load()
pflib.check(L.pf_debug_force_conv(5, 2, 0, 0), 'force')
nfail = 0
for (h, w, b, c_odd, c_even) in [(40, 100, 3, 10, 18)] * 6:
    g = torch.Generator().manual_seed(h * 7 + w + c_even)
    x = torch.randn(b, 12, h, w, generator=g) * torch.exp(0.5 * torch.randn(b, 12, 1, 1, generator=g))
    spec, P = _pair_net(g, 12, c_odd, c_even, False)
    base = MiniNet(spec, P).set_option('fuse_pairs', 0).run(x.cuda())
    ref = {k: base.tensor(k).cpu() for k in ('L2', 'out')}
    for trial in range(3):
        net = MiniNet(spec, P).set_option('fuse_pairs', 2).run(x.cuda())
        got = net.tensor('out').cpu()[:, :10]       # L1 = P of the first pair
        d = (got - ref['out'][:, :10]).abs().amax(1)    # [b, h, w]
        bad = d > 1e-3
        if bad.any():
            nfail += 1
            for bi in range(b):
                if not bad[bi].any():
                    continue
                print('batch', bi, 'bad pixels', int(bad[bi].sum()))
                for ty in range(0, h, 8):
                    for tx in range(0, w, 32):
                        t = bad[bi, ty:ty + 8, tx:tx + 32]
                        if t.any():
                            rows = [''.join('X' if v else '.' for v in r.tolist()) for r in t]
                            print(' tile y0=%d x0=%d' % (ty, tx))
                            for r in rows:
                                print('   ' + r)
                            yy, xx = t.nonzero()[0].tolist()
                            print('   sample got', got[bi, :4, ty + yy, tx + xx].tolist(), 'ref', ref['out'][bi, :4, ty + yy, tx + xx].tolist())
            net.close()
            break
        net.close()
    if nfail >= 2:
        break
print('failures', nfail)
