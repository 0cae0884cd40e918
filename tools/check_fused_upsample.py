import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import MiniNet, MiniSpec
from panoptic_forecasting_amd import hardnet_arch as arch, lib as pflib
L = pflib.load()
import os
L.pf_set_option(b'fuse_upsample', int(os.environ.get('FUSE', '1')))
cx, cs, co, h, w = [int(v) for v in sys.argv[1:6]]
if len(sys.argv) > 6:
    L.pf_debug_force_conv(*[int(v) for v in sys.argv[6:10]])
g = torch.Generator().manual_seed(0)
x = torch.randn(1, cx + cs, h, w, generator=g)
spec = MiniSpec(cx + cs)
# low-res tensor = pooled slice of the input through an identity-ish 1x1 conv + pool
lo = spec.conv('lo', [arch.Src(0, 0, cx)], cx, 1, relu=False)
p = spec.pool('p', lo)
sk = spec.conv('sk', [arch.Src(0, cx, cs)], cs, 1, relu=False)
up = spec.upsample('up', p, sk)
spec.conv('c', [arch.Src(up, 0, cx), arch.Src(sk, 0, cs)], co, 1)
P = {'lo': (torch.eye(cx).view(cx, cx, 1, 1).contiguous(), torch.zeros(cx)),
     'sk': (torch.eye(cs).view(cs, cs, 1, 1).contiguous(), torch.zeros(cs)),
     'c': (torch.randn(co, cx + cs, 1, 1, generator=g) / (cx + cs) ** 0.5, torch.randn(co, generator=g))}
net = MiniNet(spec, P).run(x.cuda())
torch.cuda.synchronize()
rp = F.avg_pool2d(x[:, :cx], 2, 2)
ru = F.interpolate(rp, size=(h, w), mode='bilinear', align_corners=True)
ref = F.relu(F.conv2d(torch.cat([ru, x[:, cx:]], 1), *P['c']))
import ctypes
buf = (ctypes.c_longlong * 64)()
if L.pf_debug_probe_read(buf) == 0:
    t = list(buf)[:5]
    print('epilogue per M-tile (10 ns):', [list(buf)[5 + m] - (list(buf)[4 + m] if m else t[3]) for m in range(4)])
    print('last probed conv_dma workgroup, 10 ns ticks: main loop %d, residual staging %d, k-reduce %d, epilogue %d' % (t[1]-t[0], t[2]-t[1], t[3]-t[2], t[4]-t[3]))
print('max err', (net.tensor('c').cpu() - ref).abs().max().item())
