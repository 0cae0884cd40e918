"""HIP MFMA convolution / pool / upsample kernels vs torch fp32 (F.conv2d etc. on the CPU), through the C ABI.

Tolerance: both sides are exact-fp32 dot products that differ only in summation order; for K = Cin*k*k
terms of O(1) magnitude the difference is bounded by ~K * 2^-24 * sum|a*b|.  We assert
max|err| <= 2e-5 * (1 + max|ref|) which holds with margin for K <= 4000.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _tol(ref):
    return 2e-5 * (1.0 + ref.abs().max().item())


CASES = [
    # cin, cout, k, stride, h, w, b
    (48, 10, 3, 1, 64, 128, 1), (58, 18, 3, 1, 32, 96, 2), (76, 28, 3, 1, 40, 72, 1), (32, 48, 3, 1, 64, 64, 1),
    (24, 32, 3, 2, 64, 128, 1), (36, 16, 3, 2, 50, 70, 2), (3, 5, 3, 2, 33, 47, 1),
    (48, 64, 1, 1, 64, 128, 1), (126, 63, 1, 1, 32, 64, 2), (534, 267, 1, 1, 8, 16, 1), (286, 320, 1, 1, 4, 8, 1),
    (402, 158, 3, 1, 4, 8, 1), (196, 88, 3, 1, 16, 32, 1), (91, 28, 3, 1, 30, 50, 1), (17, 70, 3, 1, 9, 13, 3),
    (5, 3, 1, 1, 7, 5, 1), (160, 24, 3, 1, 16, 24, 1),
]


@pytest.mark.parametrize('cin,cout,k,stride,h,w,b', CASES)
@pytest.mark.parametrize('relu', [True, False])
def test_single_conv(cin, cout, k, stride, h, w, b, relu):
    from helpers import MiniNet, MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    g = torch.Generator().manual_seed(cin * 131 + cout)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    spec = MiniSpec(cin)
    spec.conv('c', [arch.Src(0, 0, cin)], cout, k, stride, relu=relu)
    net = MiniNet(spec, {'c': (wt, bias)}).run(x.cuda())
    out = net.tensor('c').cpu()
    ref = F.conv2d(x, wt, bias, stride=stride, padding=k // 2)
    if relu:
        ref = F.relu(ref)
    assert out.shape == ref.shape
    err = (out - ref).abs().max().item()
    assert err <= _tol(ref), (err, _tol(ref))
    net.close()


def test_multi_source_and_channel_slots():
    """HarDBlock-style wiring: a conv reading three channel ranges (one a slice of a wider tensor) and
    writing into a channel slot of a wider tensor; the untouched channels must stay untouched."""
    from helpers import MiniNet, MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    g = torch.Generator().manual_seed(5)
    b, h, w = 2, 24, 40
    x = torch.randn(b, 20, h, w, generator=g)
    spec = MiniSpec(20)
    wide = spec.tensor('wide', 50)                               # slots: [0:18] L1, [18:50] L2
    spec.conv('L1', [arch.Src(0, 0, 20)], 18, 3, dst=wide, dst_choff=0)
    spec.conv('L2', [arch.Src(wide, 0, 18), arch.Src(0, 0, 20)], 32, 3, dst=wide, dst_choff=18)
    spec.conv('L3', [arch.Src(wide, 18, 32), arch.Src(wide, 0, 18), arch.Src(0, 4, 9)], 21, 1)
    P = {}
    for name, cin, cout, k in [('L1', 20, 18, 3), ('L2', 38, 32, 3), ('L3', 59, 21, 1)]:
        P[name] = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, torch.randn(cout, generator=g))
    net = MiniNet(spec, P).run(x.cuda())
    l1 = F.relu(F.conv2d(x, *P['L1'], padding=1))
    l2 = F.relu(F.conv2d(torch.cat([l1, x], 1), *P['L2'], padding=1))
    l3 = F.relu(F.conv2d(torch.cat([l2, l1, x[:, 4:13]], 1), *P['L3']))
    wide_out = net.tensor('wide').cpu()
    assert (wide_out[:, :18] - l1).abs().max() <= _tol(l1)
    assert (wide_out[:, 18:] - l2).abs().max() <= _tol(l2)
    assert (net.tensor('L3').cpu() - l3).abs().max() <= _tol(l3)
    net.close()


@pytest.mark.parametrize('h,w', [(16, 32), (17, 23), (6, 10)])
def test_pool_and_upsample(h, w):
    from helpers import MiniNet, MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    g = torch.Generator().manual_seed(h)
    x = torch.randn(2, 7, h, w, generator=g)
    spec = MiniSpec(7)
    eye = torch.eye(7).view(7, 7, 1, 1).contiguous()
    t = spec.conv('id', [arch.Src(0, 0, 7)], 7, 1, relu=False)
    p = spec.pool('pool', t)
    spec.upsample('up', p, t)
    net = MiniNet(spec, {'id': (eye, torch.zeros(7))}).run(x.cuda())
    pooled = F.avg_pool2d(x, 2, 2)
    up = F.interpolate(pooled, size=(h, w), mode='bilinear', align_corners=True)
    assert (net.tensor('pool').cpu() - pooled).abs().max() <= 1e-6
    assert (net.tensor('up').cpu() - up).abs().max() <= 1e-5
    net.close()


@pytest.fixture
def force_conv():
    from panoptic_forecasting_amd import lib as pflib
    L = pflib.load()
    yield lambda kind, p0, p1, p2: pflib.check(L.pf_debug_force_conv(kind, p0, p1, p2), 'pf_debug_force_conv')
    L.pf_debug_force_conv(0, 0, 0, 0)


FORCED = [(1, wm, nt, 0) for wm in (4, 2, 1) for nt in (1, 2, 3, 4) if not (wm == 1 and nt > 2)] + \
         [(2, mh, nt, wk) for mh in (1, 2, 4) for nt in (1, 2) for wk in (2, 4, 8, 16)]


@pytest.mark.parametrize('force', FORCED, ids=lambda f: 'k%d_%d_%d_%d' % f)
def test_every_kernel_shape(force, force_conv):
    """Each template instantiation of conv_dma / conv_wave, forced through pf_debug_force_conv, on a
    HarDBlock-style three-source 3x3 conv, a wide 1x1 conv and a tiny image (edge tiles, K tail)."""
    from helpers import MiniNet, MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    g = torch.Generator().manual_seed(11)
    b, h, w = 2, 20, 40
    x = torch.randn(b, 45, h, w, generator=g)
    spec = MiniSpec(45)
    spec.conv('A', [arch.Src(0, 30, 10), arch.Src(0, 0, 18), arch.Src(0, 2, 43)], 37, 3)     # cin 71: K tail, 3 cout tiles
    spec.conv('B', [arch.Src(0, 0, 45)], 70, 1, relu=False)
    spec.conv('C', [arch.Src(0, 5, 9)], 10, 3)
    P = {}
    for name, cin, cout, k in [('A', 71, 37, 3), ('B', 45, 70, 1), ('C', 9, 10, 3)]:
        P[name] = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, torch.randn(cout, generator=g))
    force_conv(*force)
    net = MiniNet(spec, P).run(x.cuda())
    ra = F.relu(F.conv2d(torch.cat([x[:, 30:40], x[:, 0:18], x[:, 2:45]], 1), *P['A'], padding=1))
    rb = F.conv2d(x, *P['B'])
    rc = F.relu(F.conv2d(x[:, 5:14], *P['C'], padding=1))
    for name, ref in (('A', ra), ('B', rb), ('C', rc)):
        err = (net.tensor(name).cpu() - ref).abs().max().item()
        assert err <= _tol(ref), (name, err, _tol(ref))
    net.close()


@pytest.fixture
def fuse_upsample():
    from panoptic_forecasting_amd import lib as pflib
    L = pflib.load()
    pflib.check(L.pf_set_option(b'fuse_upsample', 1), 'pf_set_option')
    yield
    L.pf_set_option(b'fuse_upsample', 1)   # library default


@pytest.mark.parametrize('force', [(0, 0, 0, 0), (1, 4, 2, 0), (1, 2, 1, 0), (1, 1, 2, 0), (2, 2, 2, 2), (2, 4, 1, 4)],
                         ids=lambda f: 'k%d_%d_%d_%d' % f)
@pytest.mark.parametrize('h,w,b', [(32, 64, 1), (20, 40, 2), (34, 52, 1)])
def test_fused_pool_and_commuted_upsample(h, w, b, force, fuse_upsample, force_conv):
    """The executor's fused stages (conv_epilogue.h): 1x1 conv + AvgPool2d in one launch, and
    TransitionUp + 1x1 conv over cat([up(x), skip]) evaluated as W_skip*skip + up(W_x*x), with the residual window
    staged in LDS - through both conv kernels (forced shapes)."""
    from helpers import MiniNet, MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    g = torch.Generator().manual_seed(h * 7 + w)
    x = torch.randn(b, 12, h, w, generator=g)
    spec = MiniSpec(12)
    c1 = spec.conv('c1', [arch.Src(0, 0, 12)], 20, 3)
    c2 = spec.conv('c2', [arch.Src(c1, 0, 20)], 24, 1)
    p = spec.pool('p', c2)
    c3 = spec.conv('c3', [arch.Src(p, 0, 24)], 40, 3)          # 40 >= 37: the commuted form is legal (cout <= cin_up)
    up = spec.upsample('up', c3, c1)
    spec.conv('c4', [arch.Src(up, 0, 40), arch.Src(c1, 0, 20)], 37, 1)
    P = {}
    for name, cin, cout, k in [('c1', 12, 20, 3), ('c2', 20, 24, 1), ('c3', 24, 40, 3), ('c4', 60, 37, 1)]:
        P[name] = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, torch.randn(cout, generator=g))
    force_conv(*force)
    net = MiniNet(spec, P).run(x.cuda())
    r1 = F.relu(F.conv2d(x, *P['c1'], padding=1))
    r2 = F.relu(F.conv2d(r1, *P['c2']))
    rp = F.avg_pool2d(r2, 2, 2)
    r3 = F.relu(F.conv2d(rp, *P['c3'], padding=1))
    ru = F.interpolate(r3, size=(h, w), mode='bilinear', align_corners=True)
    r4 = F.relu(F.conv2d(torch.cat([ru, r1], 1), *P['c4']))
    assert (net.tensor('p').cpu() - rp).abs().max() <= _tol(rp)
    assert (net.tensor('c3').cpu() - r3).abs().max() <= _tol(r3)
    assert (net.tensor('c4').cpu() - r4).abs().max() <= 2 * _tol(r4)
    net.close()


@pytest.mark.parametrize('cin,cout,h,w', [(58, 18, 32, 64), (20, 36, 24, 40), (33, 24, 40, 96), (12, 20, 17, 32),
                                          (91, 28, 24, 64), (76, 28, 9, 36), (163, 46, 16, 32), (16, 31, 20, 48),
                                          (30, 17, 12, 32), (40, 45, 8, 64), (9, 62, 8, 32)])
def test_valu_remainder_path(cin, cout, h, w, force_conv):
    """cout % 16 != 0: conv_dma runs the full 16-cout tiles on the matrix cores and the 1..15 trailing channels on
    the vector ALU (permlane-transposed A operands, DPP-selected weights; forced to conv_dma WM=4 here; big images
    take this path by themselves)."""
    from helpers import MiniNet, MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(2, cin, h, w, generator=g)
    spec = MiniSpec(cin)
    a, bch = cin // 3, cin - cin // 3
    spec.conv('c', [arch.Src(0, bch, a), arch.Src(0, 0, bch)], cout, 3)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g)
    force_conv(1, 4, 2, 0)
    net = MiniNet(spec, {'c': (wt, bias)}).run(x.cuda())
    ref = F.relu(F.conv2d(torch.cat([x[:, bch:], x[:, :bch]], 1), wt, bias, padding=1))
    err = (net.tensor('c').cpu() - ref).abs().max().item()
    assert err <= _tol(ref), (err, _tol(ref))
    net.close()


@pytest.mark.parametrize('nt,wide', [(1, 0), (2, 0), (3, 0), (1, 1), (2, 1), (3, 1)])
@pytest.mark.parametrize('cin,cout,h,w', [(48, 10, 24, 64), (58, 18, 17, 128), (91, 28, 20, 72), (7, 16, 5, 60),
                                          (33, 46, 9, 36), (163, 46, 16, 32), (16, 24, 40, 96)])
def test_split_conv(cin, cout, h, w, nt, wide, force_conv):
    """conv_split: fp32 operands split into two fp16 terms (hi + mid, 11 + 11 significand bits; weights pre-scaled by an
    exact power of two per conv), three products on v_mfma_f32_16x16x32_f16, fp32 accumulate.  Stated tolerance = the one
    of the fp32 kernels, 2e-5 * (1 + max|ref|) (the dropped cross terms are <= 2^-21 of each product); inputs with a wide
    dynamic range so that the mid terms matter: with hi-only operands this test fails by two orders of magnitude."""
    from helpers import MiniNet, MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    from panoptic_forecasting_amd import lib as pflib
    g = torch.Generator().manual_seed(cin * 5 + cout)
    x = torch.randn(2, cin, h, w, generator=g) * torch.exp(torch.randn(2, cin, 1, 1, generator=g))
    spec = MiniSpec(cin)
    a, bch = cin // 3, cin - cin // 3
    spec.conv('c', [arch.Src(0, bch, a), arch.Src(0, 0, bch)], cout, 3)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g)
    force_conv(4, nt, wide, 0)
    pflib.profile(True)
    net = MiniNet(spec, {'c': (wt, bias)}).run(x.cuda())
    labels = [r['label'] for r in pflib.profile_results()]
    pflib.profile(False)
    assert any('conv_split_kernel' in l for l in labels), labels
    ref = F.relu(F.conv2d(torch.cat([x[:, bch:], x[:, :bch]], 1).double(), wt.double(), bias.double(), padding=1)).float()
    err = (net.tensor('c').cpu() - ref).abs().max().item()
    assert err <= _tol(ref), (err, ref.abs().max().item())
    net.close()


def _tol_split(ref):
    """split kernels (two fp16 terms per operand): the tolerance of the fp32 kernels"""
    return _tol(ref)


@pytest.mark.parametrize('nt', [1, 2, 3, 4])
@pytest.mark.parametrize('cin,cout,h,w,b', [(48, 64, 16, 64, 2), (78, 96, 9, 36, 1), (45, 70, 20, 40, 2), (160, 11, 8, 32, 1),
                                            (5, 3, 7, 12, 1), (214, 224, 4, 8, 2)])
def test_split_conv1x1(cin, cout, h, w, b, nt, force_conv):
    """conv_split1 (1x1, K = 4 groups of 8 channels per fp16 MFMA): two input ranges (channel chunks of 32 with tails)."""
    from helpers import MiniNet, MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    from panoptic_forecasting_amd import lib as pflib
    g = torch.Generator().manual_seed(cin + cout * 3)
    x = torch.randn(b, cin, h, w, generator=g) * torch.exp(torch.randn(b, cin, 1, 1, generator=g))
    spec = MiniSpec(cin)
    a, bch = cin // 3, cin - cin // 3
    srcs = [arch.Src(0, bch, a), arch.Src(0, 0, bch)] if a > 0 else [arch.Src(0, 0, cin)]
    spec.conv('c', srcs, cout, 1, relu=False)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    bias = torch.randn(cout, generator=g)
    force_conv(4, nt, 0, 0)
    pflib.profile(True)
    net = MiniNet(spec, {'c': (wt, bias)}).run(x.cuda())
    labels = [r['label'] for r in pflib.profile_results()]
    pflib.profile(False)
    assert any('conv_split1_kernel' in l for l in labels), labels
    xin = torch.cat([x[:, bch:], x[:, :bch]], 1) if a > 0 else x
    ref = F.conv2d(xin.double(), wt.double(), bias.double()).float()
    err = (net.tensor('c').cpu() - ref).abs().max().item()
    assert err <= _tol_split(ref), (err, ref.abs().max().item())
    net.close()


@pytest.mark.parametrize('nt', [1, 2, 4])
@pytest.mark.parametrize('h,w,b', [(32, 64, 1), (20, 40, 2), (34, 52, 1)])
def test_split_fused_pool_and_commuted_upsample(h, w, b, nt, fuse_upsample, force_conv):
    """The fused epilogue stages (2x2 pool; TransitionUp + 1x1 evaluated as W_skip*skip + up(W_x*x)) behind the split
    1x1 kernel; the 3x3 layers of the little network run on conv_split as well."""
    from helpers import MiniNet, MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    g = torch.Generator().manual_seed(h * 7 + w)
    x = torch.randn(b, 12, h, w, generator=g)
    spec = MiniSpec(12)
    c1 = spec.conv('c1', [arch.Src(0, 0, 12)], 20, 3)
    c2 = spec.conv('c2', [arch.Src(c1, 0, 20)], 24, 1)
    p = spec.pool('p', c2)
    c3 = spec.conv('c3', [arch.Src(p, 0, 24)], 40, 3)
    up = spec.upsample('up', c3, c1)
    spec.conv('c4', [arch.Src(up, 0, 40), arch.Src(c1, 0, 20)], 37, 1)
    P = {}
    for name, cin, cout, k in [('c1', 12, 20, 3), ('c2', 20, 24, 1), ('c3', 24, 40, 3), ('c4', 60, 37, 1)]:
        P[name] = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, torch.randn(cout, generator=g))
    force_conv(4, nt, 0, 0)
    net = MiniNet(spec, P).run(x.cuda())
    r1 = F.relu(F.conv2d(x, *P['c1'], padding=1))
    r2 = F.relu(F.conv2d(r1, *P['c2']))
    rp = F.avg_pool2d(r2, 2, 2)
    r3 = F.relu(F.conv2d(rp, *P['c3'], padding=1))
    ru = F.interpolate(r3, size=(h, w), mode='bilinear', align_corners=True)
    r4 = F.relu(F.conv2d(torch.cat([ru, r1], 1), *P['c4']))
    assert (net.tensor('p').cpu() - rp).abs().max() <= _tol_split(rp)
    assert (net.tensor('c3').cpu() - r3).abs().max() <= _tol_split(r3)
    assert (net.tensor('c4').cpu() - r4).abs().max() <= 2 * _tol_split(r4)
    net.close()


# ---- packed-pair ("S4") activation layout: conv_s4.hip ---------------------------------------------------------------
def test_s4_layout_round_trip():
    """pf_s4_pack / pf_s4_unpack: [B][2][ceil(C/4)][H][W][4] fp16, hi = fp16(x), mid = fp16(x - hi), both rounded to
    nearest even (conv_mfma.h: split_terms2; torch's .half() is the same rounding): bit patterns exact, hi + mid is x to
    2^-23 |x| + 2^-25 for |x| <= 65504 - and a value outside that range raises PF_STATUS_RANGE instead of clamping."""
    from panoptic_forecasting_amd import lib as pflib
    L = pflib.load()
    g = torch.Generator().manual_seed(3)
    b, c, h, w = 2, 10, 6, 8
    x = (torch.randn(b, c, h, w, generator=g) * torch.exp(2 * torch.randn(b, c, 1, 1, generator=g)))
    x[0, 0, 0, :8] = torch.tensor([0.0, 1e-7, -3e-5, 65504.0, -65504.0, 0.24999999, 1023.4999, 6.1e-5])
    x[0, 1, 0, :4] = torch.tensor([2049.0, 2051.0, -4098.0, 1.0 + 2.0 ** -11])     # ties of the hi term: to even
    x = x.cuda()
    c4 = (c + 3) // 4
    status = torch.zeros(1, dtype=torch.int32, device='cuda')

    def pack(src):
        packed = torch.zeros(b, 2, c4, h, w, 4, dtype=torch.float16, device='cuda')
        status.zero_()
        pflib.check(L.pf_s4_pack(src.data_ptr(), packed.data_ptr(), b, c, h, w, status.data_ptr(), pflib.stream_ptr()), 'pf_s4_pack')
        return packed
    packed = pack(x)
    back = torch.empty_like(x)
    pflib.check(L.pf_s4_unpack(packed.data_ptr(), back.data_ptr(), b, c, h, w, pflib.stream_ptr()), 'pf_s4_unpack')
    torch.cuda.synchronize()
    assert status.item() == 0
    xp = torch.zeros(b, c4 * 4, h, w, device='cuda')
    xp[:, :c] = x
    hi = xp.half()
    mid = (xp - hi.float()).half()
    want = torch.stack([hi, mid], 1).view(b, 2, c4, 4, h, w).permute(0, 1, 2, 4, 5, 3).contiguous()
    assert torch.equal(packed.view(torch.int16), want.view(torch.int16))
    assert torch.equal(back, hi.float()[:, :c] + mid.float()[:, :c])
    err = (back.double() - x.double()).abs()
    assert (err <= 2.0 ** -23 * x.double().abs() + 2.0 ** -25).all(), (err / x.double().abs().clamp_min(1e-30)).max().item()
    # out of range: the flag, not a clamp
    for bad in (65505.0, -70000.0, 131000.0, 1e6, float('inf'), float('nan')):
        y = x.clone()
        y[1, 3, 2, 5] = bad
        pack(y)
        torch.cuda.synchronize()
        assert status.item() & 1, bad
    # (65504, 65520) still rounds to 65504 but is flagged all the same: the guard is |x| <= 65504, not "hi is finite"


def _block_net(g, cin0):
    """A HarDBlock-shaped little network (hardnet.py:220-240 wiring: slots of a concatenated output tensor at channel offsets
    0 / 10 / 20, an own tensor for the even layer), a pooled transition, a commuted upsample and a final conv."""
    from helpers import MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    S = arch.Src
    spec = MiniSpec(cin0)
    t0 = spec.conv('t0', [S(0, 0, cin0)], 20, 3)
    out = spec.tensor('out', 48)
    spec.conv('L1', [S(t0, 0, 20)], 10, 3, dst=out, dst_choff=0)
    l2 = spec.conv('L2', [S(out, 0, 10), S(t0, 0, 20)], 18, 3)
    spec.conv('L3', [S(l2, 0, 18)], 10, 3, dst=out, dst_choff=10)
    spec.conv('L4', [S(out, 10, 10), S(l2, 0, 18), S(t0, 0, 20)], 28, 3, dst=out, dst_choff=20)
    c5 = spec.conv('c5', [S(out, 0, 48)], 24, 1)
    p = spec.pool('p', c5)
    c6 = spec.conv('c6', [S(p, 0, 24)], 40, 3)
    up = spec.upsample('up', c6, out)
    c7 = spec.conv('c7', [S(up, 0, 40), S(out, 0, 48)], 37, 1)
    spec.conv('c8', [S(c7, 0, 37)], 11, 3, relu=False)
    shapes = [('t0', cin0, 20, 3), ('L1', 20, 10, 3), ('L2', 30, 18, 3), ('L3', 18, 10, 3), ('L4', 48, 28, 3), ('c5', 48, 24, 1),
              ('c6', 24, 40, 3), ('c7', 88, 37, 1), ('c8', 37, 11, 3)]
    P = {n: (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5, torch.randn(co, generator=g)) for n, ci, co, k in shapes}
    return spec, P


def _block_ref(x, P, h, w):
    D = {k: (v[0].double(), v[1].double()) for k, v in P.items()}
    x = x.double()
    t0 = F.relu(F.conv2d(x, *D['t0'], padding=1))
    l1 = F.relu(F.conv2d(t0, *D['L1'], padding=1))
    l2 = F.relu(F.conv2d(torch.cat([l1, t0], 1), *D['L2'], padding=1))
    l3 = F.relu(F.conv2d(l2, *D['L3'], padding=1))
    l4 = F.relu(F.conv2d(torch.cat([l3, l2, t0], 1), *D['L4'], padding=1))
    out = torch.cat([l1, l3, l4], 1)
    p = F.avg_pool2d(F.relu(F.conv2d(out, *D['c5'])), 2, 2)
    c6 = F.relu(F.conv2d(p, *D['c6'], padding=1))
    c7 = F.relu(F.conv2d(torch.cat([F.interpolate(c6, size=(h, w), mode='bilinear', align_corners=True), out], 1), *D['c7']))
    c8 = F.conv2d(c7, *D['c8'], padding=1)
    return {'t0': t0, 'L2': l2, 'out': out, 'p': p, 'c6': c6, 'c7': c7, 'c8': c8}


@pytest.mark.parametrize('nt,wide', [(1, 0), (2, 0), (3, 0), (1, 1), (2, 1), (1, 2), (2, 2)])   # wide 1: 8x64 tiles, 2: 16x32 tiles on 8 waves
@pytest.mark.parametrize('h,w,b', [(16, 64, 2), (24, 40, 1), (34, 136, 1)])
def test_packed_activation_block(h, w, b, nt, wide, force_conv):
    """Every tensor between the first and the last conv lives in the packed-pair layout; the S4 3x3 kernel (LDS-DMA halo
    tiles, slots of a shared output tensor at offsets that are not multiples of 4, zero-filled group tails) and the S4 1x1
    kernel (plain, pooled, low-resolution half, upsampled residual) against float64 torch.  Tolerance per layer as for
    conv_split (2e-5 (1 + max|ref|), as the fp32 kernels), 3x that after the chain of 6."""
    from helpers import MiniNet
    from panoptic_forecasting_amd import lib as pflib
    g = torch.Generator().manual_seed(h * 3 + w + nt)
    x = torch.randn(b, 12, h, w, generator=g) * torch.exp(torch.randn(b, 12, 1, 1, generator=g))
    spec, P = _block_net(g, 12)
    force_conv(5, nt, wide, 0)
    pflib.profile(True)
    net = MiniNet(spec, P).run(x.cuda())
    labels = [r['label'] for r in pflib.profile_results()]
    pflib.profile(False)
    assert sum('conv_s4_kernel' in l for l in labels) >= 1, labels
    assert any('32, 16, 1>' in l for l in labels) == (wide == 2), labels
    for tag in ('+pool', '+res', 'lowres-half'):
        assert any('conv_s4_1x1_kernel' in l and tag in l for l in labels), (tag, labels)
    ref = _block_ref(x, P, h, w)
    for name, scale in [('t0', 1), ('L2', 1), ('out', 2), ('p', 2), ('c6', 3), ('c7', 3), ('c8', 3)]:
        r = ref[name].float()
        err = (net.tensor(name).cpu() - r).abs().max().item()
        assert err <= scale * _tol_split(r), (name, err, _tol_split(r))
    net.close()


@pytest.mark.parametrize('h,w,b', [(16, 64, 2), (24, 40, 1)])
def test_s4_four_cout_tiles(h, w, b, force_conv):
    """conv_s4_kernel<4, 32> (the table's shape for the 52-cout layers at B=16: one pixel fragment feeds four cout tiles):
    couts 52 and 70 = 4 and 5 tiles (a second, partial cout group), against float64 torch at the split tolerance."""
    from helpers import MiniNet, MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    from panoptic_forecasting_amd import lib as pflib
    S = arch.Src
    g = torch.Generator().manual_seed(h + w)
    x = torch.randn(b, 12, h, w, generator=g) * torch.exp(torch.randn(b, 12, 1, 1, generator=g))
    spec = MiniSpec(12)
    t0 = spec.conv('t0', [S(0, 0, 12)], 20, 3)
    c1 = spec.conv('c1', [S(t0, 0, 20)], 52, 3)
    c2 = spec.conv('c2', [S(c1, 0, 52), S(t0, 0, 20)], 70, 3)
    spec.conv('c3', [S(c2, 0, 70)], 11, 3, relu=False)
    shapes = [('t0', 12, 20), ('c1', 20, 52), ('c2', 72, 70), ('c3', 70, 11)]
    P = {n: (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5, torch.randn(co, generator=g)) for n, ci, co in shapes}
    force_conv(5, 4, 0, 0)
    pflib.profile(True)
    net = MiniNet(spec, P).run(x.cuda())
    labels = [r['label'] for r in pflib.profile_results()]
    pflib.profile(False)
    assert any('conv_s4_kernel<4, 32, 8, 1>' in l for l in labels), labels
    D = {k: (v[0].double(), v[1].double()) for k, v in P.items()}
    t0r = F.relu(F.conv2d(x.double(), *D['t0'], padding=1))
    c1r = F.relu(F.conv2d(t0r, *D['c1'], padding=1))
    c2r = F.relu(F.conv2d(torch.cat([c1r, t0r], 1), *D['c2'], padding=1))
    c3r = F.conv2d(c2r, *D['c3'], padding=1)
    for name, r, scale in [('c1', c1r, 1), ('c2', c2r, 2), ('c3', c3r, 3)]:
        r = r.float()
        err = (net.tensor(name).cpu() - r).abs().max().item()
        assert err <= scale * _tol_split(r), (name, err, _tol_split(r))
    net.close()


@pytest.mark.parametrize('nt,ks', [(1, 2), (2, 2), (1, 4), (2, 4)])
@pytest.mark.parametrize('h,w,b', [(16, 64, 2), (24, 40, 1), (32, 64, 3)])
def test_s4_k_split(h, w, b, nt, ks, force_conv):
    """conv_s4_kernel<NT, 32, 8, KS> (the shapes of the small levels: KS = 2 / 4 wave groups of a workgroup split the rounds of
    one pixel tile and hand their sums over through LDS): layers of 9, 16 and 9 rounds - uneven parts, parts that end early, two
    input ranges - against float64 torch at the split tolerance; a layer with fewer than 16 (8) rounds falls back to KS = 2 (1)."""
    from helpers import MiniNet, MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    from panoptic_forecasting_amd import lib as pflib
    S = arch.Src
    g = torch.Generator().manual_seed(h + w + nt)
    x = torch.randn(b, 12, h, w, generator=g) * torch.exp(torch.randn(b, 12, 1, 1, generator=g))
    spec = MiniSpec(12)
    t0 = spec.conv('t0', [S(0, 0, 12)], 72, 3)
    c1 = spec.conv('c1', [S(t0, 0, 72)], 52, 3)
    c2 = spec.conv('c2', [S(c1, 0, 52), S(t0, 0, 72)], 70, 3)
    c3 = spec.conv('c3', [S(c2, 0, 70)], 24, 3)
    spec.conv('c4', [S(c3, 0, 24)], 11, 3, relu=False)
    shapes = [('t0', 12, 72), ('c1', 72, 52), ('c2', 124, 70), ('c3', 70, 24), ('c4', 24, 11)]
    P = {n: (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5, torch.randn(co, generator=g)) for n, ci, co in shapes}
    force_conv(5, nt, 3 if ks == 2 else 4, 0)
    pflib.profile(True)
    net = MiniNet(spec, P).run(x.cuda())
    labels = [r['label'] for r in pflib.profile_results()]
    pflib.profile(False)
    assert any('conv_s4_kernel<%d, 32, 8, 2>' % nt in l for l in labels), labels            # c1, c3 (9 rounds); c2 too when KS = 2
    assert any('conv_s4_kernel<%d, 32, 8, 4>' % nt in l for l in labels) == (ks == 4), labels    # c2: 16 rounds
    assert any('conv_s4_kernel<1, 32, 8, 1>' in l for l in labels), labels                  # c4: 3 rounds, one cout tile
    D = {k: (v[0].double(), v[1].double()) for k, v in P.items()}
    t0r = F.relu(F.conv2d(x.double(), *D['t0'], padding=1))
    c1r = F.relu(F.conv2d(t0r, *D['c1'], padding=1))
    c2r = F.relu(F.conv2d(torch.cat([c1r, t0r], 1), *D['c2'], padding=1))
    c3r = F.relu(F.conv2d(c2r, *D['c3'], padding=1))
    c4r = F.conv2d(c3r, *D['c4'], padding=1)
    for name, r, scale in [('c1', c1r, 1), ('c2', c2r, 2), ('c3', c3r, 3), ('c4', c4r, 3)]:
        r = r.float()
        err = (net.tensor(name).cpu() - r).abs().max().item()
        assert err <= scale * _tol_split(r), (name, err, _tol_split(r))
    net.close()


@pytest.mark.parametrize('force', [(0, 0, 0, 0), (5, 1, 0, 0), (5, 2, 0, 0), (5, 3, 0, 0), (1, 4, 1, 0), (1, 2, 2, 0), (4, 1, 0, 0), (4, 2, 1, 0)],
                         ids=lambda f: 'k%d_%d_%d_%d' % f)
def test_xcd_tile_order_with_many_cout_groups(force, force_conv):
    """A tile count that is a multiple of 8 (64x128 image: 8x4 tiles of 8x32) switches every conv kernel to the XCD-aware
    workgroup order (xcd_tile_order, conv_mfma.h: id % 8 = XCD band, the cout groups of a tile back to back); couts of 70 /
    90 / 40 give 2-5 cout groups per tile at the forced shapes.  3x3, 1x1 and the stride-2 conv against float64 torch."""
    from helpers import MiniNet, MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    S = arch.Src
    g = torch.Generator().manual_seed(force[0] * 7 + force[1])
    b, h, w = 2, 64, 128
    x = torch.randn(b, 12, h, w, generator=g) * torch.exp(torch.randn(b, 12, 1, 1, generator=g))
    spec = MiniSpec(12)
    t0 = spec.conv('t0', [S(0, 0, 12)], 20, 3)
    c1 = spec.conv('c1', [S(t0, 0, 20)], 70, 3)
    c2 = spec.conv('c2', [S(c1, 0, 70), S(t0, 0, 20)], 90, 1)
    c3 = spec.conv('c3', [S(c2, 0, 90)], 40, 3, stride=2)
    spec.conv('c4', [S(c3, 0, 40)], 11, 3, relu=False)
    shapes = [('t0', 12, 20, 3), ('c1', 20, 70, 3), ('c2', 90, 90, 1), ('c3', 90, 40, 3), ('c4', 40, 11, 3)]
    P = {n: (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5, torch.randn(co, generator=g)) for n, ci, co, k in shapes}
    force_conv(*force)
    net = MiniNet(spec, P).run(x.cuda())
    D = {k: (v[0].double(), v[1].double()) for k, v in P.items()}
    t0r = F.relu(F.conv2d(x.double(), *D['t0'], padding=1))
    c1r = F.relu(F.conv2d(t0r, *D['c1'], padding=1))
    c2r = F.relu(F.conv2d(torch.cat([c1r, t0r], 1), *D['c2']))
    c3r = F.relu(F.conv2d(c2r, *D['c3'], padding=1, stride=2))
    c4r = F.conv2d(c3r, *D['c4'], padding=1)
    for name, r, scale in [('c1', c1r, 1), ('c2', c2r, 2), ('c3', c3r, 3), ('c4', c4r, 4)]:
        r = r.float()
        err = (net.tensor(name).cpu() - r).abs().max().item()
        assert err <= scale * _tol_split(r), (name, err, _tol_split(r))
    net.close()


def test_packed_activations_off_is_fp32_layout(force_conv):
    """pf_set_option('packed_acts', 0): the same network, no S4 kernel launched, same results within the split tolerance."""
    from helpers import MiniNet
    from panoptic_forecasting_amd import lib as pflib
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 12, 16, 64, generator=g)
    spec, P = _block_net(g, 12)
    L = pflib.load()
    pflib.check(L.pf_set_option(b'packed_acts', 0), 'pf_set_option')
    try:
        pflib.profile(True)
        net = MiniNet(spec, P).run(x.cuda())
        labels = [r['label'] for r in pflib.profile_results()]
        pflib.profile(False)
    finally:
        pflib.check(L.pf_set_option(b'packed_acts', 1), 'pf_set_option')
    assert not any('conv_s4' in l for l in labels), labels
    ref = _block_ref(x, P, 16, 64)
    for name in ('out', 'c7', 'c8'):
        r = ref[name].float()
        assert (net.tensor(name).cpu() - r).abs().max().item() <= 3 * _tol_split(r)
    net.close()


def _pair_net(g, cin0, c_odd, c_even, deep):
    """HarDBlock wiring (hardnet.py:177-194) with free channel counts: L1 = f(x0), L2 = f(L1 ++ x0), L3 = f(L2), L4 = f(L3 ++ L2 ++ x0)
    and, `deep`, L5 = f(L4), L6 = f(L5 ++ L4), L7 = f(L6), L8 = f(L7 ++ L6 ++ L4 ++ x0); odd layers are slots of the block's
    output tensor (offsets that are not multiples of 4), even layers tensors of their own, then a 3x3 conv over the output."""
    from helpers import MiniSpec
    from panoptic_forecasting_amd import hardnet_arch as arch
    S = arch.Src
    spec = MiniSpec(cin0)
    c0 = 24
    t0 = spec.conv('t0', [S(0, 0, cin0)], c0, 3)
    n_odd = 4 if deep else 2
    out_ch = n_odd * c_odd + c_even
    out = spec.tensor('out', out_ch)
    shapes = [('t0', cin0, c0)]
    spec.conv('L1', [S(t0, 0, c0)], c_odd, 3, dst=out, dst_choff=0)
    l2 = spec.conv('L2', [S(out, 0, c_odd), S(t0, 0, c0)], c_even, 3)
    spec.conv('L3', [S(l2, 0, c_even)], c_odd, 3, dst=out, dst_choff=c_odd)
    shapes += [('L1', c0, c_odd), ('L2', c_odd + c0, c_even), ('L3', c_even, c_odd)]
    if not deep:
        spec.conv('L4', [S(out, c_odd, c_odd), S(l2, 0, c_even), S(t0, 0, c0)], c_even, 3, dst=out, dst_choff=2 * c_odd)
        shapes += [('L4', c_odd + c_even + c0, c_even)]
    else:
        l4 = spec.conv('L4', [S(out, c_odd, c_odd), S(l2, 0, c_even), S(t0, 0, c0)], c_even, 3)
        spec.conv('L5', [S(l4, 0, c_even)], c_odd, 3, dst=out, dst_choff=2 * c_odd)
        l6 = spec.conv('L6', [S(out, 2 * c_odd, c_odd), S(l4, 0, c_even)], c_even, 3)
        spec.conv('L7', [S(l6, 0, c_even)], c_odd, 3, dst=out, dst_choff=3 * c_odd)
        spec.conv('L8', [S(out, 3 * c_odd, c_odd), S(l6, 0, c_even), S(l4, 0, c_even), S(t0, 0, c0)], c_even, 3, dst=out, dst_choff=4 * c_odd)
        shapes += [('L4', c_odd + c_even + c0, c_even), ('L5', c_even, c_odd), ('L6', c_odd + c_even, c_even), ('L7', c_even, c_odd),
                   ('L8', c_odd + 2 * c_even + c0, c_even)]
    spec.conv('fin', [S(out, 0, out_ch)], 9, 3, relu=False)
    shapes += [('fin', out_ch, 9)]
    P = {n: (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5, torch.randn(co, generator=g) * 0.5) for n, ci, co in shapes}
    return spec, P


def _pair_ref(x, P, deep):
    D = {k: (v[0].double(), v[1].double()) for k, v in P.items()}
    cv = lambda n, t: F.relu(F.conv2d(t, *D[n], padding=1))
    t0 = cv('t0', x.double())
    l1 = cv('L1', t0)
    l2 = cv('L2', torch.cat([l1, t0], 1))
    l3 = cv('L3', l2)
    l4 = cv('L4', torch.cat([l3, l2, t0], 1))
    if not deep:
        out = torch.cat([l1, l3, l4], 1)
        return {'L2': l2, 'out': out, 'fin': F.conv2d(out, *D['fin'], padding=1)}
    l5 = cv('L5', l4)
    l6 = cv('L6', torch.cat([l5, l4], 1))
    l7 = cv('L7', l6)
    l8 = cv('L8', torch.cat([l7, l6, l4, t0], 1))
    out = torch.cat([l1, l3, l5, l7, l8], 1)
    return {'L2': l2, 'L4': l4, 'L6': l6, 'out': out, 'fin': F.conv2d(out, *D['fin'], padding=1)}


@pytest.mark.parametrize('c_odd,c_even,deep', [(10, 18, False), (10, 28, False), (16, 46, False), (18, 30, True), (24, 40, True), (6, 14, False), (32, 48, False),
                                               (12, 20, True), (10, 17, False)])
@pytest.mark.parametrize('h,w,b', [(16, 64, 2), (21, 44, 1), (40, 100, 3)])
def test_conv_pair_vs_float64_and_two_launches(c_odd, c_even, deep, h, w, b, force_conv):
    """conv_pair.hip: an odd HarDBlock layer computed inside its consumer (one launch per pair: S staged once with a two-pixel halo,
    P on the tile plus one halo pixel into LDS planes, zero outside the image, P's own pixels to its slot of the block output).  Every
    cout-tile combination the kernel is built for (C: 1-3 tiles, P: 1-2), 4- and 8-layer blocks (2-4 source ranges), image sizes
    with partial tiles and widths that are multiples of 4 only, batches: against float64 torch at the tolerance of the two-launch
    path, and within 1e-5 (1 + max) of what that path (fuse_pairs = 0) stores.  Pairs of (<= 12) -> (17 .. 20) channels also run
    MERGED (fuse_pairs = 2; 3 = never merged): P's weights ride in the rows C's second cout tile pads with zeros, so P at the
    tile's own pixels comes out of C's matrix instructions and only the 84 halo positions are computed separately."""
    from helpers import MiniNet
    from panoptic_forecasting_amd import lib as pflib
    g = torch.Generator().manual_seed(h * 7 + w + c_even)
    x = torch.randn(b, 12, h, w, generator=g) * torch.exp(0.5 * torch.randn(b, 12, 1, 1, generator=g))
    spec, P = _pair_net(g, 12, c_odd, c_even, deep)
    ref = _pair_ref(x, P, deep)
    force_conv(5, 2, 0, 0)
    got = {}
    merged = 16 < c_even <= 20 and c_odd <= 12
    modes = (0, 2, 3) if merged else (0, 2)
    for fuse in modes:
        net = MiniNet(spec, P).set_option('fuse_pairs', fuse)
        pflib.profile(True)
        net.run(x.cuda())
        labels = [r['label'] for r in pflib.profile_results()]
        pflib.profile(False)
        n_pair = sum('conv_pair_kernel' in l for l in labels)
        assert n_pair == (0 if fuse == 0 else 1), labels      # (one label per kernel shape: the pairs of a block share theirs)
        if fuse:
            nt, ntp = (c_even + 15) // 16, (c_odd + 15) // 16
            assert any('conv_pair_kernel<%d, %d, %d>' % (nt, ntp, merged and fuse == 2) in l for l in labels), labels
        got[fuse] = {k: net.tensor(k).cpu() for k in ref}
        assert net.status() == 0
        net.close()
    for name, r in ref.items():
        r = r.float()
        scale = 3 if name in ('out', 'fin') else 2
        for fuse in modes:
            err = (got[fuse][name] - r).abs().max().item()
            assert err <= scale * _tol_split(r), (name, fuse, err, _tol_split(r))
        for fuse in modes[1:]:
            d = (got[0][name] - got[fuse][name]).abs().max().item()
            assert d <= 1e-5 * (1.0 + r.abs().max().item()), (name, fuse, d)
