"""Test-only helpers: tiny custom op tables run through the C ABI (pf_hardnet_forward_dense)."""
import ctypes

import torch

from panoptic_forecasting_amd import hardnet_arch as arch
from panoptic_forecasting_amd import lib as _lib
from panoptic_forecasting_amd import packing


class MiniSpec:
    """An op table built by hand (same fields packing.pack_blob reads from hardnet_arch.Spec)."""

    def __init__(self, in_ch):
        self.in_ch, self.n_cls = in_ch, 1
        self.tensors = [arch.Tensor('input', in_ch)]
        self.ops = []
        self.input_tensor = 0

    def tensor(self, name, ch):
        self.tensors.append(arch.Tensor(name, ch))
        return len(self.tensors) - 1

    def conv(self, name, srcs, cout, k, stride=1, dst=None, dst_choff=0, relu=True):
        cin = sum(s.ch for s in srcs)
        if dst is None:
            dst = self.tensor(name, cout)
        self.ops.append(arch.Op(arch.OP_CONV, name, srcs, dst, dst_choff, cin, cout, k, stride, relu, bn=False))
        return dst

    def pool(self, name, src_t):
        ch = self.tensors[src_t].channels
        d = self.tensor(name, ch)
        self.ops.append(arch.Op(arch.OP_POOL, name, [arch.Src(src_t, 0, ch)], d, 0, ch, ch, 2, 2, False, False))
        return d

    def upsample(self, name, src_t, like_t):
        ch = self.tensors[src_t].channels
        d = self.tensor(name, ch)
        self.ops.append(arch.Op(arch.OP_UPSAMPLE, name, [arch.Src(src_t, 0, ch),
                                                         arch.Src(like_t, 0, self.tensors[like_t].channels)],
                                d, 0, ch, ch, 1, 1, False, False))
        return d

    def conv_ops(self):
        return [o for o in self.ops if o.kind in (arch.OP_STEM, arch.OP_CONV)]


class MiniNet:
    def __init__(self, spec, params):
        L = _lib.load()
        blob = packing.pack_blob(None, spec.in_ch, spec.n_cls, spec=spec, params=params)
        self._buf = ctypes.create_string_buffer(blob, len(blob))
        self.plan = ctypes.c_void_p()
        _lib.check(L.pf_hardnet_plan_create(self._buf, len(blob), spec.in_ch, spec.n_cls, ctypes.byref(self.plan)),
                   'pf_hardnet_plan_create')
        self.spec = spec

    def run(self, x):
        L = _lib.load()
        b, _, h, w = x.shape
        need = ctypes.c_size_t()
        _lib.check(L.pf_hardnet_workspace(self.plan, b, h, w, ctypes.byref(need)), 'pf_hardnet_workspace')
        self.ws = torch.zeros(need.value, dtype=torch.uint8, device=x.device)
        x = x.contiguous()
        rc = L.pf_hardnet_forward_dense(self.plan, x.data_ptr(), b, h, w, 0, 0, None, 0, None, None,
                                        self.ws.data_ptr(), self.ws.numel(), _lib.stream_ptr())
        _lib.check(rc, 'pf_hardnet_forward_dense')
        torch.cuda.synchronize()
        self.bhw = (b, h, w)
        return self

    def tensor(self, name):
        return view_tensor(self.plan, self.ws, name, *self.bhw)

    def close(self):
        _lib.load().pf_hardnet_plan_destroy(self.plan)


def view_tensor(plan, ws, name, b, h, w):
    L = _lib.load()
    off, c, th, tw = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(L.pf_hardnet_tensor_view(plan, name.encode(), b, h, w, ctypes.byref(off), ctypes.byref(c),
                                        ctypes.byref(th), ctypes.byref(tw)), 'pf_hardnet_tensor_view')
    n = b * c.value * th.value * tw.value
    return ws[off.value:off.value + 4 * n].view(torch.float32).view(b, c.value, th.value, tw.value)
