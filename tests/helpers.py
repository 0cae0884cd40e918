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

    def conv(self, name, srcs, cout, k, stride=1, dst=None, dst_choff=0, relu=True, bn=False):
        cin = sum(s.ch for s in srcs)
        if dst is None:
            dst = self.tensor(name, cout)
        self.ops.append(arch.Op(arch.OP_CONV, name, srcs, dst, dst_choff, cin, cout, k, stride, relu, bn=bn))
        return dst

    def head(self, logits_t):
        ch = self.tensors[logits_t].channels
        self.n_cls = ch
        self.ops.append(arch.Op(arch.OP_HEAD, 'head', [arch.Src(logits_t, 0, ch)], logits_t, 0, ch, ch, 1, 1, False, False))

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

    def set_option(self, name, value):
        _lib.check(_lib.load().pf_hardnet_plan_set_option(self.plan, name.encode(), int(value)), 'pf_hardnet_plan_set_option')
        return self

    def status(self):
        """status word of the last forward (include/pfhip.h: PF_STATUS_RANGE = 1), through the C entry point"""
        st = ctypes.c_uint(0xFFFFFFFF)
        _lib.check(_lib.load().pf_hardnet_status(self.ws.data_ptr(), ctypes.byref(st), _lib.stream_ptr()), 'pf_hardnet_status')
        assert st.value == int(self.ws[:4].view(torch.int32).item())
        return st.value

    def close(self):
        _lib.load().pf_hardnet_plan_destroy(self.plan)


def view_tensor(plan, ws, name, b, h, w):
    L = _lib.load()
    off, c, th, tw = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(L.pf_hardnet_tensor_view(plan, name.encode(), b, h, w, ctypes.byref(off), ctypes.byref(c),
                                        ctypes.byref(th), ctypes.byref(tw)), 'pf_hardnet_tensor_view')
    out = torch.empty(b, c.value, th.value, tw.value, dtype=torch.float32, device=ws.device)
    # fp32 copy of the tensor whatever layout the last forward kept it in (packed-pair tensors: hi + mid)
    _lib.check(L.pf_hardnet_tensor_read(plan, name.encode(), b, h, w, ws.data_ptr(), out.data_ptr(), _lib.stream_ptr()),
               'pf_hardnet_tensor_read')
    torch.cuda.synchronize()
    return out


class MiniTrain:
    """A hand-built op table through the training entry points (pf_train_*), dense input.  ``params``: {op name: dict(w=,
    gamma=, beta=, mean=, var=) for conv+BN ops, dict(w=, b=) for plain convs} (CPU tensors)."""

    def __init__(self, spec, params):
        L = _lib.load()
        dummy = {op.name: (torch.zeros(op.cout, op.cin, op.k, op.k), torch.zeros(op.cout)) for op in spec.conv_ops()}
        blob = packing.pack_blob(None, spec.in_ch, spec.n_cls, spec=spec, params=dummy)
        self._buf = ctypes.create_string_buffer(blob, len(blob))
        self.t = ctypes.c_void_p()
        _lib.check(L.pf_train_create(self._buf, len(blob), spec.in_ch, spec.n_cls, ctypes.byref(self.t)), 'pf_train_create')
        n = ctypes.c_size_t()
        _lib.check(L.pf_train_param_count(self.t, ctypes.byref(n)), 'pf_train_param_count')
        self.spec = spec
        host = torch.zeros(n.value)
        self.slots = {}
        for i, op in enumerate(spec.ops):
            if op.kind not in (arch.OP_STEM, arch.OP_CONV):
                continue
            wo, ao, bn = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_int()
            _lib.check(L.pf_train_param_layout(self.t, i, ctypes.byref(wo), ctypes.byref(ao), ctypes.byref(bn)), 'pf_train_param_layout')
            assert bool(bn.value) == bool(op.bn)
            pr = params[op.name]
            nw = op.cout * op.cin * op.k * op.k
            host[wo.value:wo.value + nw] = pr['w'].reshape(-1)
            self.slots[op.name + '.w'] = (wo.value, (op.cout, op.cin, op.k, op.k))
            names = ('gamma', 'beta', 'mean', 'var') if op.bn else ('b',)
            for j, nm in enumerate(names):
                host[ao.value + j * op.cout:ao.value + (j + 1) * op.cout] = pr[nm]
                self.slots[op.name + '.' + nm] = (ao.value + j * op.cout, (op.cout,))
        self.theta = host.cuda()
        self.grad = torch.zeros_like(self.theta)

    def step(self, x, labels, loss_scale=1.0):
        L = _lib.load()
        b, _, h, w = x.shape
        oh, ow = labels.shape[-2:]
        need = ctypes.c_size_t()
        _lib.check(L.pf_train_workspace(self.t, b, h, w, oh, ow, ctypes.byref(need)), 'pf_train_workspace')
        self.ws = torch.zeros(need.value, dtype=torch.uint8, device='cuda')
        self.out3 = torch.zeros(3, dtype=torch.float64, device='cuda')
        x, labels = x.contiguous(), labels.contiguous()
        rc = L.pf_train_forward_backward(self.t, self.theta.data_ptr(), self.grad.data_ptr(), 0, None, 0, None, None, 0.0, 1.0, 1,
                                         x.data_ptr(), b, h, w, labels.data_ptr(), int(labels.dtype == torch.int64), oh, ow, 255,
                                         float(loss_scale), 0.1, 1e-5, 1, self.out3.data_ptr(), self.ws.data_ptr(), self.ws.numel(),
                                         _lib.stream_ptr())
        _lib.check(rc, 'pf_train_forward_backward')
        torch.cuda.synchronize()
        self.dims = (b, h, w, oh, ow)
        return float(self.out3[0] / self.out3[1])

    def param(self, name, grad=False):
        off, shape = self.slots[name]
        n = 1
        for d in shape:
            n *= d
        return (self.grad if grad else self.theta)[off:off + n].view(shape)

    def tensor(self, name, grad=False):
        L = _lib.load()
        off, c, th, tw = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        b = self.dims[0]
        _lib.check(L.pf_train_tensor_view(self.t, name.encode(), int(grad), *self.dims, ctypes.byref(off), ctypes.byref(c),
                                          ctypes.byref(th), ctypes.byref(tw)), 'pf_train_tensor_view')
        n = b * c.value * th.value * tw.value
        return self.ws[off.value:off.value + 4 * n].view(torch.float32).view(b, c.value, th.value, tw.value)

    def close(self):
        _lib.load().pf_train_destroy(self.t)
