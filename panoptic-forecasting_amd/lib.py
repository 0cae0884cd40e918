"""ctypes binding of ``csrc/libpfhip.so`` (C ABI: include/pfhip.h).

The HIP library is the only implementation of the device path: if it is missing or
fails to load this module raises — there is no CPU or PyTorch fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PF_LIBPFHIP') or os.path.join(_HERE, 'csrc', 'libpfhip.so')   # env override: A/B builds

_c = ctypes
_vp, _i, _f, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t

SIGNATURES = {
    'pf_version': (_i, []),
    'pf_last_error': (_c.c_char_p, []),
    'pf_warp_splat_workspace': (_i, [_i, _i, _i, _i, _i, _c.POINTER(_sz)]),
    'pf_warp_splat': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i,
                           _vp, _vp, _vp, _vp, _sz, _vp]),
    'pf_hardnet_plan_create': (_i, [_vp, _sz, _i, _i, _c.POINTER(_vp)]),
    'pf_hardnet_plan_destroy': (None, [_vp]),
    'pf_hardnet_workspace': (_i, [_vp, _i, _i, _i, _c.POINTER(_sz)]),
    'pf_bg_forward': (_i, [_vp, _vp, _i, _vp, _vp, _f, _f, _i, _f, _f, _i, _i, _i, _i, _i, _i,
                           _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    'pf_hardnet_forward_dense': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    'pf_hardnet_tensor_view': (_i, [_vp, _c.c_char_p, _i, _i, _i, _c.POINTER(_sz), _c.POINTER(_i),
                                    _c.POINTER(_i), _c.POINTER(_i)]),
    'pf_hardnet_tensor_read': (_i, [_vp, _c.c_char_p, _i, _i, _i, _vp, _vp, _vp]),
    'pf_hardnet_status': (_i, [_vp, _c.POINTER(_c.c_uint), _vp]),
    'pf_hardnet_status_sticky': (_i, [_vp, _c.POINTER(_c.c_uint), _i, _vp]),
    'pf_hardnet_status_reset': (_i, [_vp, _vp]),
    'pf_hardnet_range_maxima': (_i, [_vp, _vp, _vp, _i, _c.POINTER(_i), _vp]),
    'pf_s4_pack': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'pf_s4_unpack': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'pf_hardnet_flops': (_i, [_vp, _i, _i, _c.POINTER(_c.c_double)]),
    'pf_hop_export': (_i, [_vp, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    'pf_hop_load': (_i, [_vp, _sz, _f, _f, _vp, _vp, _vp]),
    'pf_panoptic_merge_workspace': (_i, [_i, _c.POINTER(_sz)]),
    'pf_panoptic_merge': (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i,
                               _vp, _i, _vp, _sz, _vp]),
    'pf_panoptic_encode': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'pf_panoptic_max_ids': (_i, []),
    'pf_bg_dense_input': (_i, [_vp, _i, _i, _vp, _vp, _f, _f, _i, _i, _i, _i, _vp, _vp]),
    'pf_seg_loss_workspace': (_i, [_i, _i, _i, _c.POINTER(_sz)]),
    'pf_seg_loss': (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'pf_train_create': (_i, [_vp, _sz, _i, _i, _c.POINTER(_vp)]),
    'pf_train_destroy': (None, [_vp]),
    'pf_train_autotune': (_i, [_vp, _i]),
    'pf_train_tuned_shapes': (_i, [_vp, _c.POINTER(_i), _i, _c.POINTER(_i)]),
    'pf_train_path_stats': (_i, [_vp, _c.POINTER(_i), _i, _c.POINTER(_i)]),
    'pf_train_param_count': (_i, [_vp, _c.POINTER(_sz)]),
    'pf_train_param_layout': (_i, [_vp, _i, _c.POINTER(_sz), _c.POINTER(_sz), _c.POINTER(_i)]),
    'pf_train_workspace': (_i, [_vp, _i, _i, _i, _i, _i, _c.POINTER(_sz)]),
    'pf_train_forward_backward': (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _f, _f, _i, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _f, _f,
                                       _f, _i, _vp, _vp, _sz, _vp]),
    'pf_train_tensor_view': (_i, [_vp, _c.c_char_p, _i, _i, _i, _i, _i, _i, _c.POINTER(_sz), _c.POINTER(_i), _c.POINTER(_i),
                                  _c.POINTER(_i)]),
    'pf_sgd_workspace': (_i, [_c.POINTER(_sz)]),
    'pf_sgd_step': (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _vp, _sz, _vp]),
    'pf_set_option': (_i, [_c.c_char_p, _i]),
    'pf_hardnet_plan_set_option': (_i, [_vp, _c.c_char_p, _i]),
    'pf_debug_force_conv': (_i, [_i, _i, _i, _i]),
    'pf_debug_probe_read': (_i, [_c.POINTER(_c.c_longlong)]),
    'pf_profile_enable': (_i, [_i]),
    'pf_profile_collect': (_i, []),
    'pf_profile_get': (_i, [_i, _c.c_char_p, _sz, _c.POINTER(_i), _c.POINTER(_c.c_double),
                            _c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
}


def profile(enable):
    check(load().pf_profile_enable(int(enable)), 'pf_profile_enable')


def profile_results():
    """[{label, launches, ms, flops, bytes}] for everything enqueued since profile(True)."""
    L = load()
    n = L.pf_profile_collect()
    if n < 0:
        check(n, 'pf_profile_collect')
    out = []
    for i in range(n):
        buf = ctypes.create_string_buffer(160)
        la, ms, fl, by = _i(), _c.c_double(), _c.c_double(), _c.c_double()
        check(L.pf_profile_get(i, buf, 160, ctypes.byref(la), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)),
              'pf_profile_get')
        out.append({'label': buf.value.decode(), 'launches': la.value, 'ms': ms.value, 'flops': fl.value,
                    'bytes': by.value})
    return out

_lib = None


class PfError(RuntimeError):
    pass


def load():
    """Load libpfhip.so (once) and type its entry points; raises if it is not built."""
    global _lib
    if _lib is None:
        # torch bundles its own HIP runtime: it must be the one already mapped when libpfhip.so resolves
        # libamdhip64 (loading ours first pulls /opt/rocm's copy and torch then sees "no ROCm-capable device")
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise PfError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          '(make -C panoptic-forecasting_amd/csrc). There is no fallback path.' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the ABI symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise PfError('%s failed (%d): %s' % (what, rc, load().pf_last_error().decode()))


def require_cuda(t, name):
    if not t.is_cuda:
        raise PfError('%s must live on the GPU (got %s): the HIP path has no CPU fallback' % (name, t.device))
    if not t.is_contiguous():
        raise PfError('%s must be contiguous' % name)
    return t


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
