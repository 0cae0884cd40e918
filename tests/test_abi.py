"""The C-ABI library loads and exports every symbol include/pfhip.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

from conftest import ROOT
from panoptic_forecasting_amd import lib as pflib

HEADER = os.path.join(ROOT, 'include', 'pfhip.h')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(pf_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_the_expected_entry_points():
    names = declared_symbols()
    for must in ('pf_version', 'pf_last_error', 'pf_warp_splat_workspace', 'pf_warp_splat',
                 'pf_hardnet_plan_create', 'pf_hardnet_plan_destroy', 'pf_hardnet_workspace', 'pf_bg_forward',
                 'pf_hardnet_forward_dense'):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(pflib.LIB_PATH), 'libpfhip.so not built: run __graft_entry__.build()'
    import torch  # noqa: F401  (its HIP runtime must be mapped first, see lib.load)
    so = ctypes.CDLL(pflib.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(so, n)]
    assert not missing, 'declared in pfhip.h but not exported: %s' % missing


def test_ctypes_signatures_cover_the_header():
    assert sorted(pflib.SIGNATURES) == declared_symbols()


def test_version_and_argument_errors_without_a_gpu():
    L = pflib.load()
    assert L.pf_version() >= 1000
    need = ctypes.c_size_t()
    assert L.pf_warp_splat_workspace(1, 3, 0, 16, 0, ctypes.byref(need)) == -1          # PF_EINVAL
    assert b'bad dims' in L.pf_last_error()
    assert L.pf_warp_splat_workspace(1, 3, 1024, 9000, 0, ctypes.byref(need)) == -5     # PF_EUNSUPPORTED
    assert L.pf_warp_splat_workspace(2, 3, 1024, 2048, 1, ctypes.byref(need)) == 0
    assert need.value >= 2 * 3 * 1024 * 2048 * 8
    plan = ctypes.c_void_p()
    junk = ctypes.create_string_buffer(b'x' * 128, 128)
    assert L.pf_hardnet_plan_create(junk, 128, 36, 11, ctypes.byref(plan)) == -3          # PF_EBLOB


def test_every_documented_option_name_is_accepted():
    """include/pfhip.h documents the names pf_set_option takes: each is accepted (set to its documented default, so the process
    keeps the library's behaviour), an unknown name is refused - the comment block and hardnet_plan.hip's table cannot drift apart."""
    import re
    text = open(os.path.join(ROOT, 'include', 'pfhip.h')).read()
    block = text[:text.index('int pf_set_option(const char *name, int value);')]
    block = block[block.rindex('/*'):]
    opts = re.findall(r'^ \*   "([a-z0-9_]+)"\s*\(default (\d+)', block, re.M)
    assert len(opts) >= 16 and ('train_forward_s4', '0') in opts and ('wgrad_taps', '1') in opts and ('fuse_pairs', '1') in opts
    L = pflib.load()
    for name, default in opts:
        assert L.pf_set_option(name.encode(), int(default)) == 0, name
    assert L.pf_set_option(b'no_such_option', 1) != 0
