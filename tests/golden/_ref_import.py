"""Import harness for the upstream reference (build container only).

TEST INFRASTRUCTURE.  Used by ``make_golden.py`` to run the reference's own
``PCTransformModel`` / ``BGModel`` / ``data_utils`` from ``/root/reference`` and
capture golden input/output vectors.  Nothing here (and nothing under
``/root/reference``) travels to the GPU box: only the ``.npz`` fixtures do.

Why the stubbing: ``import panoptic_forecasting.models`` pulls in
``cityscapesscripts`` (absent) through ``models/__init__.py:12``; registering
empty parent packages with a ``__path__`` bypasses the package ``__init__``s so
the two hot-path modules import unmodified.

``torch_scatter`` (pinned 2.0.5 in the reference README.md:23) is absent and
not installable offline.  ``scatter_min_standin`` below restates its published
CPU semantics (sequential over the source index, strict ``<`` ⇒ the LOWEST
source index among equal minima wins; empty bins report ``arg = src.size(dim)``
and value 0).  That third-party boundary has no golden vectors in the
reference ⇒ tie-breaking is "parity unpinned" (see DESIGN.md).
"""
import sys
import types

import numpy as np
import torch

REF_ROOT = '/root/reference'


def scatter_min_standin(src, index, dim=-1, out=None, dim_size=None):
    """(values, argmin) with torch_scatter.scatter_min CPU semantics, 2-D, dim=-1."""
    assert dim in (-1, src.dim() - 1) and src.dim() == 2
    b, e_n = src.shape
    if dim_size is None:
        dim_size = int(index.max()) + 1
    vals = torch.zeros(b, dim_size, dtype=src.dtype)
    args = torch.full((b, dim_size), e_n, dtype=torch.long)
    s = src.detach().cpu().numpy()
    ix = index.detach().cpu().numpy()
    for r in range(b):
        e = np.arange(e_n)
        # order by (bin, value, source index): first of each bin is the winner
        order = np.lexsort((e, s[r], ix[r]))
        sb = ix[r][order]
        first = np.ones(e_n, dtype=bool)
        first[1:] = sb[1:] != sb[:-1]
        win = order[first]
        bins = sb[first]
        vals[r, torch.from_numpy(bins)] = torch.from_numpy(s[r][win])
        args[r, torch.from_numpy(bins)] = torch.from_numpy(win)
    return vals, args


def install():
    """Make the reference hot-path modules importable; returns (PCTransformModel, BGModel, data_utils)."""
    pkgs = [
        ('panoptic_forecasting', REF_ROOT + '/panoptic_forecasting'),
        ('panoptic_forecasting.models', REF_ROOT + '/panoptic_forecasting/models'),
        ('panoptic_forecasting.models.bg', REF_ROOT + '/panoptic_forecasting/models/bg'),
        ('panoptic_forecasting.models.pc_transform',
         REF_ROOT + '/panoptic_forecasting/models/pc_transform'),
        ('panoptic_forecasting.data', REF_ROOT + '/panoptic_forecasting/data'),
    ]
    for name, path in pkgs:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    if 'cv2' not in sys.modules:
        sys.modules['cv2'] = types.ModuleType('cv2')
    if not hasattr(np, 'float'):
        np.float = float  # data_utils.py uses the removed alias
    ts = types.ModuleType('torch_scatter')
    ts.scatter_min = scatter_min_standin
    sys.modules['torch_scatter'] = ts
    sys.dont_write_bytecode = True  # /root/reference is read-only
    from panoptic_forecasting.models.pc_transform.pc_transform_model import PCTransformModel
    from panoptic_forecasting.models.bg.bg_model import BGModel
    from panoptic_forecasting.data import data_utils
    return PCTransformModel, BGModel, data_utils
