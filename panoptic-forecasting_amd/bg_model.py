"""``task: bg`` — drop-in for reference ``models/bg/bg_model.py`` + ``models/bg/hardnet.py``.

Same constructor params, same ``predict``/``forward`` returns and the SAME state_dict keys
(``depth_mean``, ``depth_std``, ``model.base.N...``, ``model.conv1x1_up.N...``,
``model.denseBlocksUp.N.layers.M...``, ``model.finalConv``) so the reference's ``bg_model.pt`` loads
unchanged.  The nn.Modules here only *hold* parameters; the arithmetic is ``pf_bg_forward`` in
libpfhip.so (BN folded and weights re-tiled when the plan is built).
"""
import collections.abc
import ctypes

import torch
import torch.nn.functional as F
from torch import nn

from . import hardnet_arch as arch
from . import lib as _lib
from . import packing
from .model_api import BaseModel
from .pc_transform_model import _as_u8


# params['model'] key -> option name of pf_hardnet_plan_set_option
PLAN_OPTIONS = {'split_f16': 'split_f16', 'split_bf16': 'split_f16', 'fuse_pool': 'fuse_pool', 'fuse_upsample': 'fuse_upsample',
                'use_tuned_table': 'use_tuned_table', 'valu_remainder': 'valu_remainder', 'conv_table_batch': 'table_batch',
                'range_guard': 'range_guard', 'fuse_front': 'fuse_front', 'fuse_pairs': 'fuse_pairs'}
PF_STATUS_RANGE, PF_STATUS_RANGE_LOW = 1, 2          # include/pfhip.h
PF_STATUS_ANY = PF_STATUS_RANGE | PF_STATUS_RANGE_LOW
PF_WS_STATUS_BYTES = 2048


class _Pending:
    """One enqueued forward whose status words are on their way to pinned host memory."""
    __slots__ = ('slot', 'event', 'stream', 'args', 'outs', 'done', 'stamps', 'error')


def _stamp(t):
    """(storage address, in-place version) of a tensor argument at enqueue time: a late fp32 re-run reads the caller's input
    tensors again, so it must see the numbers the flagged forward saw (see ``BGModel._resolve``).  Inference tensors
    (``torch.inference_mode()``) carry no version counter - reading ``_version`` raises - and stamp as ``(address, None)``:
    ``run_async`` settles such a forward before it returns instead of trusting a check it cannot make."""
    if not torch.is_tensor(t):
        return None
    return (t.data_ptr(), None if t.is_inference() else t._version)


def _unversioned(stamps):
    return any(s is not None and s[1] is None for s in stamps)


class LazyResult(collections.abc.MutableMapping):
    """``predict``'s result.  The forward behind it was only ENQUEUED (the reference's ``predict`` is asynchronous too,
    bg_model.py:91-102); its status words (include/pfhip.h) follow the outputs to pinned host memory, and the first access
    of a VALUE waits for that one forward and applies ``on_range_overflow`` - a flagged forward is re-run on the fp32 matrix
    instructions INTO THE SAME output tensors before the caller sees them, or raises.  Keys can be listed (``keys()``,
    ``in``, ``len``, iteration) without waiting.

    A Mapping, deliberately NOT a ``dict`` subclass: CPython's fast paths for dict subclasses (``dict(res)``, ``{**res}``,
    ``res | other``, ``update``, pickling) read the underlying storage without calling overridden accessors and would hand out
    tensors whose range check has not happened.  Here every road to a value goes through ``__getitem__``; ``dict(res)`` /
    ``{**res}`` / ``copy()`` / pickling resolve and give a plain ``dict``."""

    __slots__ = ('_data', '_model', '_token')

    def __init__(self, data, model, token):
        self._data, self._model, self._token = dict(data), model, token

    def _resolve(self):
        # the token stays until its forward is settled WITHOUT an error: a failed check (policy 'raise', inputs refilled in
        # place) raises on every access of this result, not only on the first
        if self._token is not None:
            self._model._resolve(self._token)
            self._token = None

    def __getitem__(self, k):
        self._resolve()
        return self._data[k]

    def __setitem__(self, k, v):
        self._data[k] = v

    def __delitem__(self, k):
        self._resolve()        # pop() reads the value on its way out
        del self._data[k]

    def __iter__(self):
        return iter(self._data)

    def __len__(self):
        return len(self._data)

    def __contains__(self, k):
        return k in self._data

    def keys(self):
        return self._data.keys()

    def copy(self):
        self._resolve()
        return dict(self._data)

    def __or__(self, other):
        return {**self.copy(), **other}

    def __ror__(self, other):
        return {**other, **self.copy()}

    def __reduce__(self):      # pickle / copy.copy / copy.deepcopy: a settled plain dict
        return (dict, (self.copy(),))

    def __repr__(self):
        return 'LazyResult(%s%s)' % (list(self._data), ', unchecked' if self._token is not None else '')


class _Node(nn.Module):
    """Anonymous container: gives dotted state_dict paths without any behaviour."""


def _get_or_make(root, path):
    node = root
    for part in path:
        if not hasattr(node, part):
            node.add_module(part, _Node())
        node = getattr(node, part)
    return node


class HardNetParams(nn.Module):
    """Parameter holder whose state_dict mirrors reference ``hardnet`` (hardnet.py:261-327)."""

    def __init__(self, in_ch, n_cls):
        super().__init__()
        self.spec = arch.Spec(in_ch, n_cls)
        for op in self.spec.conv_ops():
            parts = op.name.split('.')
            if op.bn:          # ConvLayer: conv (no bias) + norm   (hardnet.py:16-25)
                holder = _get_or_make(self, parts)
                cin = 3 if op.kind == arch.OP_STEM else op.cin
                holder.add_module('conv', nn.Conv2d(cin, op.cout, op.k, op.stride, op.k // 2, bias=False))
                holder.add_module('norm', nn.BatchNorm2d(op.cout))
            else:              # finalConv: plain conv with bias (hardnet.py:325-327)
                parent = _get_or_make(self, parts[:-1])
                conv = nn.Conv2d(op.cin, op.cout, op.k, op.stride, 0, bias=True)
                nn.init.kaiming_normal_(conv.weight)          # expand_last_layer, hardnet.py:334-339
                parent.add_module(parts[-1], conv)
        # expand_first_layer (hardnet.py:329-332): mean over the RGB input channels, tiled to in_ch
        stem = getattr(self.base, '0').conv
        w = stem.weight.data.mean(1, keepdim=True).expand(-1, in_ch, -1, -1).clone()
        stem.weight = nn.Parameter(w)

    def load_pretrained(self, path):
        """hardnet.py:393-400: ImageNet/Cityscapes FC-HarDNet pickle, ``module.`` prefix stripped."""
        sd = torch.load(path, map_location='cpu')['model_state']
        sd = {k[len('module.'):]: v for k, v in sd.items()}
        own = self.state_dict()
        for k, v in sd.items():
            if k in own and own[k].shape == v.shape:
                own[k].copy_(v)
            elif k == 'base.0.conv.weight':
                own[k].copy_(v.mean(1, keepdim=True).expand_as(own[k]))


class BGModel(BaseModel):

    def __init__(self, params):
        super().__init__()
        self.num_classes = num_classes = params['data']['num_classes']
        self.use_depth_inps = params['model'].get('use_depth_inps')
        self.num_inputs = params['model'].get('num_inputs', 1)
        self.min_depth = params['data'].get('min_depth')
        self.max_depth = params['data'].get('max_depth')
        self.convert2onehot = params['model'].get('convert2onehot')
        self.return_logits = params['model'].get('return_logits', True)
        final_w = params['model'].get('final_w')
        final_h = params['model'].get('final_h')
        self.final_size = (final_h, final_w) if final_w is not None and final_h is not None else None
        in_ch = num_classes
        if self.use_depth_inps:
            depth_norm_params = params['data'].get('depth_norm_params')
            if depth_norm_params is None:
                mean, std = torch.zeros(1), torch.zeros(1)
            else:
                mean, std = depth_norm_params
            self.depth_mean = nn.Parameter(torch.as_tensor(mean, dtype=torch.float32).reshape(1).clone(),
                                           requires_grad=False)
            self.depth_std = nn.Parameter(torch.as_tensor(std, dtype=torch.float32).reshape(1).clone(),
                                          requires_grad=False)
            in_ch += 1
        in_ch *= self.num_inputs
        self.in_ch = in_ch
        self.model = HardNetParams(in_ch, num_classes)
        pretrain = params['model'].get('hardnet', {}).get('pretrain_path')
        if pretrain is not None:
            self.model.load_pretrained(pretrain)
        # execution options of this model's device plan (include/pfhip.h: pf_hardnet_plan_set_option); absent keys keep
        # the library defaults.  ``split_f16: 0`` (``split_bf16`` is the same switch under its round-1 name) = fp32 matrix
        # instructions in every convolution instead of two-term fp16 operands (logits within 1e-4 of the fp32 reference
        # either way, the fp16-pair path is 1.8x faster); ``conv_table_batch: n`` pins the per-layer kernel choice to the one made
        # for batches of n, so a frame's logits do not depend on the size of the batch it arrives in.
        self.plan_options = {c_name: int(params['model'][key]) for key, c_name in PLAN_OPTIONS.items()
                             if params['model'].get(key) is not None}
        # What to do when a forward on the two-term fp16 operand path met activations the pair cannot represent as well as
        # fp32 does: |x| > 65504 (PF_STATUS_RANGE) or a tensor of tiny values, max |x| < 2^-6 (PF_STATUS_RANGE_LOW; include/
        # pfhip.h - the reference's fp32 Conv2d, hardnet.py:16-25, has neither limit):
        #   'rerun' (default) run that forward again on the fp32 matrix instructions (split_f16 = 0) into the same outputs;
        #   'raise'  raise PfError;  'ignore'  return whatever the kernels produced (range_status() still tells).
        # Nothing synchronises in predict(): the status words follow the outputs to pinned host memory (a captured copy +
        # an event) and are checked when the caller first touches a result (LazyResult) or by the next predict() once the
        # event has fired.  Under hipGraph capture the copy is skipped: the kernels OR every forward's status into the
        # workspace's sticky word, which check_range() reads after the replays (bench.py does).
        self.on_range_overflow = params['model'].get('on_range_overflow', 'rerun')
        if self.on_range_overflow not in ('rerun', 'raise', 'ignore'):
            raise ValueError("model.on_range_overflow must be 'rerun', 'raise' or 'ignore'")
        # 'lazy' (default): predict() only enqueues and the check happens when a result is first touched - the caller must leave
        # the INPUT tensors of a forward unchanged until then (a flagged forward is re-run from them; an in-place refill is
        # detected through the tensors' version counters and raises).  'sync': predict() waits for its own forward's status
        # before returning (one stream synchronisation per call, what round 3 did) - for callers that refill static input
        # buffers or write them behind torch's back.
        self.range_check = params['model'].get('range_check', 'lazy')
        if self.range_check not in ('lazy', 'sync'):
            raise ValueError("model.range_check must be 'lazy' or 'sync'")
        self.range_reruns = 0
        self._pending = []          # enqueued forwards not checked yet, oldest first
        self._pinned = None         # [ring, 2] int32 pinned: (status word, sticky word) per in-flight forward
        self._ring_next = 0
        self._plan = None
        self._ws = None
        self._norm = None
        self._trainer = None
        self._params_for_trainer = {'data': dict(params['data']), 'model': dict(params['model']),
                                    'training': dict(params.get('training') or {})}

    # ---- plan lifecycle -------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        res = super().load_state_dict(state_dict, strict)
        self.invalidate()
        return res

    def invalidate(self):
        """Drop the device plan (call after mutating parameters by hand)."""
        if self._pending:
            self._drain()
        if self._plan is not None:
            _lib.load().pf_hardnet_plan_destroy(self._plan)
        self._plan = None
        self._norm = None

    def __del__(self):
        try:
            self._pending = []      # nobody can look at those results any more; never wait on the device in a destructor
            self.invalidate()
        except Exception:
            pass

    def _get_plan(self):
        if self._plan is None:
            L = _lib.load()
            blob = packing.pack_blob(self.state_dict(), self.in_ch, self.num_classes)
            buf = ctypes.create_string_buffer(blob, len(blob))
            plan = ctypes.c_void_p()
            _lib.check(L.pf_hardnet_plan_create(buf, len(blob), self.in_ch, self.num_classes, ctypes.byref(plan)),
                       'pf_hardnet_plan_create')
            self._plan = plan
            for name, value in self.plan_options.items():
                _lib.check(L.pf_hardnet_plan_set_option(plan, name.encode(), value), 'pf_hardnet_plan_set_option')
            if self.use_depth_inps:
                self._norm = (float(self.depth_mean.item()), float(self.depth_std.item()))
        return self._plan

    def _workspace(self, b, h, w, device):
        L = _lib.load()
        need = ctypes.c_size_t()
        _lib.check(L.pf_hardnet_workspace(self._get_plan(), b, h, w, ctypes.byref(need)), 'pf_hardnet_workspace')
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != device:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.PfError('bg forward: the workspace must exist before stream capture (run one forward first)')
            self._drain()
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=device)
            self._ws[:PF_WS_STATUS_BYTES].zero_()     # status block: the sticky word is only ever cleared by the host
        return self._ws

    def _guarded(self):
        return not (self.on_range_overflow == 'ignore' or self.plan_options.get('split_f16', 1) == 0
                    or self.plan_options.get('range_guard', 1) == 0)

    def range_status(self):
        """Status word of the LAST forward in this model's workspace (synchronises): PF_STATUS_RANGE / PF_STATUS_RANGE_LOW
        set = that forward met values outside what the fp16-pair path represents and its outputs must not be used."""
        if self._ws is None:
            return 0
        return int(self._ws[:4].view(torch.int32).item())

    def range_status_sticky(self, clear=False):
        """OR of the status words of every forward through this workspace since the last clear (synchronises) - what to
        read after hipGraph replays or an eager loop of several forwards."""
        if self._ws is None:
            return 0
        w = self._ws[4:8].view(torch.int32)
        v = int(w.item())
        if clear:
            w.zero_()
        return v

    def check_range(self, clear=True):
        """After hipGraph replays (where predict() can neither wait nor re-run): raises PfError if any forward since the last
        clear was flagged, unless ``on_range_overflow == 'ignore'``.  Returns the sticky status."""
        self._drain()
        v = self.range_status_sticky(clear)
        if (v & PF_STATUS_ANY) and self._guarded():
            raise _lib.PfError('bg forward: status %d (PF_STATUS_RANGE = 1: |activation| > 65504, PF_STATUS_RANGE_LOW = 2: a tensor of '
                               'tiny values) on the two-term fp16 operand path inside a captured graph: the outputs of the flagged '
                               'replays are not usable; run eagerly (re-run on fp32) or with model.split_f16 = 0' % v)
        return v

    def range_maxima(self):
        """{op name: max |stored value|} of the last forward (diagnostic of the range guard; synchronises)."""
        L, plan = _lib.load(), self._get_plan()
        names = [op.name for op in self.model.spec.ops]
        buf = (ctypes.c_float * len(names))()
        n = ctypes.c_int()
        _lib.check(L.pf_hardnet_range_maxima(plan, self._ws.data_ptr(), buf, len(names), ctypes.byref(n), _lib.stream_ptr()),
                   'pf_hardnet_range_maxima')
        return {names[i]: float(buf[i]) for i in range(n.value)}

    # ---- device forward ---------------------------------------------------------------------
    _RING = 32

    def _resolve(self, token, owner=True):
        """Wait for ONE forward's status words and apply the policy (see LazyResult).  A failed check is RECORDED on the token
        and raised to whoever owns the result, on every access; `owner=False` (the housekeeping of a later predict: ``_poll``,
        the ring wrapping) never raises - frame i + 1's predict must not die of frame i's flag."""
        if token.done:
            if token.error is not None and owner:
                raise token.error
            return
        if not token.event.query():      # (a fired event needs no wait: _poll() settles forwards without any synchronising call)
            token.event.synchronize()
        token.done = True
        if token in self._pending:
            self._pending.remove(token)
        status = int(self._pinned[token.slot, 0])
        args, outs, stamps = token.args, token.outs, token.stamps
        token.args = token.outs = token.stamps = None
        if not (status & PF_STATUS_ANY):
            return
        if self.on_range_overflow == 'raise':
            token.error = _lib.PfError('bg forward: status %d on the two-term fp16 operand path (PF_STATUS_RANGE = 1: an activation '
                                       'exceeded 65504; PF_STATUS_RANGE_LOW = 2: a tensor of tiny values, max below 2^-6); run with '
                                       "model.split_f16 = 0 or on_range_overflow = 'rerun'" % status)
        # The re-run reads the caller's input tensors NOW, not when predict() was called.  If they were refilled in place since
        # (a static input buffer with `static.copy_(batch)`, a pinned staging loop), re-running would silently put a newer
        # frame's result into the older frame's outputs: refuse instead.  (Writes that bypass torch's version counter -
        # `x.data.copy_`, a foreign kernel - are not seen: such callers use model.range_check = 'sync'.)
        elif stamps is not None and stamps != tuple(_stamp(t) for t in args[:3]):
            token.error = _lib.PfError('bg forward: status %d on the two-term fp16 operand path, and the input tensors of that forward were '
                                       'modified in place before its result was first touched - the fp32 re-run cannot see the original '
                                       "frame.  Keep a forward's inputs unchanged until its result has been read (or model.settle()), or "
                                       "build the model with model.range_check = 'sync'" % status)
        if token.error is not None:
            if owner:
                raise token.error
            return
        L, plan = _lib.load(), self._get_plan()
        self.range_reruns += 1
        prior = self.plan_options.get('split_f16', 1)
        _lib.check(L.pf_hardnet_plan_set_option(plan, b'split_f16', 0), 'pf_hardnet_plan_set_option')
        try:
            # on the stream the forward ran on (the workspace belongs to that stream's order: a later forward may be using it
            # right now); whoever touches the result on another stream is made to wait for the re-run
            here = torch.cuda.current_stream()
            with torch.cuda.stream(token.stream):
                self._run_once(*args, outs=outs)
            if here != token.stream:
                here.wait_stream(token.stream)
        finally:
            _lib.check(L.pf_hardnet_plan_set_option(plan, b'split_f16', prior), 'pf_hardnet_plan_set_option')

    def _poll(self):
        """Non-blocking: settle every enqueued forward whose status has arrived (predict i checks forward i - 1)."""
        while self._pending and self._pending[0].event.query():
            self._resolve(self._pending[0], owner=False)

    def _drain(self):
        while self._pending:
            self._resolve(self._pending[0], owner=False)

    def settle(self):
        """Wait for and check every forward enqueued so far (what the first access of each LazyResult would do)."""
        self._drain()

    def run_async(self, inps, depths, depth_masks, want_logits=True, want_orig=True, hop_flags=0, seg_dtype=torch.int64,
                  own_inputs=False):
        """Enqueue ``pf_bg_forward`` / ``pf_hardnet_forward_dense`` -> ((seg, logits|None, orig|None), token).  ``token`` is
        None when nothing has to be checked (policy 'ignore', fp32-only plan, stream capture); else pass it to
        ``_resolve`` (or wrap the outputs in a LazyResult) before using them.  ``own_inputs``: the tensors were made by the caller
        of this method for this forward alone (``task: bg_forecast``'s warped frames) - no refill check is needed or made."""
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:                      # (events cannot be queried while a stream captures)
            self._poll()
        args = (inps, depths, depth_masks, want_logits, want_orig, hop_flags, seg_dtype)
        outs = self._run_once(*args)
        if not self._guarded() or capturing:
            return outs, None
        if self._pinned is None:
            self._pinned = torch.zeros((self._RING, 2), dtype=torch.int32).pin_memory()
        slot = self._ring_next
        self._ring_next = (slot + 1) % self._RING
        for t in list(self._pending):          # the ring wrapped onto a forward nobody has looked at yet
            if t.slot == slot:
                self._resolve(t, owner=False)
        self._pinned[slot].copy_(self._ws[:8].view(torch.int32), non_blocking=True)
        token = _Pending()
        token.slot, token.args, token.outs, token.done, token.error = slot, args, outs, False, None
        token.stamps = None if own_inputs else tuple(_stamp(t) for t in args[:3])
        token.stream = torch.cuda.current_stream()
        token.event = torch.cuda.Event()
        token.event.record(token.stream)
        self._pending.append(token)
        if token.stamps is not None and _unversioned(token.stamps):
            # inference tensors: an in-place refill could not be detected later - settle this forward now (one synchronisation)
            self._resolve(token)
            return outs, None
        return outs, token

    def run(self, inps, depths, depth_masks, want_logits=True, want_orig=True, hop_flags=0, seg_dtype=torch.int64):
        """``run_async`` + the range check of the fp16-pair path (``on_range_overflow``): (seg, logits|None, orig|None)."""
        outs, token = self.run_async(inps, depths, depth_masks, want_logits, want_orig, hop_flags, seg_dtype)
        if token is not None:
            self._resolve(token)
        return outs

    def _run_once(self, inps, depths, depth_masks, want_logits, want_orig, hop_flags, seg_dtype, outs=None):
        L = _lib.load()
        plan = self._get_plan()
        fused = bool(self.convert2onehot and self.use_depth_inps and inps.dim() == 4)
        if fused:
            b, t, h, w = inps.shape
        else:
            x = self._dense_input(inps, depths, depth_masks)
            b, _, h, w = x.shape
        dev = inps.device
        oh, ow = self.final_size if self.final_size is not None else (h, w)
        ws = self._workspace(b, h, w, dev)
        if outs is not None:
            seg, logits, orig = outs
        else:
            seg = torch.empty((b, oh, ow), dtype=seg_dtype, device=dev)
            logits = torch.empty((b, self.num_classes, oh, ow), dtype=torch.float32, device=dev) if want_logits else None
            orig = None
            if want_orig:
                vo, vc, vh, vw = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                _lib.check(L.pf_hardnet_tensor_view(plan, b'finalConv', b, h, w, ctypes.byref(vo), ctypes.byref(vc),
                                                    ctypes.byref(vh), ctypes.byref(vw)), 'pf_hardnet_tensor_view')
                orig = torch.empty((b, vc.value, vh.value, vw.value), dtype=torch.float32, device=dev)
        ptr = lambda t_: t_.data_ptr() if t_ is not None else None
        i64 = int(seg_dtype == torch.int64)
        if fused:
            if inps.dtype not in (torch.uint8, torch.int64):
                inps = inps.long()
            inps = _lib.require_cuda(inps.contiguous(), 'seg')
            depths = _lib.require_cuda(depths.float().contiguous(), 'depth')
            mask = None
            if not (hop_flags & 2):
                mask = _lib.require_cuda(_as_u8(depth_masks), 'depth_mask')
            mean, std = self._norm
            rc = L.pf_bg_forward(plan, inps.data_ptr(), int(inps.dtype == torch.int64), depths.data_ptr(), ptr(mask),
                                 mean, std, hop_flags, float(self.min_depth or 0.0), float(self.max_depth or 0.0),
                                 b, t, h, w, oh, ow, seg.data_ptr(), i64, ptr(logits), ptr(orig),
                                 ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, 'pf_bg_forward')
        else:
            x = _lib.require_cuda(x.contiguous(), 'input')
            rc = L.pf_hardnet_forward_dense(plan, x.data_ptr(), b, h, w, oh, ow, seg.data_ptr(), i64, ptr(logits),
                                            ptr(orig), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, 'pf_hardnet_forward_dense')
        return seg, logits, orig

    def _dense_input(self, inps, depths, depth_masks):
        """bg_model.py:61-69 for the non-default configurations (convert2onehot False, or no depth channels): builds
        [B, in_ch, H, W] f32 with ONE device call (``pf_bg_dense_input``; ATen glue until round 6)."""
        L = _lib.load()
        if self.convert2onehot:
            b, t, h, w = inps.shape
            if inps.dtype not in (torch.uint8, torch.int64):
                inps = inps.long()
            frames, kind, c = _lib.require_cuda(inps.contiguous(), 'seg'), int(inps.dtype == torch.int64), self.num_classes
        else:
            b, t, c, h, w = inps.shape
            frames, kind = _lib.require_cuda(inps.float().contiguous(), 'seg'), 2
        dptr = mptr = None
        mean = std = 0.0
        if self.use_depth_inps:
            depths = _lib.require_cuda(depths.float().contiguous(), 'depth')
            mask = _lib.require_cuda(_as_u8(depth_masks), 'depth_mask')
            dptr, mptr = depths.data_ptr(), mask.data_ptr()
            self._get_plan()
            mean, std = self._norm
        x = torch.empty((b, t * c + (t if self.use_depth_inps else 0), h, w), dtype=torch.float32, device=inps.device)
        _lib.check(L.pf_bg_dense_input(frames.data_ptr(), kind, c, dptr, mptr, mean, std, b, t, h, w, x.data_ptr(), _lib.stream_ptr()),
                   'pf_bg_dense_input')
        return x

    # ---- reference surface ------------------------------------------------------------------
    def forward(self, inps, depths, depth_masks, return_orig_size=False):
        _, logits, orig = self.run(inps, depths, depth_masks, True, return_orig_size)
        return (logits, orig) if return_orig_size else logits

    def loss(self, inputs, labels):
        """Reference ``BGModel.loss`` (bg_model.py:73-89): ``{'loss', 'accuracy'}`` for a batch.

        ``model.train()`` with autograd enabled (the reference's training loop, train.py:186-201): forward with
        batch-statistics BatchNorm, cross entropy and the whole backward pass run as ONE device call
        (``pf_train_forward_backward``); the returned loss carries an autograd node that delivers the parameter
        gradients on ``backward()`` (``bg_train.TrainStepFunction``), so ``clip_grad_norm_`` / ``opt.step()`` / DDP work
        unchanged.  Under ``torch.no_grad()`` a model in training mode still runs that call (batch statistics, running-stat
        update - what ``nn.BatchNorm2d`` does in the reference) and returns the loss without an autograd node.  In ``eval()``
        mode the validation form: eval-mode network (folded BN) + ``pf_seg_loss`` (upsample + cross entropy + accuracy fused;
        the full-resolution logits are never written)."""
        if self.training:
            return self._train_loss(inputs, labels)
        with torch.no_grad():
            return self._eval_loss(inputs, labels)

    def _get_trainer(self):
        from . import bg_train
        if self._trainer is None:
            self._trainer = bg_train.BGTrainer(self._params_for_trainer, device='cuda')
        if not self._trainer.is_adopted():
            self._trainer.adopt(self)
        return self._trainer

    def _train_loss(self, inputs, labels):
        from . import bg_train
        tr = self._get_trainer()
        self.invalidate()            # the folded-BN inference plan goes stale as soon as the parameters move
        loss, acc = bg_train.TrainStepFunction.apply(tr, inputs, labels, *tr._adopted)
        torch._foreach_add_(self._bn_counters(), 1)          # nn.BatchNorm2d.num_batches_tracked
        return {'loss': loss, 'accuracy': acc}

    def _bn_counters(self):
        return [b for k, b in self.named_buffers() if k.endswith('num_batches_tracked')]

    def _eval_loss(self, inputs, labels):
        L = _lib.load()
        seg_labels = labels['seg']
        _, _, orig = self.run(inputs['seg'], inputs.get('depth'), inputs.get('depth_mask'), want_logits=False, want_orig=True)
        if seg_labels.dtype not in (torch.uint8, torch.int64):
            seg_labels = seg_labels.long()
        seg_labels = _lib.require_cuda(seg_labels.contiguous(), 'labels')
        b, c, hin, win = orig.shape
        oh, ow = seg_labels.shape[-2], seg_labels.shape[-1]
        need = ctypes.c_size_t()
        _lib.check(L.pf_seg_loss_workspace(b, oh, ow, ctypes.byref(need)), 'pf_seg_loss_workspace')
        ws = torch.empty(need.value, dtype=torch.uint8, device=orig.device)
        out3 = torch.empty(3, dtype=torch.float64, device=orig.device)
        _lib.check(L.pf_seg_loss(orig.data_ptr(), b, c, hin, win, seg_labels.data_ptr(), int(seg_labels.dtype == torch.int64), oh, ow,
                                 255, out3.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), 'pf_seg_loss')
        return {'loss': (out3[0] / out3[1]).float(), 'accuracy': (out3[2] / out3[1]).float()}

    @torch.no_grad()
    def predict(self, inputs, labels=None):
        """Reference ``BGModel.predict`` (bg_model.py:91-102): ``{'seg', 'orig_size_logits'[, 'logits']}``.  Only ENQUEUES (see
        ``LazyResult``).  Lifetime rule of the default ``range_check = 'lazy'``: leave ``inputs['seg' / 'depth' / 'depth_mask']``
        unchanged until a value of the result has been read (or ``settle()``): a forward flagged by the range guard is re-run
        from them at that point.  An in-place refill in between raises PfError instead of re-running on the wrong frame;
        ``model.range_check = 'sync'`` checks before returning."""
        (seg, logits, orig), token = self.run_async(inputs['seg'], inputs.get('depth'), inputs.get('depth_mask'),
                                                    want_logits=self.return_logits, want_orig=True)
        out = {'seg': seg, 'orig_size_logits': orig}
        if logits is not None:
            out['logits'] = logits
        if token is not None and self.range_check == 'sync':
            self._resolve(token)
            token = None
        return LazyResult(out, self, token)
