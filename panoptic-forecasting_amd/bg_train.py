"""bg training step on the device (scope row f4) — host side of ``pf_train_*`` / ``pf_sgd_step`` (include/pfhip.h).

What the reference does per batch for ``task: bg`` (``training/train.py:185-222`` around ``BGModel.loss``,
``models/bg/bg_model.py:73-89``): forward in training mode (batch-statistics BatchNorm), bilinear upsample to the label size,
cross entropy with ``ignore_index=255``, backward, ``clip_grad_norm_`` / ``clip_grad_value_``, SGD(momentum, weight decay)
— here as two C-ABI calls on ONE flat parameter array, with one RCCL all-reduce of the flat gradient in between when several
ranks train (the reference wraps the model in DistributedDataParallel, ``train.py:96-103``).

``BGTrainer`` owns the flat arrays (``theta``, ``grad``, momentum) and maps them to and from the reference's
``state_dict`` keys, so checkpoints stay interchangeable with ``bg_model.pt``.  ``BGTrainer.adopt(module)`` re-seats the
parameters and BatchNorm buffers of a ``BGModel`` as views into ``theta`` (no copies between "the model" and "the device
arrays" afterwards), and ``TrainStepFunction`` plugs the fused step into autograd: ``BGModel.loss`` in training mode returns
a loss whose ``backward()`` delivers the gradient of every ``nn.Parameter`` — which lets the reference's unmodified loop
(``loss.backward(); clip_grad_norm_; opt.step()``, ``train.py:201-210``) and DistributedDataParallel drive it.
"""
import ctypes

import torch
import torch.distributed as dist

from . import hardnet_arch as arch
from . import lib as _lib
from . import packing

BN_MOMENTUM, BN_EPS = 0.1, 1e-5       # nn.BatchNorm2d defaults, as constructed at hardnet.py:21


def param_layout(spec, prefix='model.'):
    """[(state_dict key, offset, shape, trainable)] in the order of the C library's flat array (train_plan.hip)."""
    out, cur = [], 0
    for op in spec.conv_ops():
        cin = op.cin
        p = prefix + op.name
        wkey = p + ('.conv.weight' if op.bn else '.weight')
        out.append((wkey, cur, (op.cout, cin, op.k, op.k), True))
        cur += op.cout * cin * op.k * op.k
        if op.bn:
            for suffix, trainable in (('.norm.weight', True), ('.norm.bias', True), ('.norm.running_mean', False),
                                      ('.norm.running_var', False)):
                out.append((p + suffix, cur, (op.cout,), trainable))
                cur += op.cout
        else:
            out.append((p + '.bias', cur, (op.cout,), True))
            cur += op.cout
    return out, cur


class BGTrainer:
    """Flat-array trainer for the bg network.  ``params`` is the reference's params dict (``data.num_classes``,
    ``model.num_inputs`` / ``use_depth_inps``, ``training.lr`` / ``mom`` / ``wd`` / ``clip_grad`` / ``clip_grad_norm`` /
    ``accumulate_steps``)."""

    def __init__(self, params, device='cuda'):
        self.n_cls = params['data']['num_classes']
        self.T = params['model'].get('num_inputs', 1)
        if not (params['model'].get('use_depth_inps') and params['model'].get('convert2onehot')):
            raise _lib.PfError('BGTrainer: built for the one-hot + depth input of configs/bg/bg_train.yaml')
        self.in_ch = self.T * (self.n_cls + 1)
        tr = params.get('training', {})
        self.lr, self.mom, self.wd = float(tr.get('lr', 2e-3)), float(tr.get('mom', 0.)), float(tr.get('wd', 0.))
        self.clip_value = float(tr.get('clip_grad') or 0.)
        self.clip_norm = 0. if tr.get('clip_grad') is not None else float(tr.get('clip_grad_norm') or 0.)   # train.py:205-208
        self.accumulate_steps = int(tr.get('accumulate_steps', 1))
        # Two launch forms (same bits, tests/test_gpu_train.py):
        #  * default - eager launches with the weight gradients (leaves of the backward pass) on the training plan's own
        #    lower-priority stream (``training.weight_gradient_stream``, library option "train_side_stream"): they fill the
        #    GPU beside the small-grid kernels of the bn-backward -> input-gradient chain; 17.7 ms per step at batch 8 of
        #    800x800 against 19.6 ms on one stream;
        #  * ``training.use_hip_graph`` - the call is captured into a hipGraph the second time a configuration (shapes, dtypes,
        #    accumulate / loss-scale / running-stat flags) is seen and replayed from then on, on static copies of the inputs:
        #    no host launch cost (a busy host, many ranks per socket), 19.6 ms.  A captured step stays on one stream: every
        #    cross-stream edge of a hipGraph costs a barrier packet on this runtime (30.5 ms with the fork per layer).
        self.use_graph = bool(tr.get('use_hip_graph', False))
        self.side_stream = bool(tr.get('weight_gradient_stream', not self.use_graph))
        # ``training.autotune``: measure the workgroup shape of every forward / backward-data convolution on first sight instead of
        # taking the cost model's (include/pfhip.h: pf_train_autotune) - faster, but the shapes then depend on timing
        self.autotune = bool(tr.get('autotune', False))
        self._graphs = {}
        dn = params['data'].get('depth_norm_params')
        self.depth_mean, self.depth_std = (float(dn[0]), float(dn[1])) if dn is not None else (0., 0.)
        self.device = torch.device(device)
        self.spec = arch.Spec(self.in_ch, self.n_cls)
        self.layout, self.n = param_layout(self.spec)
        L = _lib.load()
        # the op table travels in the same blob format as the inference plan; its weight section is not used here
        blob = packing.pack_blob(None, self.in_ch, self.n_cls, spec=self.spec,
                                 params={op.name: (torch.zeros(op.cout, op.cin, op.k, op.k), torch.zeros(op.cout))
                                         for op in self.spec.conv_ops()})
        self._buf = ctypes.create_string_buffer(blob, len(blob))
        self._t = ctypes.c_void_p()
        _lib.check(L.pf_set_option(b'train_side_stream', int(self.side_stream)), 'pf_set_option')   # read by pf_train_create
        # training.forward_packed_pairs (default: the library's, off): the forward conv + BatchNorm layers on the inference path's
        # fp16-pair kernels with blocked sums - 1.5 % faster steps, the forward pass as close to float64 as the fp32 step's
        fwd_s4 = params.get('training', {}).get('forward_packed_pairs')
        if fwd_s4 is not None:
            _lib.check(L.pf_set_option(b'train_forward_s4', int(bool(fwd_s4))), 'pf_set_option')      # read by pf_train_create
        try:
            _lib.check(L.pf_train_create(self._buf, len(blob), self.in_ch, self.n_cls, ctypes.byref(self._t)), 'pf_train_create')
        finally:
            L.pf_set_option(b'train_side_stream', 1)
            if fwd_s4 is not None:
                L.pf_set_option(b'train_forward_s4', 0)
        if self.autotune:
            _lib.check(L.pf_train_autotune(self._t, 1), 'pf_train_autotune')
        n = ctypes.c_size_t()
        _lib.check(L.pf_train_param_count(self._t, ctypes.byref(n)), 'pf_train_param_count')
        if n.value != self.n:
            raise _lib.PfError('parameter layout mismatch: library %d, host %d' % (n.value, self.n))
        self.theta = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros_like(self.theta)
        self.momentum_buf = torch.zeros_like(self.theta)
        mask = torch.zeros(self.n, dtype=torch.uint8)
        for _, off, shape, trainable in self.layout:
            if trainable:
                mask[off:off + int(torch.tensor(shape).prod())] = 1
        self.trainable = mask.to(self.device)
        self.steps = 0
        self._micro = 0
        self._ws = None
        self._trainable_layout = None
        self._adopted = None
        need = ctypes.c_size_t()
        _lib.check(L.pf_sgd_workspace(ctypes.byref(need)), 'pf_sgd_workspace')
        self._sgd_ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        self.out3 = torch.zeros(3, dtype=torch.float64, device=self.device)

    def __del__(self):
        try:
            if self._t:
                _lib.load().pf_train_destroy(self._t)
        except Exception:
            pass

    # ---- checkpoints (reference key set) --------------------------------------------------------
    def load_state_dict(self, sd):
        host = torch.empty(self.n, dtype=torch.float32)
        for key, off, shape, _ in self.layout:
            t = sd[key].detach().float().cpu()
            if tuple(t.shape) != tuple(shape):
                raise _lib.PfError('%s: shape %s, expected %s' % (key, tuple(t.shape), shape))
            host[off:off + t.numel()] = t.reshape(-1)
        self.theta.copy_(host)
        if 'depth_mean' in sd:
            self.depth_mean, self.depth_std = float(sd['depth_mean']), float(sd['depth_std'])

    def state_dict(self):
        host = self.theta.detach().cpu()
        sd = {'depth_mean': torch.tensor([self.depth_mean]), 'depth_std': torch.tensor([self.depth_std])}
        for key, off, shape, _ in self.layout:
            n = 1
            for s in shape:
                n *= s
            sd[key] = host[off:off + n].reshape(shape).clone()
            if key.endswith('.norm.running_var'):
                sd[key[:-len('running_var')] + 'num_batches_tracked'] = torch.tensor(self.steps, dtype=torch.long)
        return sd

    # ---- optimizer state in torch.optim.SGD's own format ------------------------------------------
    def optimizer_state_dict(self):
        """What ``opt.state_dict()`` of the reference's optimizer holds at this point (train.py:130-136: ``torch.optim.SGD`` over
        the parameters with ``requires_grad``, in ``model.parameters()`` order; saved at train.py:285): the reference's
        ``opt.load_state_dict(train_params['optimizer'])`` (train.py:148) accepts it.  Built by a real ``torch.optim.SGD`` on
        views of the momentum arena, so the dictionary has exactly the fields of the installed torch."""
        host = self.momentum_buf.detach().cpu()
        by_key = {key: (off, shape, n) for key, off, shape, n in self.trainable_layout()}
        params, bufs = [], []
        for key in reference_parameter_order(list(by_key)):
            off, shape, n = by_key[key]
            params.append(torch.nn.Parameter(torch.empty(shape)))
            bufs.append(host[off:off + n].reshape(shape).clone())
        opt = torch.optim.SGD(params, lr=self.lr, momentum=self.mom, weight_decay=self.wd)
        if self.steps > 0 and self.mom != 0.:
            for p, b in zip(params, bufs):
                opt.state[p]['momentum_buffer'] = b
        return opt.state_dict()

    def load_optimizer_state_dict(self, st):
        """Accepts ``torch.optim.SGD.state_dict()`` of the reference loop (or of ``optimizer_state_dict``) and the flat
        ``{'momentum_buffer': tensor}`` form this driver wrote in round 2."""
        if 'momentum_buffer' in st and 'state' not in st:
            self.momentum_buf.copy_(st['momentum_buffer'])
            return
        by_key = {key: (off, shape, n) for key, off, shape, n in self.trainable_layout()}
        order = reference_parameter_order(list(by_key))
        ids = [i for g in st['param_groups'] for i in g['params']]
        if len(ids) != len(order):
            raise _lib.PfError('optimizer state has %d parameters, the bg network has %d trainable tensors' % (len(ids), len(order)))
        host = torch.zeros(self.n, dtype=torch.float32)
        for i, key in zip(ids, order):
            buf = st['state'].get(i, {}).get('momentum_buffer')
            if buf is None:
                continue
            off, shape, n = by_key[key]
            if tuple(buf.shape) != tuple(shape):
                raise _lib.PfError('optimizer state %d (%s): shape %s, expected %s' % (i, key, tuple(buf.shape), shape))
            host[off:off + n] = buf.detach().float().cpu().reshape(-1)
        self.momentum_buf.copy_(host)

    def named_grads(self):
        """{state_dict key: gradient tensor (view into the flat array)} for the trainable entries."""
        out = {}
        for key, off, shape, trainable in self.layout:
            if trainable:
                n = 1
                for s in shape:
                    n *= s
                out[key] = self.grad[off:off + n].view(shape)
        return out

    def trainable_layout(self):
        """[(key, offset, shape, numel)] of the trainable entries, layout order."""
        if self._trainable_layout is None:
            out = []
            for key, off, shape, trainable in self.layout:
                if trainable:
                    n = 1
                    for d in shape:
                        n *= d
                    out.append((key, off, tuple(shape), n))
            self._trainable_layout = out
        return self._trainable_layout

    # ---- parameter adoption ---------------------------------------------------------------------
    def adopt(self, module):
        """Re-seat ``module``'s parameters and float buffers (reference key set) as views into ``theta``: afterwards the
        module and the device arrays are the same memory (``load_state_dict``, ``opt.step()`` and the running-statistics
        update of the kernels all act on it).  Returns the trainable parameters in ``trainable_layout()`` order."""
        self.load_state_dict(module.state_dict())
        named = dict(module.named_parameters())
        named.update({k: v for k, v in module.named_buffers()})
        params = []
        with torch.no_grad():
            for key, off, shape, trainable in self.layout:
                n = 1
                for d in shape:
                    n *= d
                t = named[key]
                t.data = self.theta[off:off + n].view(shape)
                if trainable:
                    params.append(t)
        self._adopted = params
        return params

    def is_adopted(self):
        """True while the adopted parameters still alias ``theta`` (``module.to(...)`` / ``.cuda()`` re-allocates them)."""
        if not self._adopted:
            return False
        base = self.theta.data_ptr()
        return all(p.data_ptr() == base + 4 * off for p, (_, off, _, _) in zip(self._adopted, self.trainable_layout()))

    MAX_GRAPHS = 8      # captured step configurations kept (see forward_backward)

    # ---- one micro-batch: forward + loss + backward ---------------------------------------------
    def _workspace(self, b, h, w, oh, ow):
        need = ctypes.c_size_t()
        _lib.check(_lib.load().pf_train_workspace(self._t, b, h, w, oh, ow, ctypes.byref(need)), 'pf_train_workspace')
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
            self._graphs = {}          # captured graphs hold the old workspace's address
        return self._ws

    def forward_backward(self, inputs, labels, accumulate=False, update_running_stats=True, loss_scale=1.0):
        """Gradients of mean-CE land in ``self.grad``; returns {'loss', 'accuracy'} as 0-d device tensors (no host sync)."""
        L = _lib.load()
        seg, depth, mask = inputs['seg'], inputs['depth'], inputs['depth_mask']
        lab = labels['seg']
        if seg.dtype not in (torch.uint8, torch.int64):
            seg = seg.long()
        if lab.dtype not in (torch.uint8, torch.int64):
            lab = lab.long()
        seg = _lib.require_cuda(seg.contiguous(), 'seg')
        depth = _lib.require_cuda(depth.float().contiguous(), 'depth')
        mask = _lib.require_cuda((mask if mask.dtype == torch.bool else mask != 0).contiguous().view(torch.uint8), 'depth_mask')
        lab = _lib.require_cuda(lab.contiguous(), 'labels')
        b, t, h, w = seg.shape
        oh, ow = lab.shape[-2], lab.shape[-1]
        ws = self._workspace(b, h, w, oh, ow)

        def enqueue(seg_, depth_, mask_, lab_):
            rc = L.pf_train_forward_backward(self._t, self.theta.data_ptr(), self.grad.data_ptr(), int(accumulate), seg_.data_ptr(),
                                             int(seg_.dtype == torch.int64), depth_.data_ptr(), mask_.data_ptr(), self.depth_mean,
                                             self.depth_std, t, None, b, h, w, lab_.data_ptr(), int(lab_.dtype == torch.int64), oh, ow, 255,
                                             float(loss_scale), BN_MOMENTUM, BN_EPS, int(update_running_stats), self.out3.data_ptr(),
                                             ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, 'pf_train_forward_backward')

        graph = None
        if self.use_graph and not torch.cuda.is_current_stream_capturing():
            key = (tuple(seg.shape), seg.dtype, tuple(lab.shape), lab.dtype, bool(accumulate), bool(update_running_stats),
                   float(loss_scale), self.depth_mean, self.depth_std)
            ent = self._graphs.get(key)
            if ent is None:
                # first batch of this configuration: eagerly (the library's one-time kernel-attribute calls are not capturable).
                # The cache is bounded: a key holds float(loss_scale), so a caller with dynamic loss scaling would otherwise
                # capture (and keep) one hipGraph + static input copies per distinct scale - the oldest entry goes first
                while len(self._graphs) >= self.MAX_GRAPHS:
                    self._graphs.pop(next(iter(self._graphs)))
                self._graphs[key] = 'seen'
            else:
                if ent == 'seen':
                    static = tuple(torch.empty_like(x) for x in (seg, depth, mask, lab))
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        enqueue(*static)      # capture only: nothing runs, theta / grad / running statistics are untouched
                    ent = self._graphs[key] = (g, static)
                graph, static = ent
                for dst, src in zip(static, (seg, depth, mask, lab)):
                    dst.copy_(src)
        if graph is not None:
            graph.replay()
        else:
            enqueue(seg, depth, mask, lab)
        return {'loss': (self.out3[0] / self.out3[1]).float(), 'accuracy': (self.out3[2] / self.out3[1]).float()}

    PATH_STATS = ('table_shapes', 'model_shapes', 'autotuned_shapes', 'padded_copy_convs', 'padded_output_layers',
                  'single_backward_data_convs', 'generic_kernel_launches', 'packed_pair_forward_convs')

    def path_stats(self):
        """Which code paths the last ``forward_backward`` took (include/pfhip.h: pf_train_path_stats) - counts of convolutions by
        where their workgroup shape came from, of odd-width layers on the padded forms, of generic-kernel fallbacks."""
        buf, n = (ctypes.c_int * 8)(), ctypes.c_int()
        _lib.check(_lib.load().pf_train_path_stats(self._t, buf, 8, ctypes.byref(n)), 'pf_train_path_stats')
        return {k: int(buf[i]) for i, k in enumerate(self.PATH_STATS[:n.value])}

    def all_reduce_grads(self):
        """Data-parallel exchange: ONE all-reduce of the flat gradient (RCCL over xGMI; DDP-style average)."""
        from . import dist as pfdist
        pfdist.all_reduce_mean_(self.grad)

    def optimizer_step(self, lr=None):
        L = _lib.load()
        rc = L.pf_sgd_step(self.theta.data_ptr(), self.grad.data_ptr(), self.momentum_buf.data_ptr(), self.trainable.data_ptr(), self.n,
                           float(self.lr if lr is None else lr), self.mom, self.wd, self.clip_norm, self.clip_value, int(self.steps == 0),
                           self._sgd_ws.data_ptr(), self._sgd_ws.numel(), _lib.stream_ptr())
        _lib.check(rc, 'pf_sgd_step')
        self.steps += 1

    def train_step(self, inputs, labels, lr=None):
        """train.py:185-216 for one batch: loss/backward (scaled by 1/accumulate_steps), and on every accumulate_steps-th
        call gradient exchange, clipping and the SGD update."""
        acc = max(1, self.accumulate_steps)
        out = self.forward_backward(inputs, labels, accumulate=self._micro > 0, loss_scale=1.0 / acc)
        self._micro += 1
        if self._micro >= acc:
            self._micro = 0
            self.all_reduce_grads()
            self.optimizer_step(lr)
        return out


def reference_parameter_order(keys):
    """The trainable state_dict keys in the order of the reference's ``model.parameters()``: ``hardnet`` registers ``base``,
    ``transUpBlocks`` (no parameters), ``denseBlocksUp``, ``conv1x1_up``, ``finalConv`` in that order (hardnet.py:274-327), a
    ConvLayer holds ``conv.weight, norm.weight, norm.bias`` (hardnet.py:16-25).  Pinned by the ``keys`` array of fixture G6."""
    group = {'base': 0, 'denseBlocksUp': 1, 'conv1x1_up': 2, 'finalConv': 3}
    leaf = {'conv.weight': 0, 'norm.weight': 1, 'norm.bias': 2, 'weight': 0, 'bias': 1}

    def rank(key):
        parts = key.split('.')          # model.base.4.layers.1.conv.weight
        nums = tuple(int(q) for q in parts if q.isdigit())
        tail = '.'.join(q for q in parts[2:] if not q.isdigit() and q != 'layers')
        return (group[parts[1]], nums, leaf[tail])
    return sorted(keys, key=rank)


class TrainStepFunction(torch.autograd.Function):
    """Autograd bridge.  forward = the fused device step (forward + loss + backward in one C call; the gradients are then
    already in ``trainer.grad``); backward hands them to autograd scaled by the incoming ``grad_output`` (the reference
    divides the loss by ``accumulate_steps`` before ``backward()``, train.py:200), one tensor per parameter in the order
    they were passed — so ``.grad`` accumulation, ``zero_grad`` and DDP's reducer hooks all behave as for a torch model.
    ``params`` must be the adopted parameters (views of ``trainer.theta``) in ``trainer.trainable_layout()`` order."""

    @staticmethod
    def forward(ctx, trainer, inputs, labels, *params):
        out = trainer.forward_backward(inputs, labels)
        ctx.trainer = trainer
        # the gradients of THIS forward live in the shared arena until the next forward overwrites them: backward() checks
        trainer._fwd_generation = getattr(trainer, '_fwd_generation', 0) + 1
        ctx.generation = trainer._fwd_generation
        ctx.mark_non_differentiable(out['accuracy'])
        return out['loss'].clone(), out['accuracy']

    @staticmethod
    def backward(ctx, grad_loss, _grad_acc):
        tr = ctx.trainer
        if ctx.generation != tr._fwd_generation:
            raise RuntimeError('BGModel.loss (training mode): backward() of a loss whose gradients were overwritten by a later '
                               'loss() call on the same model - call backward() after each loss() (the fused step keeps one '
                               'gradient arena), e.g. loss_a.backward(); loss_b.backward() instead of (loss_a + loss_b).backward()')
        scaled = tr.grad * grad_loss           # one pass over the flat array; the per-parameter results are views of it
        grads = []
        for _, off, shape, n in tr.trainable_layout():
            grads.append(scaled[off:off + n].view(shape))
        return (None, None, None) + tuple(grads)
