"""FC-HarDNet-70 as a flat op table (the bg decoder the HIP library executes).

The reference builds the network as nested ``nn.Module``s and concatenates
activations with ``torch.cat`` (``models/bg/hardnet.py:176-240,261-387``).  Here
the same network is *described*: a list of tensors and a list of ops whose
inputs are channel ranges of earlier tensors, so the device code never
concatenates — every HarDBlock layer writes straight into the channel slot its
consumers read.  The table is serialised into the weight blob
(``packing.py``) and interpreted by ``csrc/hardnet_plan.cpp``.

Architecture constants: ``hardnet.py:265-269``.  Link rule: ``hardnet.py:177-194``
(layer l reads layers l-2^i for every 2^i dividing l, most recent first; width
grows by 1.7 per extra link, rounded to even).  Block output = odd layers + last
(``hardnet.py:233-239``).  State-dict names follow ``hardnet.py:274-327``.
"""
from collections import namedtuple

FIRST_CH = (16, 24, 32, 48)
CH_LIST = (64, 96, 160, 224, 320)
GROWTH = (10, 16, 18, 24, 32)
GRMUL = 1.7
N_LAYERS = (4, 4, 8, 8, 8)

# op kinds (shared with csrc/pf_blob.h)
OP_STEM, OP_CONV, OP_POOL, OP_UPSAMPLE, OP_HEAD = 0, 1, 2, 3, 4

Tensor = namedtuple('Tensor', 'name channels')
Src = namedtuple('Src', 'tensor choff ch')


class Op:
    def __init__(self, kind, name, srcs, dst, dst_choff, cin, cout, k=1, stride=1, relu=True,
                 bn=True):
        self.kind, self.name, self.srcs = kind, name, list(srcs)
        self.dst, self.dst_choff = dst, dst_choff
        self.cin, self.cout, self.k, self.stride, self.relu, self.bn = cin, cout, k, stride, relu, bn

    def __repr__(self):
        return 'Op(%s %s %d->%d k%d s%d)' % (self.kind, self.name, self.cin, self.cout, self.k,
                                             self.stride)


def layer_width_and_links(layer, growth):
    """(out_channels, links) of HarDBlock layer ``layer`` (1-based); layer 0 is the block input."""
    width = growth
    links = []
    p = 1
    for i in range(10):
        if layer % p == 0:
            links.append(layer - p)
            if i > 0:
                width *= GRMUL          # float multiply, as hardnet.py:188
        p *= 2
    width = int(int(width + 1) / 2) * 2
    return width, links


class Spec:
    """Tensors + ops of the whole bg net for ``in_ch`` input channels and ``n_cls`` classes."""

    def __init__(self, in_ch=36, n_cls=11):
        self.in_ch, self.n_cls = in_ch, n_cls
        self.tensors = []
        self.ops = []
        self._build()

    # -- helpers -----------------------------------------------------------
    def _tensor(self, name, ch):
        self.tensors.append(Tensor(name, ch))
        return len(self.tensors) - 1

    def _conv(self, name, srcs, cout, k, stride=1, dst=None, dst_choff=0, relu=True, bn=True,
              kind=OP_CONV):
        cin = sum(s.ch for s in srcs)
        if dst is None:
            dst = self._tensor(name, cout)
        self.ops.append(Op(kind, name, srcs, dst, dst_choff, cin, cout, k, stride, relu, bn))
        return dst

    def _whole(self, t):
        return [Src(t, 0, self.tensors[t].channels)]

    def _hardblock(self, prefix, x, growth, n_layers):
        """Returns the block-output tensor id; x = tensor id of the block input."""
        widths = [self.tensors[x].channels]
        links = [None]
        for l in range(1, n_layers + 1):
            w, lk = layer_width_and_links(l, growth)
            widths.append(w)
            links.append(lk)
        keep = [l for l in range(1, n_layers + 1) if l % 2 == 1 or l == n_layers]
        out_ch = sum(widths[l] for l in keep)
        out_t = self._tensor(prefix + '.out', out_ch)
        # where each layer's output lives: (tensor, channel offset)
        home = {0: (x, 0)}
        off = 0
        for l in keep:
            home[l] = (out_t, off)
            off += widths[l]
        for l in range(1, n_layers + 1):
            if l not in home:
                home[l] = (self._tensor('%s.L%d' % (prefix, l), widths[l]), 0)
        for l in range(1, n_layers + 1):
            srcs = [Src(home[j][0], home[j][1], widths[j]) for j in links[l]]
            self._conv('%s.layers.%d' % (prefix, l - 1), srcs, widths[l], 3,
                       dst=home[l][0], dst_choff=home[l][1])
        return out_t

    # -- the network ---------------------------------------------------------
    def _build(self):
        x = self._tensor('input', self.in_ch)          # virtual: never materialised (fused stem)
        self.input_tensor = x
        t = self._conv('base.0', self._whole(x), FIRST_CH[0], 3, 2, kind=OP_STEM)
        t = self._conv('base.1', self._whole(t), FIRST_CH[1], 3, 1)
        t = self._conv('base.2', self._whole(t), FIRST_CH[2], 3, 2)
        t = self._conv('base.3', self._whole(t), FIRST_CH[3], 3, 1)
        idx = 4
        skips = []
        nb = len(N_LAYERS)
        for i in range(nb):
            blk = self._hardblock('base.%d' % idx, t, GROWTH[i], N_LAYERS[i])
            if i < nb - 1:
                skips.append(blk)
            idx += 1
            t = self._conv('base.%d' % idx, self._whole(blk), CH_LIST[i], 1)
            idx += 1
            if i < nb - 1:
                ch = self.tensors[t].channels
                p = self._tensor('base.%d' % idx, ch)
                self.ops.append(Op(OP_POOL, 'base.%d' % idx, self._whole(t), p, 0, ch, ch, 2, 2,
                                   relu=False, bn=False))
                t = p
                idx += 1
        self.n_base = idx
        prev = t
        for j, i in enumerate(range(nb - 2, -1, -1)):
            skip = skips[i]
            ch = self.tensors[prev].channels
            up = self._tensor('transUp.%d' % j, ch)
            # the upsample takes its output size from the skip tensor (second source)
            self.ops.append(Op(OP_UPSAMPLE, 'transUp.%d' % j, self._whole(prev) + self._whole(skip),
                               up, 0, ch, ch, 1, 1, relu=False, bn=False))
            cat_ch = ch + self.tensors[skip].channels
            t = self._conv('conv1x1_up.%d' % j, self._whole(up) + self._whole(skip), cat_ch // 2, 1)
            prev = self._hardblock('denseBlocksUp.%d' % j, t, GROWTH[i], N_LAYERS[i])
        self.logits_tensor = self._conv('finalConv', self._whole(prev), self.n_cls, 1, relu=False,
                                        bn=False)
        self.ops.append(Op(OP_HEAD, 'head', self._whole(self.logits_tensor), self.logits_tensor, 0,
                           self.n_cls, self.n_cls, 1, 1, relu=False, bn=False))

    # -- views ---------------------------------------------------------------
    def conv_ops(self):
        return [o for o in self.ops if o.kind in (OP_STEM, OP_CONV)]

    def flops(self, h, w):
        """Dense conv FLOPs for an h×w input (2·Cout·Hout·Wout·Cin·k²)."""
        dims = {self.input_tensor: (h, w)}
        total = 0
        for o in self.ops:
            ih, iw = dims[o.srcs[0].tensor]
            if o.kind in (OP_STEM, OP_CONV):
                oh = (ih + 2 * (o.k // 2) - o.k) // o.stride + 1
                ow = (iw + 2 * (o.k // 2) - o.k) // o.stride + 1
                total += 2 * o.cout * oh * ow * o.cin * o.k * o.k
            elif o.kind == OP_POOL:
                oh, ow = ih // 2, iw // 2
            elif o.kind == OP_UPSAMPLE:
                oh, ow = dims[o.srcs[1].tensor]
            else:
                oh, ow = ih, iw
            dims[o.dst] = (oh, ow)
        return total
