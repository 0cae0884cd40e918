"""Checkpoint -> weight blob for ``pf_hardnet_plan_create`` (include/pfhip.h).

Takes a state_dict with the reference's key names (``model.base.N.conv.weight``,
``model.base.N.norm.*``, ``model.base.N.layers.M.*``, ``model.conv1x1_up.N.*``,
``model.denseBlocksUp.N.layers.M.*``, ``model.finalConv.{weight,bias}``; see
reference ``models/bg/hardnet.py:274-327``), folds every eval-mode BatchNorm
into its conv (``ConvLayer`` = conv(no bias) + BN + ReLU, ``hardnet.py:16-25``)
and serialises the op table of ``hardnet_arch.Spec`` plus the folded weights.

Blob layout (little endian, all offsets from the start of the blob):

  header   64 B : magic "PFHNET02", u32 version, n_tensors, n_ops, in_ch, n_cls, pad,
                  u64 tensor_table_off, op_table_off, weights_off, total_bytes
  tensors  n_tensors x 48 B {u32 channels, u32 pad, char name[40]}
  ops      n_ops x 128 B {u32 kind,k,stride,relu,cin,cout,n_src,dst,dst_choff,bn_tag (1 + has BatchNorm; convs only),pad[2],
                          {u32 tensor,choff,ch}[4], u64 w_off, u64 b_off, pad to 128}
  weights  fp32: per conv OIHW [cout][cin][k][k] then bias[cout]  (w_off/b_off in floats)
"""
import struct

import numpy as np
import torch

from . import hardnet_arch as arch

MAGIC = b'PFHNET02'
VERSION = 2
TENSOR_BYTES = 48
OP_BYTES = 128
MAX_SRC = 4
BN_EPS = 1e-5  # nn.BatchNorm2d default, as constructed at hardnet.py:21


def fold_conv_bn(sd, prefix):
    """Folded (weight[cout,cin,k,k], bias[cout]) of one ConvLayer, computed in float64."""
    w = sd[prefix + '.conv.weight'].detach().double().cpu()
    gamma = sd[prefix + '.norm.weight'].detach().double().cpu()
    beta = sd[prefix + '.norm.bias'].detach().double().cpu()
    mean = sd[prefix + '.norm.running_mean'].detach().double().cpu()
    var = sd[prefix + '.norm.running_var'].detach().double().cpu()
    scale = gamma / torch.sqrt(var + BN_EPS)
    return (w * scale.view(-1, 1, 1, 1)).float(), (beta - mean * scale).float()


def folded_params(sd, spec, prefix='model.'):
    """{op name: (w, b)} for every conv op of ``spec``."""
    out = {}
    for op in spec.conv_ops():
        if op.bn:
            w, b = fold_conv_bn(sd, prefix + op.name)
        else:
            w = sd[prefix + op.name + '.weight'].detach().float().cpu()
            b = sd[prefix + op.name + '.bias'].detach().float().cpu()
        if tuple(w.shape) != (op.cout, op.cin, op.k, op.k):
            raise ValueError('%s: weight shape %s, expected %s' %
                             (op.name, tuple(w.shape), (op.cout, op.cin, op.k, op.k)))
        out[op.name] = (w.contiguous(), b.contiguous())
    return out


def pack_blob(sd, in_ch=36, n_cls=11, prefix='model.', spec=None, params=None):
    """bytes: the blob handed to the C library.

    ``spec``/``params`` let tests serialise an arbitrary op table with ready-made
    {op name: (weight, bias)} instead of the bg network + checkpoint."""
    if spec is None:
        spec = arch.Spec(in_ch, n_cls)
    if params is None:
        params = folded_params(sd, spec, prefix)
    chunks = []
    w_off = {}
    n_floats = 0
    for op in spec.conv_ops():
        w, b = params[op.name]
        w_off[op.name] = (n_floats, n_floats + w.numel())
        chunks += [w.numpy().ravel(), b.numpy().ravel()]
        n_floats += w.numel() + b.numel()
    weights = np.concatenate(chunks).astype('<f4')

    t_off = 64
    o_off = t_off + TENSOR_BYTES * len(spec.tensors)
    wts_off = (o_off + OP_BYTES * len(spec.ops) + 63) // 64 * 64
    total = wts_off + weights.nbytes
    out = bytearray(total)
    struct.pack_into('<8sIIIIII4Q', out, 0, MAGIC, VERSION, len(spec.tensors), len(spec.ops), in_ch,
                     n_cls, 0, t_off, o_off, wts_off, total)
    for i, t in enumerate(spec.tensors):
        struct.pack_into('<II40s', out, t_off + TENSOR_BYTES * i, t.channels, 0, t.name.encode()[:39])
    for i, op in enumerate(spec.ops):
        if len(op.srcs) > MAX_SRC:
            raise ValueError('%s has %d sources (max %d)' % (op.name, len(op.srcs), MAX_SRC))
        base = o_off + OP_BYTES * i
        struct.pack_into('<12I', out, base, op.kind, op.k, op.stride, int(op.relu), op.cin, op.cout,
                         len(op.srcs), op.dst, op.dst_choff,
                         (1 + int(op.bn)) if op.kind in (arch.OP_STEM, arch.OP_CONV) else 0, 0, 0)
        for j, s in enumerate(op.srcs):
            struct.pack_into('<3I', out, base + 48 + 12 * j, s.tensor, s.choff, s.ch)
        wo, bo = w_off.get(op.name, (0, 0))
        struct.pack_into('<2Q', out, base + 96, wo, bo)
    out[wts_off:wts_off + weights.nbytes] = weights.tobytes()
    return bytes(out)
