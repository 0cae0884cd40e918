"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

Plain torch fp32 restatement of the reference bg model's forward pass, driven
directly by a reference-format state_dict (no nn.Module tree):

  BGModel._inp2onehot / forward / predict  /root/reference/panoptic_forecasting/models/bg/bg_model.py:50-102
  hardnet.forward, HarDBlock.forward, TransitionUp  .../models/bg/hardnet.py:220-258,353-387

It is a floating-point kernel, so the checker is torch (F.conv2d etc.); pinned
against outputs of the reference itself: fixtures tests/golden/g3_*.npz (generator
tests/golden/make_golden.py).  Also serves as bench.py's cpu_baseline leg
("port": it runs the same ATen CPU convolutions the reference would).
"""
import torch
import torch.nn.functional as F

GROWTH = (10, 16, 18, 24, 32)        # hardnet.py:265-269
N_LAYERS = (4, 4, 8, 8, 8)
N_CLS_BG = 11


def _conv_bn_relu(sd, p, x, stride=1):
    """ConvLayer (hardnet.py:16-25): conv(no bias, pad k//2) -> BN(eval) -> ReLU."""
    w = sd[p + '.conv.weight']
    x = F.conv2d(x, w, None, stride=stride, padding=w.shape[2] // 2)
    x = F.batch_norm(x, sd[p + '.norm.running_mean'], sd[p + '.norm.running_var'],
                     sd[p + '.norm.weight'], sd[p + '.norm.bias'], training=False, eps=1e-5)
    return F.relu(x)


def _links(layer):
    """hardnet.py:177-194: layer reads layer-2^i for every 2^i dividing it, most recent first."""
    out, p = [], 1
    while p <= layer:
        if layer % p == 0:
            out.append(layer - p)
        p *= 2
    return out


def _hardblock(sd, p, x, n_layers):
    """HarDBlock.forward (hardnet.py:220-240)."""
    outs = [x]
    for l in range(1, n_layers + 1):
        ins = [outs[j] for j in _links(l)]
        xin = torch.cat(ins, 1) if len(ins) > 1 else ins[0]
        outs.append(_conv_bn_relu(sd, '%s.layers.%d' % (p, l - 1), xin))
    keep = [outs[i] for i in range(1, n_layers + 1) if i % 2 == 1 or i == n_layers]
    return torch.cat(keep, 1)


def hardnet_forward(sd, x, final_size=None, prefix='model.', taps=None):
    """hardnet.forward (hardnet.py:353-387). Returns (final_logits, orig_size_logits)."""
    size_in = x.shape[-2:]
    strides = (2, 1, 2, 1)
    for i in range(4):
        x = _conv_bn_relu(sd, '%sbase.%d' % (prefix, i), x, strides[i])
        if taps is not None:
            taps['base.%d' % i] = x
    idx, skips = 4, []
    nb = len(N_LAYERS)
    for i in range(nb):
        x = _hardblock(sd, '%sbase.%d' % (prefix, idx), x, N_LAYERS[i])
        if taps is not None:
            taps['base.%d' % idx] = x
        if i < nb - 1:
            skips.append(x)
        x = _conv_bn_relu(sd, '%sbase.%d' % (prefix, idx + 1), x)
        idx += 2
        if i < nb - 1:
            x = F.avg_pool2d(x, 2, 2)
            idx += 1
    for j in range(nb - 1):
        skip = skips.pop()
        x = F.interpolate(x, size=skip.shape[-2:], mode='bilinear', align_corners=True)   # :248-253
        x = torch.cat([x, skip], 1)                                                       # :256
        x = _conv_bn_relu(sd, '%sconv1x1_up.%d' % (prefix, j), x)
        x = _hardblock(sd, '%sdenseBlocksUp.%d' % (prefix, j), x, N_LAYERS[nb - 2 - j])
        if taps is not None:
            taps['denseBlocksUp.%d' % j] = x
    out = F.conv2d(x, sd[prefix + 'finalConv.weight'], sd[prefix + 'finalConv.bias'])
    size = tuple(final_size) if final_size is not None else tuple(size_in)
    final = F.interpolate(out, size=size, mode='bilinear', align_corners=True)
    return final, out


def bg_inputs_to_tensor(sd, seg, depth, depth_mask, num_classes=N_CLS_BG):
    """bg_model.py:53-69: one-hot (labels >= num_classes -> zero vector), normalised masked depth, cat."""
    seg = seg.long()
    m = seg < num_classes
    oh = F.one_hot(torch.where(m, seg, torch.zeros_like(seg)), num_classes) * m.unsqueeze(-1)
    b, t, h, w = seg.shape
    x = oh.permute(0, 1, 4, 2, 3).float().reshape(b, t * num_classes, h, w)
    dn = (depth - sd['depth_mean']) / sd['depth_std']
    dn = dn * depth_mask
    return torch.cat([x, dn], 1)


@torch.no_grad()
def bg_predict(sd, inputs, final_size=None, taps=None):
    """BGModel.predict (bg_model.py:91-102) -> {'seg','logits','orig_size_logits'}."""
    x = bg_inputs_to_tensor(sd, inputs['seg'], inputs['depth'], inputs['depth_mask'])
    logits, orig = hardnet_forward(sd, x, final_size, taps=taps)
    return {'seg': logits.argmax(1), 'logits': logits, 'orig_size_logits': orig}
