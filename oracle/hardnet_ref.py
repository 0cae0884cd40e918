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


_TRAINING = False      # module switch set by bg_train_step only (keeps the inference signatures unchanged)


def _conv_bn_relu(sd, p, x, stride=1):
    """ConvLayer (hardnet.py:16-25): conv(no bias, pad k//2) -> BN -> ReLU.  BN in eval form, or (inside bg_train_step)
    nn.BatchNorm2d's training form: batch statistics, running stats updated in place with momentum 0.1."""
    w = sd[p + '.conv.weight']
    x = F.conv2d(x, w, None, stride=stride, padding=w.shape[2] // 2)
    x = F.batch_norm(x, sd[p + '.norm.running_mean'], sd[p + '.norm.running_var'],
                     sd[p + '.norm.weight'], sd[p + '.norm.bias'], training=_TRAINING, momentum=0.1, eps=1e-5)
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


def trainable_keys(sd):
    """model.parameters() with requires_grad of the reference BGModel: conv weights, BN affine, finalConv weight/bias
    (depth_mean/depth_std are requires_grad=False, bg_model.py:40-41; running stats are buffers)."""
    return [k for k in sd if k.startswith('model.') and (k.endswith('.weight') or k.endswith('.bias'))]


def bg_loss(sd, inputs, labels, final_size=None):
    """BGModel.loss (bg_model.py:73-89): logits at final_size (or the input size) -> CrossEntropyLoss(ignore_index=255),
    accuracy = #(argmax == label) / #(label != 255)."""
    x = bg_inputs_to_tensor(sd, inputs['seg'], inputs['depth'], inputs['depth_mask'])
    logits, _ = hardnet_forward(sd, x, final_size)
    lab = labels['seg'].long()
    loss = F.cross_entropy(logits, lab, ignore_index=255)
    correct = (logits.argmax(1) == lab).sum()
    total = (lab != 255).sum()
    return {'loss': loss, 'accuracy': correct.float() / total.float()}


def bg_train_step(sd, inputs, labels, momentum_bufs=None, lr=2e-3, mom=0.9, wd=1e-4, clip_grad_norm=5.0, clip_grad=None,
                  final_size=None, apply_update=True):
    """One batch of the reference training loop (training/train.py:185-216, accumulate_steps=1) on a state_dict:
    model.train(); loss; backward; clip_grad_value_ | clip_grad_norm_; torch.optim.SGD step (weight decay added to the
    gradient, buf = g on the first step, then mom*buf + g; p -= lr*buf).  Updates ``sd`` in place (parameters and BN
    running statistics).  Returns {'loss','accuracy','grads' (after clipping),'grad_norm' (before),'momentum_bufs'}."""
    global _TRAINING
    keys = trainable_keys(sd)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    _TRAINING = True
    try:
        with torch.enable_grad():
            out = bg_loss(work, inputs, labels, final_size)
            out['loss'].backward()
    finally:
        _TRAINING = False
    grads = {k: leaves[k].grad.detach() for k in keys}
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    if clip_grad is not None:
        grads = {k: g.clamp(-clip_grad, clip_grad) for k, g in grads.items()}
    elif clip_grad_norm is not None:
        coef = torch.clamp(clip_grad_norm / (total + 1e-6), max=1.0)
        grads = {k: g * coef for k, g in grads.items()}
    new_bufs = {}
    if apply_update:
        with torch.no_grad():
            for k in keys:
                g = grads[k] + wd * sd[k]
                buf = g.clone() if (momentum_bufs is None or mom == 0) else mom * momentum_bufs[k] + g
                new_bufs[k] = buf
                sd[k] = sd[k] - lr * buf
    return {'loss': out['loss'].detach(), 'accuracy': out['accuracy'].detach(), 'grads': grads, 'grad_norm': total,
            'momentum_bufs': new_bufs}
