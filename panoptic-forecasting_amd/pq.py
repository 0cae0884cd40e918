"""Panoptic-quality accumulators for bg ("stuff") label maps + their cross-rank reduction.

The reference computes PQ with the external ``cityscapesscripts.evaluation.evalPanopticSemanticLabeling``
(scripts/fg/run_fg_eval_panoptic.sh:28-33; version unpinned, absent here => "parity unpinned").  This
restates the standard definition for stuff classes, where every class forms at most one segment per
image: a predicted and a ground-truth segment of the same class match iff IoU > 0.5 (void pixels,
label 255 in the ground truth, are removed from the prediction first);
    PQ_c = sum(IoU over TP) / (TP + FP/2 + FN/2),   PQ = mean over classes with TP+FP+FN > 0.
Accumulators are [n_cls, 4] float64 rows (sum_iou, TP, FP, FN): integer-valued counts add exactly, so a
sharded evaluation reproduces the single-process numbers (sum_iou to float64 rounding).
"""
import torch

VOID = 255


def pq_accumulate(pred, gt, n_cls, acc=None):
    """pred, gt: [B,H,W] integer label maps (any device). Returns/updates acc [n_cls,4] float64 on that device."""
    if acc is None:
        acc = torch.zeros(n_cls, 4, dtype=torch.float64, device=pred.device)
    b = pred.shape[0]
    p = pred.reshape(b, -1).long()
    g = gt.reshape(b, -1).long()
    keep = g != VOID
    pc = torch.where(keep & (p < n_cls), p, torch.full_like(p, n_cls))      # bucket n_cls = ignored
    gc = torch.where(keep & (g < n_cls), g, torch.full_like(g, n_cls))
    k = n_cls + 1
    idx = (torch.arange(b, device=p.device).view(b, 1) * k + gc) * k + pc
    conf = torch.bincount(idx.reshape(-1), minlength=b * k * k).view(b, k, k)[:, :n_cls + 1, :n_cls + 1].double()
    inter = torch.diagonal(conf, dim1=1, dim2=2)[:, :n_cls]                  # [B, n_cls]
    g_area = conf.sum(2)[:, :n_cls]
    p_area = conf[:, :n_cls + 1, :n_cls].sum(1)                               # predictions on non-void gt pixels
    union = g_area + p_area - inter
    iou = torch.where(union > 0, inter / union.clamp(min=1), torch.zeros_like(union))
    tp = iou > 0.5
    fp = (p_area > 0) & ~tp
    fn = (g_area > 0) & ~tp
    acc[:, 0] += (iou * tp).sum(0)
    acc[:, 1] += tp.sum(0).double()
    acc[:, 2] += fp.sum(0).double()
    acc[:, 3] += fn.sum(0).double()
    return acc


def pq_from_acc(acc):
    """{'pq','sq','rq','per_class'} in percent from [n_cls,4] accumulators."""
    acc = acc.double().cpu()
    siou, tp, fp, fn = acc[:, 0], acc[:, 1], acc[:, 2], acc[:, 3]
    denom = tp + 0.5 * fp + 0.5 * fn
    have = denom > 0
    pq_c = torch.where(have, siou / denom.clamp(min=1e-12), torch.zeros_like(siou))
    sq_c = torch.where(tp > 0, siou / tp.clamp(min=1), torch.zeros_like(siou))
    rq_c = torch.where(have, tp / denom.clamp(min=1e-12), torch.zeros_like(siou))
    n = max(int(have.sum()), 1)
    return {'pq': 100.0 * float(pq_c[have].sum()) / n, 'sq': 100.0 * float(sq_c[have].sum()) / n,
            'rq': 100.0 * float(rq_c[have].sum()) / n, 'per_class': (100.0 * pq_c).tolist(),
            'n_classes': int(have.sum())}
