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


def _category(ids):
    return torch.where(ids > 100, ids // 1000, ids)


def pq_accumulate_panoptic(pred, gt, n_cls=19, acc=None):
    """Panoptic maps with instances: [B,H,W] integer maps in the merge output format (stuff = trainId, thing =
    (trainId)*1000 + instance, void = 255; panoptic.py / fg_model.py:572).  Standard PQ matching: a predicted and a
    ground-truth segment of the same category match iff IoU > 0.5, the union not counting predicted pixels on void;
    unmatched predictions lying more than half on void are not false positives.  Same [n_cls,4] accumulators as
    pq_accumulate (the segment tables are tiny: the matching itself runs on the host)."""
    if acc is None:
        acc = torch.zeros(n_cls, 4, dtype=torch.float64, device=pred.device)
    add = torch.zeros(n_cls, 4, dtype=torch.float64)
    for b in range(pred.shape[0]):
        p = pred[b].reshape(-1).long()
        g = gt[b].reshape(-1).long()
        uk, cnt = torch.unique(g * 65536 + p, return_counts=True)
        uk, cnt = uk.cpu(), cnt.cpu().double()
        ug, up = uk // 65536, uk % 65536
        g_ids, g_inv = torch.unique(ug, return_inverse=True)
        p_ids, p_inv = torch.unique(up, return_inverse=True)
        g_area = torch.zeros(len(g_ids), dtype=torch.float64).index_add_(0, g_inv, cnt)
        p_area = torch.zeros(len(p_ids), dtype=torch.float64).index_add_(0, p_inv, cnt)
        p_void = torch.zeros(len(p_ids), dtype=torch.float64).index_add_(0, p_inv, cnt * (ug == VOID))
        g_cat, p_cat = _category(g_ids), _category(p_ids)
        g_hit = torch.zeros(len(g_ids), dtype=torch.bool)
        p_hit = torch.zeros(len(p_ids), dtype=torch.bool)
        same = (g_cat[g_inv] == p_cat[p_inv]) & (ug != VOID) & (up != VOID) & (g_cat[g_inv] < n_cls)
        union = g_area[g_inv] + p_area[p_inv] - cnt - p_void[p_inv]
        iou = cnt / union.clamp(min=1)
        for k in torch.nonzero(same & (iou > 0.5)).flatten().tolist():
            c = int(g_cat[g_inv[k]])
            add[c, 0] += float(iou[k])
            add[c, 1] += 1
            g_hit[g_inv[k]] = True
            p_hit[p_inv[k]] = True
        for i in range(len(g_ids)):
            if not g_hit[i] and g_ids[i] != VOID and g_cat[i] < n_cls:
                add[int(g_cat[i]), 3] += 1
        for i in range(len(p_ids)):
            if not p_hit[i] and p_ids[i] != VOID and p_cat[i] < n_cls and p_void[i] <= 0.5 * p_area[i]:
                add[int(p_cat[i]), 2] += 1
    acc += add.to(acc.device)
    return acc
