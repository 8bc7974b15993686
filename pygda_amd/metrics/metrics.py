"""Evaluation metrics with the reference's names (pygda/metrics/metrics.py).  F1 is
computed from a confusion matrix on the device the predictions live on (one small D2H of
C*C counts) instead of shipping every label to sklearn each epoch."""
import torch


def _confusion(label, pred):
    label, pred = label.reshape(-1).long(), pred.reshape(-1).long()
    c = int(max(int(label.max()), int(pred.max()))) + 1 if label.numel() else 1
    cm = torch.bincount(label * c + pred, minlength=c * c).reshape(c, c)
    return cm.cpu().double(), c


def eval_micro_f1(label, pred):
    """sklearn ``f1_score(average='micro')`` == accuracy for single-label data (:160-193)."""
    cm, _ = _confusion(label, pred)
    tot = cm.sum().item()
    return float(cm.diag().sum().item() / tot) if tot else 0.0


def eval_macro_f1(label, pred):
    """Unweighted mean of per-class F1 over classes present in labels or predictions (:196-229)."""
    cm, _ = _confusion(label, pred)
    tp = cm.diag()
    fp, fn = cm.sum(0) - tp, cm.sum(1) - tp
    present = (cm.sum(0) + cm.sum(1)) > 0
    denom = 2 * tp + fp + fn
    f1 = torch.where(denom > 0, 2 * tp / denom.clamp(min=1), torch.zeros_like(tp))
    return float(f1[present].mean().item()) if present.any() else 0.0


def eval_roc_auc(label, score):
    from sklearn.metrics import roc_auc_score
    v = roc_auc_score(y_true=label.cpu().numpy(), y_score=score.cpu().numpy())
    return 1 - v if v < 0.5 else v


def eval_average_precision(label, score):
    from sklearn.metrics import average_precision_score
    return average_precision_score(y_true=label.cpu().numpy(), y_score=score.cpu().numpy())


def eval_recall_at_k(label, score, k=None):
    k = int(sum(label)) if k is None else k
    return sum(label[score.topk(k).indices]) / sum(label)


def eval_precision_at_k(label, score, k=None):
    k = int(sum(label)) if k is None else k
    return sum(label[score.topk(k).indices]) / k
