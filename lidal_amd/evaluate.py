"""Validation mIoU on the GPU: counterpart of /root/reference/evaluate.py:95-124 and
utils/iou_sk.py:14-52.  The reference copies the logits to the host, gathers
voxel -> point, arg-maxes and bincounts in numpy per batch, then all-reduces the 19x19 int32 matrix;
here one kernel per batch accumulates the matrix on the device and the same all-reduce follows."""
import numpy as np
import torch
import torch.distributed as dist

from . import SparseTensor
from . import backend as B

__all__ = ['confusion_accumulate', 'evaluate_batches', 'iou_from_confusion']


def confusion_accumulate(conf, logits_v_b, inverse_indices_b, labels_p_b):
    """conf i32 [C, C] (device, updated in place) += confusion of this batch
    (rows = prediction, columns = ground truth, labels >= 100 ignored: iou_sk.py:17)."""
    B.require_gpu(conf, logits_v_b, inverse_indices_b, labels_p_b)
    logits = logits_v_b.contiguous().float()
    inv = inverse_indices_b.contiguous().long()
    lab = labels_p_b.contiguous().long()
    assert conf.dtype == torch.int32 and conf.is_contiguous() and inv.shape == lab.shape
    c = logits.shape[1]
    B.check(B.lib().lidal_confusion_accumulate(B.ptr(logits), B.ptr(inv), B.ptr(lab), inv.numel(), c,
                                               B.ptr(conf), B.stream()), 'confusion_accumulate')
    return conf


def iou_from_confusion(confusion):
    """utils/iou_sk.py:21-52 without the printing: (per-class IoU list, mean IoU)."""
    confusion = np.asarray(confusion)
    n = confusion.shape[0]
    ious = []
    for i in range(n):
        tp = np.int32(confusion[i, i])
        fp = np.int32(confusion[i, :].sum()) - tp
        fn = np.int32(confusion[:, i].sum()) - tp
        denom = tp + fp + fn
        ious.append(float('nan') if denom == 0 else float(tp) / denom)
    return ious, sum(ious) / n


@torch.no_grad()
def evaluate_batches(model, batches, n_classes=19, group=None):
    """batches: iterable of dicts with coords_v_b, feats_v_b, inverse_indices_b, labels_p_b on the
    GPU (the reference's val collate).  Returns (confusion i32 [C,C] numpy, per-class IoU, mIoU)."""
    model.eval()
    # on the model's device from the start: a rank without batches still joins the all-reduce
    conf = torch.zeros((n_classes, n_classes), dtype=torch.int32, device=next(model.parameters()).device)
    for b in batches:
        logits, _ = model(SparseTensor(b['feats_v_b'], b['coords_v_b']))
        confusion_accumulate(conf, logits, b['inverse_indices_b'], b['labels_p_b'])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(conf, op=dist.ReduceOp.SUM, group=group)          # evaluate.py:117-119
    conf = conf.cpu().numpy()
    ious, miou = iou_from_confusion(conf)
    return conf, ious, miou
