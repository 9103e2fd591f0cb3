"""Losses of the reference training loop (train.py:321-331), evaluated on device with tensor ops
instead of the reference's B x T Python double loop (utils.py:653-679)."""
import torch
import torch.nn.functional as F


def class_weight_tensor(bce_weight, labels):
    """utils.py:653-679: label 1 -> w[j][0], label 0 -> w[j][1], anything else (missing) -> 0."""
    w = torch.as_tensor(bce_weight, dtype=torch.float32, device=labels.device)
    return ((labels == 1).to(torch.float32) * w[:, 0].view(1, -1) +
            (labels == 0).to(torch.float32) * w[:, 1].view(1, -1)).reshape(-1)


def classification_loss(outputs, labels, bce_weight):
    """train.py:326-331: weighted BCE-with-logits, summed, divided by the number of labels in {0,1}."""
    weights = class_weight_tensor(bce_weight, labels)
    non_nan = ((labels == 1).sum() + (labels == 0).sum()).to(torch.float32)
    return F.binary_cross_entropy_with_logits(outputs.view(-1), labels.float().view(-1), weight=weights,
                                              reduction='sum') / non_nan


def regression_loss(outputs, labels):
    """train.py:321-325."""
    return F.mse_loss(outputs.view(-1), labels.float().view(-1))
