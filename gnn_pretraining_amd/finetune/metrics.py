"""Fine-tuning metrics with the keys and aggregation of src/finetune/metrics.py (sklearn scores on host copies;
sample-weighted means over batches).  Reporting only: nothing here feeds back into training."""
from __future__ import annotations

import time
from typing import Dict, List

import numpy as np
import torch
from torch import Tensor

from ..constants import NUM_CLASSES


def _aggregate_batch_metrics(batch_metrics: List[Dict[str, float]], epoch: int, prefix: str) -> Dict[str, float]:
    """metrics.py:15-36"""
    names = set(batch_metrics[0].keys()) - {"num_samples"}
    total = sum(b["num_samples"] for b in batch_metrics)
    out = {k: sum(b[k] * b["num_samples"] for b in batch_metrics) / total for k in names}
    if prefix != "val":
        out[f"{prefix}/progress/epoch"] = epoch
    return out


def compute_batch_metrics(domain_name: str, targets: Tensor, predictions: Tensor, probabilities: Tensor, loss: Tensor,
                          prefix: str) -> Dict[str, float]:
    """metrics.py:39-82"""
    from sklearn.metrics import accuracy_score, f1_score, precision_score, recall_score, roc_auc_score
    binary = NUM_CLASSES[domain_name] == 2
    y_true, y_pred = targets.detach().cpu().numpy(), predictions.detach().cpu().numpy()
    y_prob = probabilities.detach().cpu().numpy()
    if binary:
        y_prob = y_prob[:, 1]
    avg = "binary" if binary else "macro"
    m = {f"{prefix}/accuracy": float(accuracy_score(y_true, y_pred)),
         f"{prefix}/f1": float(f1_score(y_true, y_pred, average=avg, zero_division=0)),
         f"{prefix}/precision": float(precision_score(y_true, y_pred, average=avg, zero_division=0)),
         f"{prefix}/recall": float(recall_score(y_true, y_pred, average=avg, zero_division=0))}
    if len(np.unique(y_true)) < 2:
        m[f"{prefix}/auc"] = 0.0
    else:
        try:
            m[f"{prefix}/auc"] = float(roc_auc_score(y_true, y_prob) if binary else roc_auc_score(y_true, y_prob, multi_class="ovr"))
        except (ValueError, RuntimeWarning):
            m[f"{prefix}/auc"] = 0.0
    m[f"{prefix}/loss"] = float(loss.item())
    m["num_samples"] = len(targets)
    return m


def compute_training_metrics(epoch: int, step: int, loss: Tensor, optimizer: torch.optim.Optimizer, domain_name: str,
                             targets: Tensor, predictions: Tensor, probabilities: Tensor, step_start_time: float,
                             model: torch.nn.Module) -> Dict[str, float]:
    """metrics.py:85-118"""
    m = compute_batch_metrics(domain_name, targets, predictions, probabilities, loss, "train")
    for pg in optimizer.param_groups:
        m[f'train/lr/{pg["name"]}'] = pg["lr"]
    sq = [p.grad.detach().pow(2).sum() for p in model.parameters() if p.grad is not None and p.requires_grad]
    m["train/gradients/model_grad_norm"] = float(torch.stack(sq).sum().sqrt()) if sq else 0.0     # one sync, not one per tensor
    m["train/progress/epoch"], m["train/progress/step"] = epoch, step
    m["train/system/time_per_step"] = time.time() - step_start_time
    return m


def compute_validation_metrics(batch_metrics: List[Dict[str, float]], epoch: int) -> Dict[str, float]:
    return _aggregate_batch_metrics(batch_metrics, epoch, "val")


def compute_test_metrics(batch_metrics: List[Dict[str, float]], epoch: int, epochs_since_improvement: int,
                         training_start_time: float, model: torch.nn.Module) -> Dict[str, float]:
    """metrics.py:128-144"""
    m = _aggregate_batch_metrics(batch_metrics, epoch, "test")
    m["test/convergence_epochs"] = epoch - epochs_since_improvement
    m["test/training_time"] = time.time() - training_start_time
    m["test/total_parameters"] = sum(p.numel() for p in model.parameters())
    m["test/trainable_parameters"] = sum(p.numel() for p in model.parameters() if p.requires_grad)
    return m
