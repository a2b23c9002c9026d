"""Evaluation metrics and schedules of the drivers (SURVEY.md §2a util/util.py, §8f rank 2).

* ``intersection_and_union``  - /root/reference/util/util.py:139-156 (``intersectionAndUnionGPU``) kept on the device
  (bincount instead of a CPU ``histc`` round trip), multiple ignore labels
* ``cosine_learning_rate`` / ``poly_learning_rate`` - util.py:112-121
* ``open_vocab_scores`` - mIoU over base / novel classes and their harmonic mean hIoU, as reported in README.md:77-83
"""
from __future__ import annotations

import math

import torch


def intersection_and_union(output, target, K, ignore_indexs=(255,)):
    assert output.shape == target.shape
    output = output.reshape(-1).clone()
    target = target.reshape(-1)
    for ig in ignore_indexs:
        output[target == ig] = ig
    valid = (target >= 0) & (target < K)
    inter = output[(output == target) & valid]
    area_i = torch.bincount(inter, minlength=K)[:K].float()
    area_o = torch.bincount(output[(output >= 0) & (output < K)], minlength=K)[:K].float()
    area_t = torch.bincount(target[valid], minlength=K)[:K].float()
    return area_i, area_o + area_t - area_i, area_t


def cosine_learning_rate(base_lr, curr_iter, max_iter):
    return base_lr * 0.5 * (1 + math.cos(math.pi * curr_iter / max_iter))


def poly_learning_rate(base_lr, curr_iter, max_iter, power=0.9):
    return base_lr * (1 - float(curr_iter) / max_iter) ** power


def open_vocab_scores(intersection, union, base, novel):
    iou = intersection / (union + 1e-10)
    m_base, m_novel = float(iou[list(base)].mean()), float(iou[list(novel)].mean())
    h = 2 * m_base * m_novel / (m_base + m_novel + 1e-10)
    return {"hIoU": h, "mIoU_base": m_base, "mIoU_novel": m_novel}


class AverageMeter:
    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum = self.sum + val * n
        self.count += n
        self.avg = self.sum / self.count
