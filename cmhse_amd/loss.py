"""Host-side mirror of the reference's `loss.py` on the MI355X hot path.

`cosine_sim` and `ContrastiveLoss` keep the names, constructor arguments and forward contract of
/root/reference/loss.py:12-13 and :74-118; the arithmetic is the HIP library
(cmhse_cosine_sim / cmhse_contrastive_fwd: exact-fp32 MFMA similarity + fused hinge reduction).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


def cosine_sim(im, s):
  """/root/reference/loss.py:12-13: im.mm(s.t()) (callers pass L2-normalised rows)."""
  return ops.cosine_sim(im, s)


class _ContrastiveFn(torch.autograd.Function):
  """Forward = HIP similarity + hinge reduction.  Backward is SURVEY.md §8(f) row 1."""

  @staticmethod
  def forward(ctx, im, s, margin, max_violation, norm):
    return ops.contrastive_fwd(im.detach(), s.detach(), margin, max_violation, norm)

  @staticmethod
  def backward(ctx, grad):
    raise NotImplementedError(
        'cmhse_amd: ContrastiveLoss backward is not built yet (SURVEY.md §8(f) row 1)')


class ContrastiveLoss(nn.Module):
  """/root/reference/loss.py:74-118.  measure='order' is a no-op upstream that leaves `sim`
  undefined (loss.py:78-81); only the cosine measure exists."""

  def __init__(self, margin=0, measure=False, max_violation=False, norm=True):
    super(ContrastiveLoss, self).__init__()
    self.margin = margin
    if measure == 'order':
      raise NotImplementedError("measure='order' is undefined in the reference (loss.py:78-81)")
    self.sim = cosine_sim
    self.norm = norm
    self.max_violation = max_violation

  def forward(self, im, s):
    return _ContrastiveFn.apply(im, s, self.margin, self.max_violation, self.norm)
