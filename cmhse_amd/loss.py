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
  """Forward = cmhse_contrastive_fwd (keeps the score matrix), backward = cmhse_contrastive_bwd."""

  @staticmethod
  def forward(ctx, im, s, margin, max_violation, norm, need):
    imd, sd = im.detach(), s.detach()
    if not need:
      return ops.contrastive_fwd(imd, sd, margin, max_violation, norm)
    loss, scores = ops.contrastive_fwd(imd, sd, margin, max_violation, norm, want_scores=True)
    ctx.save_for_backward(imd, sd, scores)
    ctx.cfg = (margin, max_violation, norm)
    return loss

  @staticmethod
  def backward(ctx, grad):
    im, s, scores = ctx.saved_tensors
    margin, max_violation, norm = ctx.cfg
    d_im, d_s = ops.contrastive_bwd(im, s, scores, margin, max_violation, norm, grad)
    return d_im, d_s, None, None, None, None


class _ContrastiveBlocksFn(torch.autograd.Function):
  """Several ContrastiveLoss evaluations in one launch set: block b = rows of two row-blocked
  matrices (cmhse_contrastive_blocks_fwd / _bwd).  Returns the vector of per-block losses."""

  @staticmethod
  def forward(ctx, im, s, sizes, margin, max_violation, norm, need):
    imd, sd = im.detach(), s.detach()
    if not need:
      return ops.contrastive_blocks_fwd(imd, sd, sizes, margin, max_violation, norm)
    losses, state = ops.contrastive_blocks_fwd(imd, sd, sizes, margin, max_violation, norm,
                                               keep=True)
    ctx.save_for_backward(imd, sd)
    ctx.state, ctx.cfg = state, (margin, max_violation, norm)
    return losses

  @staticmethod
  def backward(ctx, grad):
    im, s = ctx.saved_tensors
    margin, max_violation, norm = ctx.cfg
    d_im, d_s = ops.contrastive_blocks_bwd(im, s, ctx.state, margin, max_violation, norm,
                                           grad.contiguous())
    return d_im, d_s, None, None, None, None, None


def contrastive_losses(criterion, pairs):
  """criterion(a_k, b_k) for every pair of a list, as ONE launch set forward and one backward:
  the 4-7 ContrastiveLoss calls of a training step (model.py:333-343) are tiny (n = 32 ... ~130)
  and launch-bound one by one.  Returns the float32 vector of the losses, differentiable wrt every
  a_k / b_k.  Values are those of criterion(a_k, b_k) bit for bit."""
  im = torch.cat([a for a, _ in pairs], 0)
  s = torch.cat([b for _, b in pairs], 0)
  sizes = [int(a.shape[0]) for a, _ in pairs]
  need = torch.is_grad_enabled() and (im.requires_grad or s.requires_grad)
  return _ContrastiveBlocksFn.apply(im, s, sizes, criterion.margin, criterion.max_violation,
                                    criterion.norm, need)


class _StepLossesFn(torch.autograd.Function):
  """normalize + the step's contrastive terms + their weighted total as one node
  (ops.step_losses_fwd / _bwd): two host calls and ten launches for what model.py:333-343 and its
  backward spell as ~40 small operators."""

  @staticmethod
  def forward(ctx, terms, margin, max_violation, norm, *xs):
    values, total, st = ops.step_losses_fwd([x.detach() for x in xs], terms, margin, max_violation,
                                            norm)
    ctx.st = st
    ctx.mark_non_differentiable(values)
    return total.reshape(()), values

  @staticmethod
  def backward(ctx, grad_total, _):
    return (None, None, None, None) + tuple(ops.step_losses_bwd(ctx.st, grad_total))


def step_losses(criterion, xs, terms):
  """(total, values): values[k] = criterion(normalize(xs[a_k]), normalize(xs[b_k])) for
  terms[k] = (a_k, b_k, weight_k), total = sum_k weight_k * values[k], differentiable wrt xs."""
  return _StepLossesFn.apply(tuple(terms), criterion.margin, criterion.max_violation,
                             criterion.norm, *xs)


class _L2NormFn(torch.autograd.Function):
  """F.normalize (model.py:333-343) with its backward on the HIP path."""

  @staticmethod
  def forward(ctx, x):
    xd = x.detach()
    ctx.save_for_backward(xd)
    return ops.l2norm_rows(xd)

  @staticmethod
  def backward(ctx, g):
    (x,) = ctx.saved_tensors
    return ops.l2norm_rows_bwd(x, g)


def normalize(x):
  """torch.nn.functional.normalize(x) for [rows, cols] on the HIP path, differentiable."""
  return _L2NormFn.apply(x)


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
    need = torch.is_grad_enabled() and (im.requires_grad or s.requires_grad)
    return _ContrastiveFn.apply(im, s, self.margin, self.max_violation, self.norm, need)


class _GroupWiseFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, im, s, num_clips, num_caps, margin, max_violation, norm):
    imd = im.detach().float().contiguous()
    sd = s.detach().float().contiguous()
    loss, st = ops.groupwise_fwd(imd, sd, num_clips, num_caps, margin, max_violation, norm)
    ctx.save_for_backward(imd, sd)
    ctx.st, ctx.cfg = st, (margin, max_violation, norm)
    return loss

  @staticmethod
  def backward(ctx, grad):
    im, s = ctx.saved_tensors
    margin, max_violation, norm = ctx.cfg
    d_im, d_s = ops.groupwise_bwd(im, s, ctx.st, margin, max_violation, norm, grad)
    return d_im, d_s, None, None, None, None, None


class GroupWiseContrastiveLoss(nn.Module):
  """/root/reference/loss.py:15-72 (--weak_low_level_loss): clip x caption scores reduced per
  video pair by max (max_violation) or mean, then the contrastive hinge on the [B,B] matrix."""

  def __init__(self, margin=0, measure=False, max_violation=False, norm=True):
    super(GroupWiseContrastiveLoss, self).__init__()
    self.margin = margin
    if measure == 'order':
      raise NotImplementedError("measure='order' is undefined in the reference (loss.py:18-21)")
    self.sim = cosine_sim
    self.norm = norm
    self.max_violation = max_violation

  def forward(self, im, s, num_clips, num_caps):
    return _GroupWiseFn.apply(im, s, [int(c) for c in num_clips], [int(c) for c in num_caps],
                              self.margin, self.max_violation, self.norm)
