"""Host-side mirror of the reference's `decoder/` package (reconstruction path).

  Seq2Seq_Decode   /root/reference/decoder/layers.py:12-52   GRU that returns every hidden state
  DecoderSequence  /root/reference/decoder/model.py:16-47    + packing of the valid steps
  EuclideanLoss    /root/reference/decoder/loss.py:12-29

Same names, constructor arguments, parameter names; the arithmetic is the HIP library
(cmhse_gru_pool_fwd/bwd with CMHSE_POOL_ALL, cmhse_euclid_fwd/bwd).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .layers import SeqInput, _GRUPoolBase, _lens_numpy


class Seq2Seq_Decode(_GRUPoolBase):
  """decoder/layers.py:12-52.  forward() returns the packed valid hidden states
  [sum(len), H] (what DecoderSequence builds from the padded tensor, decoder/model.py:39-45)."""
  POOL = ops.POOL_ALL

  def forward(self, q_emb, q_len, hidden=None):
    return self._run(SeqInput('padded', _lens_numpy(q_len), self.POOL), q_emb, hidden, None)

  def forward_repeat(self, rows, counts):
    """Input row s repeated counts[s] times (model.py:261-265) without materialising it."""
    counts = np.asarray(counts, dtype=np.int64)
    return self._run(SeqInput('repeat', counts, self.POOL), rows, None, None)


class DecoderSequence(nn.Module):
  """decoder/model.py:16-47.  Note the upstream call sites pass (embed_size, target_dim), so the
  GRU maps embed_size -> target_dim."""

  def __init__(self, img_dim, embed_size, dropout=0, no_imgnorm=False, bidirectional=False,
               rnn_type='seq2seq'):
    super(DecoderSequence, self).__init__()
    self.embed_size = embed_size
    self.no_imgnorm = no_imgnorm
    self.bidirectional = bidirectional
    self.img_dim = img_dim
    if dropout > 0:
      self.dropout = nn.Dropout(dropout)
    if rnn_type == 'seq2seq':
      self.rnn = Seq2Seq_Decode(img_dim, embed_size, rnn_bidirectional=bidirectional)
    else:
      raise ValueError('Unsupported RNN type')

  def forward(self, x, lengths):
    return self.rnn(x, lengths)

  def forward_repeat(self, rows, counts):
    return self.rnn.forward_repeat(rows, counts)


class _EuclidFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, a, b, b_rows, norm, keepalive):
    ad = a.detach()
    bd = None if b is None else b.detach()
    loss, rows_dev = ops.euclid_fwd(ad, bd, b_rows, norm)
    ctx.save_for_backward(ad)
    ctx.b, ctx.rows_dev, ctx.norm = bd, rows_dev, norm
    # the backward kernel reads the target rows through the raw addresses in `rows_dev`: the
    # storage they point into must live as long as this graph node, not as long as the caller's
    # local variables
    ctx.keepalive = keepalive
    return loss

  @staticmethod
  def backward(ctx, g):
    (a,) = ctx.saved_tensors
    return ops.euclid_bwd(a, ctx.b, ctx.rows_dev, ctx.norm, g), None, None, None, None


class EuclideanLoss(nn.Module):
  """decoder/loss.py:12-29: mean (norm) or sum of the row-wise Euclidean distances."""

  def __init__(self, norm=True):
    super(EuclideanLoss, self).__init__()
    self.norm = norm

  def forward_loss(self, clip_remap, clip_emb):
    return _EuclidFn.apply(clip_remap, clip_emb, None, self.norm, None)

  def forward(self, clip_remap, clip_emb):
    return self.forward_loss(clip_remap, clip_emb)

  def forward_rows(self, clip_remap, target_row_addrs, keepalive):
    """Targets addressed row by row (numpy uint64 device addresses into `keepalive`), e.g. the
    valid frames of a padded clip batch (model.py:350-361) without copying them out."""
    return _EuclidFn.apply(clip_remap, None, target_row_addrs, self.norm, keepalive)
