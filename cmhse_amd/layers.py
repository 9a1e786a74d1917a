"""Host-side mirror of the reference's `layers.py` operators on the MI355X hot path.

Same class names, constructor arguments, parameter names (checkpoint keys `rnn.weight_ih_l0`,
`lin.weight`, `att_w.weight`, ...) and `forward(q_emb, q_len, hidden=None)` contract as
/root/reference/layers.py:26-66 (Seq2Seq), :69-119 (Attention), :164-204 (Maxout); the body of
each forward is ONE call into the HIP library (cmhse_gru_pool_fwd).  `nn.GRU` / `nn.Linear` are
instantiated only as parameter containers so initialisation and state-dicts match the reference
bit for bit; their own forward is never called.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.init as init

from . import ops


class _PackedGRUPoolFn(torch.autograd.Function):
  """Forward = HIP packed GRU + pooling.  Backward (BPTT) is SURVEY.md §8(f) row 1."""

  @staticmethod
  def forward(ctx, pool_mode, q_emb, lens_np, hidden, w_ih, w_hh, b_ih, b_hh, w_lin, b_lin,
              w_att):
    ops._require_cuda(q_emb, 'q_emb')
    x = q_emb.detach()
    if x.dtype != torch.float32:
      x = x.float()
    x = x.contiguous()
    S, T, I = x.shape
    H = w_hh.shape[1]
    h0_ptrs = None
    if hidden is not None:
      ops._require_cuda(hidden, 'hidden')
      h0 = hidden.detach().float().contiguous()
      h0_ptrs = ops.padded_row_ptrs(h0)
    weights = dict(w_ih=w_ih.detach(), w_hh=w_hh.detach(), b_ih=b_ih.detach(),
                   b_hh=b_hh.detach())
    if pool_mode == ops.POOL_ATTN:
      weights.update(w_lin=w_lin.detach(), b_lin=b_lin.detach(), w_att=w_att.detach().reshape(-1))
    out, fctx = ops.gru_pool_fwd(weights, pool_mode, lens_np, I, H, x.device,
                                 x_ptrs=ops.padded_row_ptrs(x), h0_ptrs=h0_ptrs)
    ctx.fctx = fctx
    return out

  @staticmethod
  def backward(ctx, grad_out):
    raise NotImplementedError(
        'cmhse_amd: BPTT backward of the packed GRU is not built yet (SURVEY.md §8(f) row 1)')


def _lens_numpy(q_len):
  if isinstance(q_len, torch.Tensor):
    return q_len.detach().cpu().numpy().astype(np.int64)   # lengths live on the host (layers.py:97)
  return np.asarray(q_len, dtype=np.int64)


class _GRUPoolBase(nn.Module):
  POOL = None

  def __init__(self, embedding_features, rnn_features, rnn_bidirectional=False):
    super(_GRUPoolBase, self).__init__()
    if rnn_bidirectional:
      raise ValueError('cmhse_amd: bidirectional encoders are not on the reference hot path '
                       '(every reference call site passes bidirectional=False, model.py:107-114)')
    self.bidirectional = rnn_bidirectional
    self.features = rnn_features
    self.rnn = nn.GRU(input_size=embedding_features, hidden_size=rnn_features, num_layers=1,
                      batch_first=True, bidirectional=False)
    self._build_extra(rnn_features)
    self._init_rnn(self.rnn.weight_ih_l0)
    self._init_rnn(self.rnn.weight_hh_l0)
    self.rnn.bias_ih_l0.data.zero_()
    self.rnn.bias_hh_l0.data.zero_()

  def _build_extra(self, rnn_features):
    pass

  def _init_rnn(self, weight):
    # xavier-uniform per gate chunk, layers.py:41-43
    for w in weight.chunk(3, 0):
      init.xavier_uniform_(w)

  def _extra_weights(self):
    return None, None, None

  def forward(self, q_emb, q_len, hidden=None):
    w_lin, b_lin, w_att = self._extra_weights()
    return _PackedGRUPoolFn.apply(self.POOL, q_emb, _lens_numpy(q_len), hidden,
                                  self.rnn.weight_ih_l0, self.rnn.weight_hh_l0,
                                  self.rnn.bias_ih_l0, self.rnn.bias_hh_l0, w_lin, b_lin, w_att)

  def _weights(self):
    w_lin, b_lin, w_att = self._extra_weights()
    weights = dict(w_ih=self.rnn.weight_ih_l0.detach(), w_hh=self.rnn.weight_hh_l0.detach(),
                   b_ih=self.rnn.bias_ih_l0.detach(), b_hh=self.rnn.bias_hh_l0.detach())
    if self.POOL == ops.POOL_ATTN:
      weights.update(w_lin=w_lin.detach(), b_lin=b_lin.detach(), w_att=w_att.detach().reshape(-1))
    return weights

  def forward_ptrs(self, lens, in_dim, device, x_ptrs=None, tok_ptrs=None, table=None,
                   h0_ptrs=None, out=None):
    """Inference-only entry used by the fused paths (EncoderText, structure_emb, encode_data):
    sequences are given by base address (numpy uint64, input order), so padded batches, several
    loader batches at once and consecutive-row level-2 inputs are consumed in place."""
    out, _ = ops.gru_pool_fwd(self._weights(), self.POOL, lens, in_dim,
                              self.rnn.weight_hh_l0.shape[1], device, x_ptrs=x_ptrs,
                              tok_ptrs=tok_ptrs, emb_table=table, h0_ptrs=h0_ptrs, out=out)
    return out

  def forward_tokens(self, tokens, q_len, table):
    """Fused embedding-lookup + encoder (model.EncoderText.forward, model.py:92-99): the word
    vectors are gathered inside the GRU operand load and never materialised."""
    ops._require_cuda(tokens, 'tokens')
    tok = tokens.detach().contiguous()
    if tok.dtype != torch.int64:
      tok = tok.long()
    return self.forward_ptrs(_lens_numpy(q_len), table.shape[1], tok.device,
                             tok_ptrs=ops.padded_row_ptrs(tok), table=table.detach())


class Seq2Seq(_GRUPoolBase):
  """/root/reference/layers.py:26-66 — final hidden state."""
  POOL = ops.POOL_LAST


class Maxout(_GRUPoolBase):
  """/root/reference/layers.py:164-204 — per-sequence max over valid steps."""
  POOL = ops.POOL_MAX


class Attention(_GRUPoolBase):
  """/root/reference/layers.py:69-119 — masked exp-softmax attention pooling."""
  POOL = ops.POOL_ATTN

  def _build_extra(self, rnn_features):
    # construction order matches layers.py:75-82 so a seeded init reproduces the reference's
    self.lin = nn.Linear(rnn_features, rnn_features)
    self.att_w = nn.Linear(rnn_features, 1, bias=False)
    self.tanh = nn.Tanh()

  def _extra_weights(self):
    return self.lin.weight, self.lin.bias, self.att_w.weight
