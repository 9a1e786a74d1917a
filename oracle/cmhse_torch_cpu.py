"""CPU BASELINE in PyTorch for the CMHSE embedding-and-ranking hot path.  TEST INFRASTRUCTURE ONLY.

`BASELINE.json: north_star` names "the reference PyTorch CPU path" as the baseline timed next to
the GPU numbers.  The reference itself cannot travel to the GPU box, so this file restates how the
reference computes the path on a CPU with the same third-party operators it calls — `torch.nn.GRU`
over `pack_padded_sequence`, `pad_packed_sequence`, `nn.Linear`, `F.normalize`, `torch.mm` — on CPU
tensors, one loader batch at a time like evaluation.encode_data does.  It is the second CPU
restatement beside the NumPy oracle (cmhse_oracle.py) and is pinned the same way: against the
reference's own outputs in tests/golden/*.npz (tests/test_oracle_golden.py::test_torch_cpu_*).
Only `tests/` and `bench.py`'s `cpu_baseline` leg import it; the product (`cmhse_amd`) never does.

Cited reference lines are what each function follows; nothing here is reference text.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence


def _gru_module(sd, prefix='rnn.rnn.'):
  """An nn.GRU carrying the checkpoint's weights (layers.py:31-34, 75-79, 169-172)."""
  w_ih = torch.as_tensor(sd[prefix + 'weight_ih_l0'])
  w_hh = torch.as_tensor(sd[prefix + 'weight_hh_l0'])
  gru = torch.nn.GRU(input_size=w_ih.shape[1], hidden_size=w_hh.shape[1], num_layers=1,
                     batch_first=True, bidirectional=False)
  with torch.no_grad():
    gru.weight_ih_l0.copy_(w_ih)
    gru.weight_hh_l0.copy_(w_hh)
    gru.bias_ih_l0.copy_(torch.as_tensor(sd[prefix + 'bias_ih_l0']))
    gru.bias_hh_l0.copy_(torch.as_tensor(sd[prefix + 'bias_hh_l0']))
  gru.eval()
  return gru


class Encoder(object):
  """One encoder layer (layers.Seq2Seq / Attention / Maxout) on CPU tensors."""

  def __init__(self, rnn_type, sd):
    self.rnn_type = rnn_type
    self.gru = _gru_module(sd)
    if rnn_type == 'attention':
      self.w_lin = torch.as_tensor(sd['rnn.lin.weight']).float()
      self.b_lin = torch.as_tensor(sd['rnn.lin.bias']).float()
      self.w_att = torch.as_tensor(sd['rnn.att_w.weight']).float()

  def __call__(self, x, lens, h0=None):
    """x [S, T, I] zero-padded float32, lens [S]; returns [S, H] in input order."""
    lens = torch.as_tensor(np.asarray(lens)).long()
    order = torch.argsort(lens, descending=True, stable=True)    # layers.py:94-96
    inv = torch.empty_like(order)
    inv[order] = torch.arange(len(order))
    packed = pack_padded_sequence(x[order], lens[order].tolist(), batch_first=True)
    hidden = None if h0 is None else h0[order].unsqueeze(0)      # layers.py:98-100
    out, h_n = self.gru(packed, hidden)
    if self.rnn_type == 'seq2seq':                               # layers.py:57-66
      return h_n[0][inv]
    hs, _ = pad_packed_sequence(out, batch_first=True)           # layers.py:103, 195
    ls = lens[order]
    T = hs.shape[1]
    mask = (torch.arange(T)[None, :] < ls[:, None])
    if self.rnn_type == 'maxout':                                # layers.py:196-204
      pooled = hs.masked_fill(~mask[:, :, None], float('-inf')).max(dim=1).values
      return pooled[inv]
    # attention, layers.py:104-117 and the masked softmax of :158-162 (no max-subtraction, +1e-4)
    e = torch.tanh(F.linear(hs, self.w_lin, self.b_lin)) @ self.w_att.reshape(-1)
    ex = torch.exp(e) * mask.float()
    att = ex / (ex.sum(dim=1, keepdim=True) + 0.0001)
    return (att.unsqueeze(2) * hs).sum(dim=1)[inv]


class Model(object):
  """The four encoders of model.VSE (model.py:107-114) from its state-dict list."""

  def __init__(self, rnn_type, sds):
    self.clip_enc = Encoder(rnn_type, sds[0])
    self.txt_enc = Encoder(rnn_type, sds[1])
    self.table = torch.as_tensor(sds[1]['embed.weight']).float()
    self.vid_seq_enc = Encoder(rnn_type, sds[2])
    self.txt_seq_enc = Encoder(rnn_type, sds[3])

  def forward_emb(self, clips, captions, lengths_clip, lengths_cap):
    """model.py:222-236."""
    clip_emb = self.clip_enc(torch.as_tensor(clips).float(), lengths_clip)
    cap_emb = self.txt_enc(self.table[torch.as_tensor(captions).long()], lengths_cap)
    return clip_emb, cap_emb

  @staticmethod
  def _scatter(emb, counts):
    """model.py:239-250: consecutive rows -> zero-padded [B, max(counts), H]."""
    out = torch.zeros(len(counts), max(counts), emb.shape[1])
    pos = 0
    for i, c in enumerate(counts):
      out[i, :c] = emb[pos:pos + c]
      pos += c
    return out

  def structure_emb(self, clip_emb, cap_emb, num_clips, num_caps, vid_ctx=None, para_ctx=None):
    """model.py:238-255."""
    vid = self.vid_seq_enc(self._scatter(clip_emb, num_clips), list(num_clips), vid_ctx)
    par = self.txt_seq_enc(self._scatter(cap_emb, num_caps), list(num_caps), para_ctx)
    return vid, par

  def encode_batch(self, batch, contextual_model=True):
    """evaluation.py:97-116 for one loader batch: the six L2-normalised matrices."""
    clip_emb, cap_emb = self.forward_emb(batch[0], batch[1], batch[4], batch[5])
    vid_ctx, para_ctx = self.forward_emb(batch[2], batch[3], batch[6], batch[7])
    if contextual_model:
      vid, par = self.structure_emb(clip_emb, cap_emb, batch[8], batch[9], vid_ctx, para_ctx)
    else:
      vid, par = self.structure_emb(clip_emb, cap_emb, batch[8], batch[9])
    n = F.normalize
    return n(vid), n(par), n(clip_emb), n(cap_emb), n(vid_ctx), n(para_ctx)


def contrastive_loss(im, s, margin=0.0, max_violation=False, norm=True):
  """loss.ContrastiveLoss.forward, loss.py:86-117."""
  scores = im.mm(s.t())
  diag = scores.diag().view(-1, 1)
  cost_s = (margin + scores - diag.expand_as(scores)).clamp(min=0)
  cost_im = (margin + scores - diag.t().expand_as(scores)).clamp(min=0)
  eye = torch.eye(scores.size(0)) > 0.5
  cost_s = cost_s.masked_fill(eye, 0)
  cost_im = cost_im.masked_fill(eye, 0)
  if max_violation:
    cost_s, cost_im = cost_s.max(1)[0], cost_im.max(0)[0]
  total = cost_s.sum() + cost_im.sum()
  return total / (im.shape[0] * s.shape[0]) if norm else total


def encode_data(rnn_type, sds, batches, margin=0.2, max_violation=False, norm=False,
                contextual_model=True, model=None):
  """evaluation.encode_data (evaluation.py:80-158) on CPU tensors: the reference's 8-tuple (NumPy
  arrays, like the reference's) plus the per-batch 'Letest' losses (:129)."""
  model = model or Model(rnn_type, sds)
  outs = [[] for _ in range(6)]
  num_clips_total, cur_vid_total, losses = [], [], []
  with torch.no_grad():
    for batch in batches:
      embs = model.encode_batch(batch, contextual_model)
      for o, e in zip(outs, embs):
        o.append(e.numpy().copy())               # evaluation.py:111-116 `.data.cpu().numpy().copy()`
      num_clips_total.extend(batch[8])
      cur_vid_total.extend(batch[11])
      losses.append(float(contrastive_loss(embs[0], embs[1], margin, max_violation, norm)))
  cat = [np.concatenate(o, 0) for o in outs]
  return (cat[0], cat[1], cat[2], cat[3], cat[4], cat[5], num_clips_total, cur_vid_total, losses)


def rank_report(queries, gallery):
  """evaluation.i2t / t2i (evaluation.py:160-213): numpy.dot + one argsort per row, as upstream."""
  d = np.dot(queries, gallery.T)
  n = d.shape[0]
  ranks, top1 = np.zeros(n), np.zeros(n)
  for i in range(n):
    inds = np.argsort(d[i])[::-1]
    ranks[i] = np.where(inds == i)[0][0]
    top1[i] = inds[0]
  rep = {k: 100.0 * np.count_nonzero(ranks < c) / n for k, c in (('r1', 1), ('r5', 5), ('r10', 50))}
  rep['medr'] = np.floor(np.median(ranks)) + 1
  rep['meanr'] = ranks.mean() + 1
  rep['sum'] = rep['r1'] + rep['r5'] + rep['r10']
  return rep, top1, ranks


def i2t(images, captions):
  return rank_report(images, captions)


def t2i(images, captions):
  return rank_report(captions, images)
