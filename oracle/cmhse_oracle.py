"""CPU ORACLE for the CMHSE embedding-and-ranking hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-NumPy restatement of the reference's algorithm for the path named in
BASELINE.json (`north_star`) and mapped in SURVEY.md §8(a).  It is the checker: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  The product
(`cmhse_amd`) never imports it and has no CPU fallback.

Parity status: PINNED.  The reference ships no tests or golden vectors of its own,
so the oracle is pinned against outputs of the reference itself, run in the build container by
`tools/make_golden.py` (reference imported read-only from /root/reference with in-memory shims,
torch 2.10.0 CPU / numpy 2.2.6) and committed as `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function below against those vectors.

The arithmetic of the reference lives in un-vendored third-party code:
  * `torch.nn.GRU` (PyTorch >= 0.4 per /root/reference/README.md:10; fixtures made with 2.10.0):
        r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)
        z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
        n = tanh(W_in x + b_in + r * (W_hn h + b_hn))
        h' = (1 - z) * n + z * h            gate row order in weight_ih_l0/weight_hh_l0: r, z, n
  * `torch.nn.functional.normalize`:  x / max(||x||_2, 1e-12)
  * `numpy.dot`, `numpy.argsort`
restated here from their published definitions and anchored on the reference's call sites.

Every function takes `dtype` (default float32, the reference's arithmetic type); float64 gives a
higher-precision yardstick for tolerance tests.
"""
from __future__ import annotations

import numpy as np

POOL_LAST, POOL_ATTN, POOL_MAX = 0, 1, 2
POOL_OF = {'seq2seq': POOL_LAST, 'attention': POOL_ATTN, 'maxout': POOL_MAX}


def _sigmoid(x):
  return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------------------------
# GRU (torch.nn.GRU, 1 layer, unidirectional, batch_first) over ragged sequences.
# Call sites: /root/reference/layers.py:31-34,54-56 (Seq2Seq), :75-79,98-102 (Attention),
# :169-172,192-194 (Maxout).  pack_padded_sequence (layers.py:52,97,190) only reorders the
# sequences; each sequence is independent, so the oracle masks by `t < len` instead of sorting.
# --------------------------------------------------------------------------------------------
def gru_forward(x, lens, w_ih, w_hh, b_ih, b_hh, h0=None, dtype=np.float32):
  """x [S,T,I] zero-padded, lens [S] ints (>=1).  Returns hs [S,T,H] (zeros at t >= len, like
  pad_packed_sequence, layers.py:103) and h_last [S,H] (= h at t = len-1)."""
  x = np.asarray(x, dtype=dtype)
  lens = np.asarray(lens).astype(np.int64)
  w_ih = np.asarray(w_ih, dtype=dtype); w_hh = np.asarray(w_hh, dtype=dtype)
  b_ih = np.asarray(b_ih, dtype=dtype); b_hh = np.asarray(b_hh, dtype=dtype)
  S, T, I = x.shape
  H = w_hh.shape[1]
  Tmax = int(lens.max())
  h = np.zeros((S, H), dtype=dtype) if h0 is None else np.array(h0, dtype=dtype)
  hs = np.zeros((S, T, H), dtype=dtype)
  # input projection for every (s,t) at once: gi = x W_ih^T + b_ih
  gi = (x[:, :Tmax].reshape(S * Tmax, I) @ w_ih.T + b_ih).reshape(S, Tmax, 3 * H)
  for t in range(Tmax):
    act = np.nonzero(lens > t)[0]
    gh = h[act] @ w_hh.T + b_hh
    g = gi[act, t]
    r = _sigmoid(g[:, :H] + gh[:, :H])
    z = _sigmoid(g[:, H:2 * H] + gh[:, H:2 * H])
    n = np.tanh(g[:, 2 * H:] + r * gh[:, 2 * H:])
    hn = ((1.0 - z) * n + z * h[act]).astype(dtype)
    h[act] = hn
    hs[act, t] = hn
  return hs, h


def seq2seq_forward(x, lens, p, h0=None, dtype=np.float32):
  """layers.Seq2Seq.forward, /root/reference/layers.py:47-66: final hidden state."""
  _, h_last = gru_forward(x, lens, p['rnn.rnn.weight_ih_l0'], p['rnn.rnn.weight_hh_l0'],
                          p['rnn.rnn.bias_ih_l0'], p['rnn.rnn.bias_hh_l0'], h0, dtype)
  return h_last


def maxout_forward(x, lens, p, h0=None, dtype=np.float32):
  """layers.Maxout.forward, /root/reference/layers.py:185-204: per-sequence max over valid
  steps (F.max_pool1d over outputs[i,:len_i], :201-202)."""
  hs, _ = gru_forward(x, lens, p['rnn.rnn.weight_ih_l0'], p['rnn.rnn.weight_hh_l0'],
                      p['rnn.rnn.bias_ih_l0'], p['rnn.rnn.bias_hh_l0'], h0, dtype)
  lens = np.asarray(lens).astype(np.int64)
  T = hs.shape[1]
  mask = (np.arange(T)[None, :] < lens[:, None])[:, :, None]
  return np.where(mask, hs, -np.inf).max(axis=1).astype(dtype)


def attention_energies(hs, p, dtype=np.float32):
  """e = att_w(tanh(lin(h))), /root/reference/layers.py:105-106."""
  S, T, H = hs.shape
  w_lin = np.asarray(p['rnn.lin.weight'], dtype=dtype)
  b_lin = np.asarray(p['rnn.lin.bias'], dtype=dtype)
  w_att = np.asarray(p['rnn.att_w.weight'], dtype=dtype).reshape(-1)
  emb_h = np.tanh(hs.reshape(S * T, H) @ w_lin.T + b_lin)
  return (emb_h @ w_att).reshape(S, T).astype(dtype)


def attention_forward(x, lens, p, h0=None, dtype=np.float32):
  """layers.Attention.forward, /root/reference/layers.py:93-119.
  Masked softmax without max-subtraction and with +1e-4 in the denominator (:158-162)."""
  hs, _ = gru_forward(x, lens, p['rnn.rnn.weight_ih_l0'], p['rnn.rnn.weight_hh_l0'],
                      p['rnn.rnn.bias_ih_l0'], p['rnn.rnn.bias_hh_l0'], h0, dtype)
  lens = np.asarray(lens).astype(np.int64)
  Tmax = int(lens.max())
  hs = hs[:, :Tmax]                      # pad_packed_sequence pads to the batch maximum
  e = attention_energies(hs, p, dtype)
  mask = (np.arange(Tmax)[None, :] < lens[:, None]).astype(dtype)
  ex = np.exp(e) * mask
  att = ex / (ex.sum(axis=1, keepdims=True) + dtype(0.0001))
  return (att[:, :, None] * hs).sum(axis=1).astype(dtype)


def pooled_gru_forward(rnn_type, x, lens, p, h0=None, dtype=np.float32):
  """Dispatch on `rnn_type` as model.EncoderImage/EncoderSequence do,
  /root/reference/model.py:27-34,50-57."""
  if rnn_type == 'attention':
    return attention_forward(x, lens, p, h0, dtype)
  if rnn_type == 'seq2seq':
    return seq2seq_forward(x, lens, p, h0, dtype)
  if rnn_type == 'maxout':
    return maxout_forward(x, lens, p, h0, dtype)
  raise ValueError('Unsupported RNN type')    # model.py:34


def encoder_text_forward(rnn_type, tokens, lens, p, dtype=np.float32):
  """model.EncoderText.forward, /root/reference/model.py:92-99: embed(x) then rnn.
  Returns (outputs, word_embeddings)."""
  table = np.asarray(p['embed.weight'], dtype=dtype)
  cap_emb = table[np.asarray(tokens)]
  return pooled_gru_forward(rnn_type, cap_emb, lens, p, None, dtype), cap_emb


def l2_normalize(x, dtype=np.float32):
  """torch.nn.functional.normalize(x) (p=2, dim=1, eps=1e-12); call sites
  /root/reference/model.py:333-343, evaluation.py:111-116."""
  x = np.asarray(x, dtype=dtype)
  nrm = np.sqrt((x * x).sum(axis=1, keepdims=True))
  return (x / np.maximum(nrm, dtype(1e-12))).astype(dtype)


# --------------------------------------------------------------------------------------------
# Similarity + contrastive loss, /root/reference/loss.py:12-13, 74-118
# --------------------------------------------------------------------------------------------
def cosine_sim(im, s, dtype=np.float32):
  """loss.cosine_sim, /root/reference/loss.py:12-13 (inputs are pre-normalised by callers)."""
  return np.asarray(im, dtype=dtype) @ np.asarray(s, dtype=dtype).T


def contrastive_loss(im, s, margin=0.0, max_violation=False, norm=True, dtype=np.float32):
  """loss.ContrastiveLoss.forward, /root/reference/loss.py:86-117."""
  scores = cosine_sim(im, s, dtype)
  n, m = scores.shape
  diag = np.diag(scores).reshape(n, 1)
  cost_s = np.maximum(dtype(margin) + scores - diag, 0)         # :94  row-wise (caption retrieval)
  cost_im = np.maximum(dtype(margin) + scores - diag.T, 0)      # :97  column-wise (image retrieval)
  eye = np.eye(n, dtype=bool)
  cost_s = np.where(eye, 0, cost_s)                             # :100-106
  cost_im = np.where(eye, 0, cost_im)
  if max_violation:                                             # :109-111
    cost_s = cost_s.max(axis=1)
    cost_im = cost_im.max(axis=0)
  loss = cost_s.sum(dtype=dtype) + cost_im.sum(dtype=dtype)
  if norm:                                                      # :114-115  divides by n*m
    loss = loss / dtype(n * m)
  return dtype(loss)


def euclidean_loss(a, b, norm=True, dtype=np.float32):
  """decoder.loss.EuclideanLoss, /root/reference/decoder/loss.py:17-26."""
  d = np.asarray(a, dtype=dtype) - np.asarray(b, dtype=dtype)
  sub = np.sqrt((d * d).sum(axis=1))
  return dtype(sub.mean() if norm else sub.sum())


# --------------------------------------------------------------------------------------------
# Ranking, /root/reference/evaluation.py:160-213
# --------------------------------------------------------------------------------------------
def rank_rows(d):
  """Per-row rank of the diagonal element and arg-max column.

  The reference sorts each row with `numpy.argsort(d[i])[::-1]` (evaluation.py:167,195), which is
  only well-defined when no score ties with d[i,i]; on such rows it equals
      rank_i = #{ j != i : d_ij > d_ii },   top1_i = argmax_j d_ij.
  Documented tie rule of the build (ties are implementation-defined upstream):
  strict '>' for the rank, smallest column index for top1."""
  n = d.shape[0]
  diag = d[np.arange(n), np.arange(n)][:, None]
  gt = d > diag
  gt[np.arange(n), np.arange(n)] = False
  return gt.sum(axis=1).astype(np.int64), d.argmax(axis=1).astype(np.int64)


def recall_report(ranks):
  """evaluation.py:173-184: note 'r10' is Recall@50 (`ranks < 50`, :175)."""
  ranks = np.asarray(ranks, dtype=np.float64)
  r1 = 100.0 * len(np.where(ranks < 1)[0]) / len(ranks)
  r5 = 100.0 * len(np.where(ranks < 5)[0]) / len(ranks)
  r10 = 100.0 * len(np.where(ranks < 50)[0]) / len(ranks)
  medr = np.floor(np.median(ranks)) + 1
  meanr = ranks.mean() + 1
  return {'r1': r1, 'r5': r5, 'r10': r10, 'medr': medr, 'meanr': meanr, 'sum': r1 + r5 + r10}


def i2t(images, captions, dtype=np.float32):
  """evaluation.i2t, /root/reference/evaluation.py:160-185."""
  d = np.dot(np.asarray(images, dtype=dtype), np.asarray(captions, dtype=dtype).T)
  ranks, top1 = rank_rows(d)
  return recall_report(ranks), top1.astype(np.float64), ranks.astype(np.float64)


def t2i(images, captions, dtype=np.float32):
  """evaluation.t2i, /root/reference/evaluation.py:188-213."""
  d = np.dot(np.asarray(captions, dtype=dtype), np.asarray(images, dtype=dtype).T)
  ranks, top1 = rank_rows(d)
  return recall_report(ranks), top1.astype(np.float64), ranks.astype(np.float64)


# --------------------------------------------------------------------------------------------
# Model orchestration, /root/reference/model.py:222-255 and evaluation.py:80-158
# `params` is the reference's state-dict list [clip_enc, txt_enc, vid_seq_enc, txt_seq_enc]
# (model.py:166-168) with numpy values.
# --------------------------------------------------------------------------------------------
def forward_emb(rnn_type, params, clips, captions, lengths_clip, lengths_cap, dtype=np.float32):
  """VSE.forward_emb, /root/reference/model.py:222-236.  Returns (clip_emb, cap_emb, word)."""
  clip_emb = pooled_gru_forward(rnn_type, clips, lengths_clip, params[0], None, dtype)
  cap_emb, word = encoder_text_forward(rnn_type, captions, lengths_cap, params[1], dtype)
  return clip_emb, cap_emb, word


def scatter_rows(emb, counts, dtype=np.float32):
  """model.py:239-250: consecutive rows of `emb` -> zero-padded [B, max(counts), H]."""
  B, H = len(counts), emb.shape[1]
  out = np.zeros((B, max(counts), H), dtype=dtype)
  pos = 0
  for i, c in enumerate(counts):
    out[i, :c] = emb[pos:pos + c]
    pos += c
  return out


def structure_emb(rnn_type, params, clip_emb, cap_emb, num_clips, num_caps,
                  vid_context=None, para_context=None, dtype=np.float32):
  """VSE.structure_emb, /root/reference/model.py:238-255: level-2 GRUs with h0 = context."""
  x_v = scatter_rows(clip_emb, num_clips, dtype)
  x_p = scatter_rows(cap_emb, num_caps, dtype)
  vid_emb = pooled_gru_forward(rnn_type, x_v, num_clips, params[2], vid_context, dtype)
  para_emb = pooled_gru_forward(rnn_type, x_p, num_caps, params[3], para_context, dtype)
  return vid_emb, para_emb


def encode_batch(rnn_type, params, batch, contextual_model=True, dtype=np.float32):
  """One loader batch of evaluation.encode_data, /root/reference/evaluation.py:97-116.
  Returns the six L2-normalised embedding matrices in the order encode_data returns them."""
  (clips, captions, videos, paragraphs, lengths_clip, lengths_cap, lengths_video,
   lengths_paragraph, num_clips, num_caps) = [np.asarray(b) if not isinstance(b, tuple) else b
                                              for b in batch[:10]]
  clip_emb, cap_emb, _ = forward_emb(rnn_type, params, clips, captions, lengths_clip,
                                     lengths_cap, dtype)
  vid_ctx, para_ctx, _ = forward_emb(rnn_type, params, videos, paragraphs, lengths_video,
                                     lengths_paragraph, dtype)
  if contextual_model:
    vid_emb, para_emb = structure_emb(rnn_type, params, clip_emb, cap_emb, num_clips, num_caps,
                                      vid_ctx, para_ctx, dtype)
  else:
    vid_emb, para_emb = structure_emb(rnn_type, params, clip_emb, cap_emb, num_clips, num_caps,
                                      None, None, dtype)
  n = lambda a: l2_normalize(a, dtype)
  return n(vid_emb), n(para_emb), n(clip_emb), n(cap_emb), n(vid_ctx), n(para_ctx)


def encode_data(rnn_type, params, batches, margin=0.2, max_violation=False, norm=False,
                contextual_model=True, dtype=np.float32):
  """evaluation.encode_data, /root/reference/evaluation.py:80-158.
  Returns the reference's 8-tuple plus the list of per-batch 'Letest' losses (:129)."""
  outs = [[] for _ in range(6)]
  num_clips_total, cur_vid_total, test_losses = [], [], []
  for batch in batches:
    embs = encode_batch(rnn_type, params, batch, contextual_model, dtype)
    for o, e in zip(outs, embs):
      o.append(e)
    num_clips_total.extend(batch[8])
    cur_vid_total.extend(batch[11])
    test_losses.append(float(contrastive_loss(embs[0], embs[1], margin, max_violation, norm,
                                              dtype)))
  cat = [np.concatenate(o, 0) for o in outs]
  return (cat[0], cat[1], cat[2], cat[3], cat[4], cat[5], num_clips_total, cur_vid_total,
          test_losses)


def train_losses(rnn_type, params, batch, margin=0.2, max_violation=False, norm=False,
                 low_level_loss=False, dtype=np.float32):
  """Forward half of VSE.train_emb, /root/reference/model.py:319-343 (no reconstruction, no weak
  loss): the (name, value, n) triples `forward_loss` sends to the logger (model.py:291), in call
  order, and the total loss (model.py:336,344)."""
  (clips, captions, videos, paragraphs, lengths_clip, lengths_cap, lengths_video,
   lengths_paragraph, num_clips, num_caps) = batch[:10]
  clip_emb, cap_emb, _ = forward_emb(rnn_type, params, clips, captions, lengths_clip,
                                     lengths_cap, dtype)
  vid_ctx, para_ctx, _ = forward_emb(rnn_type, params, videos, paragraphs, lengths_video,
                                     lengths_paragraph, dtype)
  vid_emb, para_emb = structure_emb(rnn_type, params, clip_emb, cap_emb, num_clips, num_caps,
                                    vid_ctx, para_ctx, dtype)
  n = lambda a: l2_normalize(a, dtype)
  cl = lambda a, b: contrastive_loss(a, b, margin, max_violation, norm, dtype)
  log = []

  def fl(a, b, name):
    v = cl(a, b)
    log.append(('Le' + name, float(v), a.shape[0]))
    return v

  loss_1 = fl(n(vid_emb), n(para_emb), '_vid')
  loss_3 = fl(n(vid_ctx), n(para_ctx), '_ctx_low_lvel')
  loss_5 = (fl(n(vid_emb), n(vid_emb), '_vid_inloss') +
            fl(n(para_emb), n(para_emb), '_para_inloss')) / 2
  loss = loss_1 + loss_3 + loss_5
  if low_level_loss:
    loss_2 = fl(n(clip_emb), n(cap_emb), '_low_lvel')
    loss_6 = (fl(n(clip_emb), n(clip_emb), '_clip_inloss') +
              fl(n(cap_emb), n(cap_emb), '_cap_inloss')) / 2
    loss = loss + loss_2 + loss_6
  return log, float(loss)


# --------------------------------------------------------------------------------------------
# Backward pass.  The reference gets its gradients from torch.autograd
# (`loss.backward()`, /root/reference/model.py:367); these are the same derivatives written out,
# pinned against gradients of the reference itself (tests/golden: `grad.*`, `bwd.*`).
# --------------------------------------------------------------------------------------------
def gru_forward_cache(x, lens, p, h0=None, dtype=np.float64):
  """gru_forward that also keeps what BPTT needs."""
  x = np.asarray(x, dtype=dtype)
  lens = np.asarray(lens).astype(np.int64)
  w_ih = np.asarray(p['rnn.rnn.weight_ih_l0'], dtype=dtype)
  w_hh = np.asarray(p['rnn.rnn.weight_hh_l0'], dtype=dtype)
  b_ih = np.asarray(p['rnn.rnn.bias_ih_l0'], dtype=dtype)
  b_hh = np.asarray(p['rnn.rnn.bias_hh_l0'], dtype=dtype)
  S, T, I = x.shape
  H = w_hh.shape[1]
  Tmax = int(lens.max())
  h = np.zeros((S, H), dtype=dtype) if h0 is None else np.array(h0, dtype=dtype)
  hs = np.zeros((S, Tmax, H), dtype=dtype)
  hprev = np.zeros((S, Tmax, H), dtype=dtype)
  gates = np.zeros((S, Tmax, 4, H), dtype=dtype)      # r, z, n, (W_hn h + b_hn)
  for t in range(Tmax):
    act = lens > t
    gi = x[:, t] @ w_ih.T + b_ih
    gh = h @ w_hh.T + b_hh
    r = _sigmoid(gi[:, :H] + gh[:, :H])
    z = _sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    hn = (1.0 - z) * n + z * h
    hprev[:, t] = h
    gates[:, t, 0], gates[:, t, 1], gates[:, t, 2], gates[:, t, 3] = r, z, n, gh[:, 2 * H:]
    h = np.where(act[:, None], hn, h)
    hs[:, t] = np.where(act[:, None], hn, 0)
  return dict(x=x, lens=lens, hs=hs, hprev=hprev, gates=gates, w_ih=w_ih, w_hh=w_hh, H=H, I=I,
              Tmax=Tmax, has_h0=h0 is not None)


def gru_backward(c, dhs):
  """BPTT.  dhs [S,Tmax,H]: gradient arriving at every hidden state from the pooling.
  Returns (param grads with reference keys, dx [S,Tmax,I], dh0 [S,H])."""
  S, Tmax, H, I = c['hs'].shape[0], c['Tmax'], c['H'], c['I']
  dt = c['hs'].dtype
  dw_ih = np.zeros_like(c['w_ih']); dw_hh = np.zeros_like(c['w_hh'])
  db_ih = np.zeros(3 * H, dtype=dt); db_hh = np.zeros(3 * H, dtype=dt)
  dx = np.zeros((S, Tmax, I), dtype=dt)
  dh = np.zeros((S, H), dtype=dt)
  for t in range(Tmax - 1, -1, -1):
    act = (c['lens'] > t)[:, None]
    dh = np.where(act, dh + dhs[:, t], dh)        # finished sequences carry nothing back
    r, z, n, ghn = (c['gates'][:, t, k] for k in range(4))
    hp = c['hprev'][:, t]
    dn = dh * (1.0 - z)
    dz = dh * (hp - n)
    dn_pre = dn * (1.0 - n * n)
    dr = dn_pre * ghn
    dr_pre = dr * r * (1.0 - r)
    dz_pre = dz * z * (1.0 - z)
    dgx = np.where(act, np.concatenate([dr_pre, dz_pre, dn_pre], 1), 0)
    dgh = np.where(act, np.concatenate([dr_pre, dz_pre, dn_pre * r], 1), 0)
    dw_ih += dgx.T @ c['x'][:, t]
    dw_hh += dgh.T @ hp
    db_ih += dgx.sum(0)
    db_hh += dgh.sum(0)
    dx[:, t] = dgx @ c['w_ih']
    dh = np.where(act, dh * z + dgh @ c['w_hh'], dh)
  grads = {'rnn.rnn.weight_ih_l0': dw_ih, 'rnn.rnn.weight_hh_l0': dw_hh,
           'rnn.rnn.bias_ih_l0': db_ih, 'rnn.rnn.bias_hh_l0': db_hh}
  return grads, dx, dh


def pooled_gru_forward_cache(rnn_type, x, lens, p, h0=None, dtype=np.float64):
  c = gru_forward_cache(x, lens, p, h0, dtype)
  c['rnn_type'] = rnn_type
  hs, lens = c['hs'], c['lens']
  S, Tmax, H = hs.shape
  mask = np.arange(Tmax)[None, :] < lens[:, None]
  if rnn_type == 'seq2seq':
    out = hs[np.arange(S), lens - 1]
  elif rnn_type == 'maxout':
    masked = np.where(mask[:, :, None], hs, -np.inf)
    c['argmax'] = masked.argmax(axis=1)          # first maximum on ties
    out = masked.max(axis=1)
  elif rnn_type == 'attention':
    w_lin = np.asarray(p['rnn.lin.weight'], dtype=dtype)
    b_lin = np.asarray(p['rnn.lin.bias'], dtype=dtype)
    w_att = np.asarray(p['rnn.att_w.weight'], dtype=dtype).reshape(-1)
    v = np.tanh(hs @ w_lin.T + b_lin)
    e = v @ w_att
    ex = np.exp(e) * mask
    att = ex / (ex.sum(axis=1, keepdims=True) + 1e-4)
    c.update(v=v, att=att, w_lin=w_lin, w_att=w_att, mask=mask)
    out = (att[:, :, None] * hs).sum(axis=1)
  else:
    raise ValueError('Unsupported RNN type')
  return out.astype(dtype), c


def pooled_gru_backward(c, dout):
  """Gradient of pooled_gru_forward wrt parameters, inputs and h0 given d(out) [S,H]."""
  hs, lens = c['hs'], c['lens']
  S, Tmax, H = hs.shape
  dhs = np.zeros_like(hs)
  extra = {}
  if c['rnn_type'] == 'seq2seq':
    dhs[np.arange(S), lens - 1] = dout
  elif c['rnn_type'] == 'maxout':
    s_idx, u_idx = np.meshgrid(np.arange(S), np.arange(H), indexing='ij')
    dhs[s_idx, c['argmax'], u_idx] = dout
  else:
    att, v, mask = c['att'], c['v'], c['mask']
    da = (hs * dout[:, None, :]).sum(axis=2)                 # [S,T]
    de = att * (da - (att * da).sum(axis=1, keepdims=True))  # softmax-with-eps Jacobian
    de = de * mask
    dv = de[:, :, None] * c['w_att'][None, None, :]
    du = dv * (1.0 - v * v)
    extra['rnn.att_w.weight'] = (de[:, :, None] * v).sum(axis=(0, 1)).reshape(1, H)
    extra['rnn.lin.weight'] = du.reshape(S * Tmax, H).T @ hs.reshape(S * Tmax, H)
    extra['rnn.lin.bias'] = du.sum(axis=(0, 1))
    dhs = att[:, :, None] * dout[:, None, :] + du @ c['w_lin']
    dhs = dhs * mask[:, :, None]
  grads, dx, dh0 = gru_backward(c, dhs)
  grads.update(extra)
  return grads, dx, dh0


def l2_normalize_backward(x, g, dtype=np.float64):
  x = np.asarray(x, dtype=dtype); g = np.asarray(g, dtype=dtype)
  nrm = np.maximum(np.sqrt((x * x).sum(axis=1, keepdims=True)), 1e-12)
  y = x / nrm
  return (g - y * (y * g).sum(axis=1, keepdims=True)) / nrm


def contrastive_loss_backward(im, s, margin=0.0, max_violation=False, norm=True, dtype=np.float64):
  """d loss / d im, d loss / d s of contrastive_loss (upstream gradient 1)."""
  im = np.asarray(im, dtype=dtype); s = np.asarray(s, dtype=dtype)
  scores = im @ s.T
  n = scores.shape[0]
  diag = np.diag(scores).reshape(n, 1)
  eye = np.eye(n, dtype=bool)
  cost_s = np.where(eye, 0, np.maximum(margin + scores - diag, 0))
  cost_im = np.where(eye, 0, np.maximum(margin + scores - diag.T, 0))
  if max_violation:
    g_s = np.zeros_like(scores); g_im = np.zeros_like(scores)
    js = cost_s.argmax(axis=1)
    g_s[np.arange(n), js] = (cost_s[np.arange(n), js] > 0)
    is_ = cost_im.argmax(axis=0)
    g_im[is_, np.arange(n)] = (cost_im[is_, np.arange(n)] > 0)
  else:
    g_s = (cost_s > 0).astype(dtype); g_im = (cost_im > 0).astype(dtype)
  G = g_s + g_im
  G[np.arange(n), np.arange(n)] -= g_s.sum(axis=1) + g_im.sum(axis=0)
  if norm:
    G = G / (n * n)
  return G @ s, G.T @ im


def apply_argmax_route(caches, argmax_route, report=None):
  """Max pooling sends each output's gradient to ONE step, the arg-max — a discrete choice.  Where
  two steps of a (sequence, unit) pair tie to within fp32 rounding, an fp32 forward and this fp64
  one may choose differently, and the parameter gradients then differ by a routing change, not by
  an arithmetic error.  To compare gradients element-wise, a test hands over the routing the fp32
  forward actually used: `argmax_route[name]` = int array [S, H] (input order) for name in
  `caches` ('clip', 'vid', 'cap', 'par', 'v2', 'p2').  The backward pass below then routes through
  it.  `report[name]` receives (number of pairs routed differently from this forward's own
  arg-max, the largest fp64 gap hs[own] - hs[routed] among them, number of pairs): a routing that
  differs anywhere but at a near-tie shows up as a large gap."""
  if not argmax_route:
    return
  for name, c in caches.items():
    if c.get('rnn_type') != 'maxout' or name not in argmax_route:
      continue
    own = c['argmax']
    route = np.asarray(argmax_route[name]).astype(own.dtype)
    if route.shape != own.shape:
      raise ValueError('argmax_route[%s]: shape %s, want %s' % (name, route.shape, own.shape))
    if (route < 0).any() or (route >= c['lens'][:, None]).any():
      raise ValueError('argmax_route[%s]: a step outside its sequence' % name)
    diff = route != own
    if report is not None:
      s_idx, u_idx = np.nonzero(diff)
      gap = 0.0
      if len(s_idx):
        gap = float((c['hs'][s_idx, own[s_idx, u_idx], u_idx] -
                     c['hs'][s_idx, route[s_idx, u_idx], u_idx]).max())
      report[name] = (int(diff.sum()), gap, int(diff.size))
    c['argmax'] = route


def train_step_grads(rnn_type, params, batch, margin=0.2, max_violation=False, norm=False,
                     low_level_loss=False, dtype=np.float64, argmax_route=None, route_report=None):
  """Parameter gradients of the total loss of VSE.train_emb (model.py:319-344, no reconstruction):
  a list of four dicts keyed like the reference's state-dicts."""
  (clips, captions, videos, paragraphs, lengths_clip, lengths_cap, lengths_video,
   lengths_paragraph, num_clips, num_caps) = batch[:10]
  table = np.asarray(params[1]['embed.weight'], dtype=dtype)
  fw = lambda p, x, l, h0=None: pooled_gru_forward_cache(rnn_type, x, l, p, h0, dtype)
  clip_emb, c_clip = fw(params[0], clips, lengths_clip)
  cap_emb, c_cap = fw(params[1], table[np.asarray(captions)], lengths_cap)
  vid_ctx, c_vid = fw(params[0], videos, lengths_video)
  para_ctx, c_par = fw(params[1], table[np.asarray(paragraphs)], lengths_paragraph)
  x_v = scatter_rows(clip_emb, num_clips, dtype)
  x_p = scatter_rows(cap_emb, num_caps, dtype)
  vid_emb, c_v2 = fw(params[2], x_v, num_clips, vid_ctx)
  para_emb, c_p2 = fw(params[3], x_p, num_caps, para_ctx)
  apply_argmax_route(dict(clip=c_clip, cap=c_cap, vid=c_vid, par=c_par, v2=c_v2, p2=c_p2),
                     argmax_route, route_report)

  d = {k: 0.0 for k in ['vid', 'para', 'vctx', 'pctx', 'clip', 'cap']}
  raw = dict(vid=vid_emb, para=para_emb, vctx=vid_ctx, pctx=para_ctx, clip=clip_emb, cap=cap_emb)

  def add_loss(a, b, scale):
    na, nb = l2_normalize(raw[a], dtype), l2_normalize(raw[b], dtype)
    ga, gb = contrastive_loss_backward(na, nb, margin, max_violation, norm, dtype)
    d[a] = d[a] + scale * l2_normalize_backward(raw[a], ga, dtype)
    d[b] = d[b] + scale * l2_normalize_backward(raw[b], gb, dtype)

  add_loss('vid', 'para', 1.0)
  add_loss('vctx', 'pctx', 1.0)
  add_loss('vid', 'vid', 0.5)
  add_loss('para', 'para', 0.5)
  if low_level_loss:
    add_loss('clip', 'cap', 1.0)
    add_loss('clip', 'clip', 0.5)
    add_loss('cap', 'cap', 0.5)

  def gather_rows_grad(dxp, counts):
    return np.concatenate([dxp[i, :c] for i, c in enumerate(counts)], 0)

  g_v2, dx_v2, dh0_v2 = pooled_gru_backward(c_v2, d['vid'])
  g_p2, dx_p2, dh0_p2 = pooled_gru_backward(c_p2, d['para'])
  d_clip = d['clip'] + gather_rows_grad(dx_v2, num_clips)
  d_cap = d['cap'] + gather_rows_grad(dx_p2, num_caps)
  g_clip, _, _ = pooled_gru_backward(c_clip, d_clip)
  g_vid, _, _ = pooled_gru_backward(c_vid, d['vctx'] + dh0_v2)
  g_cap, dx_cap, _ = pooled_gru_backward(c_cap, d_cap)
  g_par, dx_par, _ = pooled_gru_backward(c_par, d['pctx'] + dh0_p2)
  g0 = {k: g_clip[k] + g_vid[k] for k in g_clip}
  g1 = {k: g_cap[k] + g_par[k] for k in g_cap}
  dtable = np.zeros_like(table)
  for toks, lens, dxx in [(captions, lengths_cap, dx_cap), (paragraphs, lengths_paragraph, dx_par)]:
    toks = np.asarray(toks)
    for i, l in enumerate(np.asarray(lens)):
      np.add.at(dtable, toks[i, :l], dxx[i, :l])
  g1['embed.weight'] = dtable
  return [g0, g1, g_v2, g_p2]


# --------------------------------------------------------------------------------------------
# Reconstruction path: decoder/model.py:16-47 DecoderSequence,
# decoder/layers.py:34-52 Seq2Seq_Decode, decoder/loss.py:17-26 EuclideanLoss,
# model.py:257-285 reconstruct_emb / lowest_reconstruct_emb, model.py:346-364 loss composition.
# --------------------------------------------------------------------------------------------
def decoder_forward_cache(rows, counts, p, dtype=np.float64):
  """DecoderSequence.forward on `rows[i]` repeated counts[i] times (model.py:261-268): a GRU with a
  time-constant input, no h0; returns all hidden states concatenated sequence by sequence
  [sum(counts), H_dec] (decoder/model.py:39-45) and the BPTT cache."""
  rows = np.asarray(rows, dtype=dtype)
  counts = np.asarray(counts).astype(np.int64)
  B, T = len(counts), int(counts.max())
  x = np.zeros((B, T, rows.shape[1]), dtype=dtype)
  for i, c in enumerate(counts):
    x[i, :c] = rows[i]
  c = gru_forward_cache(x, counts, p, None, dtype)
  out = np.concatenate([c['hs'][i, :n] for i, n in enumerate(counts)], 0)
  c['counts'] = counts
  return out, c


def decoder_backward(c, dout):
  counts = c['counts']
  dhs = np.zeros_like(c['hs'])
  pos = 0
  for i, n in enumerate(counts):
    dhs[i, :n] = dout[pos:pos + n]
    pos += n
  grads, dx, _ = gru_backward(c, dhs)
  drows = np.stack([dx[i, :n].sum(0) for i, n in enumerate(counts)], 0)
  return grads, drows


def euclidean_loss_backward(a, b, norm=True, dtype=np.float64):
  """d EuclideanLoss(a, b) / d a  (b is detached upstream, model.py:347,363)."""
  d = np.asarray(a, dtype=dtype) - np.asarray(b, dtype=dtype)
  sub = np.sqrt((d * d).sum(axis=1, keepdims=True))
  g = d / sub
  return g / d.shape[0] if norm else g


def train_step_recon(rnn_type, params, batch, margin=0.2, max_violation=False, norm=False,
                     low_level_loss=False, lowest=False, weight_recon=0.0005,
                     lowest_weight_recon=0.0001, dtype=np.float64, argmax_route=None,
                     route_report=None):
  """VSE.train_emb with --reconstruct_loss (and optionally --lowest_reconstruct_loss),
  model.py:319-364.  `params`: the 6 (or 8) state-dicts.  Returns (logger triples, total loss,
  list of per-module gradient dicts)."""
  (clips, captions, videos, paragraphs, lengths_clip, lengths_cap, lengths_video,
   lengths_paragraph, num_clips, num_caps) = batch[:10]
  table = np.asarray(params[1]['embed.weight'], dtype=dtype)
  fw = lambda p, x, l, h0=None: pooled_gru_forward_cache(rnn_type, x, l, p, h0, dtype)
  clip_emb, c_clip = fw(params[0], clips, lengths_clip)
  word = table[np.asarray(captions)]
  cap_emb, c_cap = fw(params[1], word, lengths_cap)
  vid_ctx, c_vid = fw(params[0], videos, lengths_video)
  para_ctx, c_par = fw(params[1], table[np.asarray(paragraphs)], lengths_paragraph)
  vid_emb, c_v2 = fw(params[2], scatter_rows(clip_emb, num_clips, dtype), num_clips, vid_ctx)
  para_emb, c_p2 = fw(params[3], scatter_rows(cap_emb, num_caps, dtype), num_caps, para_ctx)
  apply_argmax_route(dict(clip=c_clip, cap=c_cap, vid=c_vid, par=c_par, v2=c_v2, p2=c_p2),
                     argmax_route, route_report)
  clip_recon, d_vdec = decoder_forward_cache(vid_emb, num_clips, params[4], dtype)
  cap_recon, d_tdec = decoder_forward_cache(para_emb, num_caps, params[5], dtype)
  if lowest:
    frame_recon, d_cdec = decoder_forward_cache(clip_recon, lengths_clip, params[6], dtype)
    sent_recon, d_sdec = decoder_forward_cache(cap_recon, lengths_cap, params[7], dtype)

  log = []
  d = {k: 0.0 for k in ['vid', 'para', 'vctx', 'pctx', 'clip', 'cap']}
  raw = dict(vid=vid_emb, para=para_emb, vctx=vid_ctx, pctx=para_ctx, clip=clip_emb, cap=cap_emb)
  total = 0.0

  def add_loss(a, b, scale, name):
    nonlocal total
    na, nb = l2_normalize(raw[a], dtype), l2_normalize(raw[b], dtype)
    v = float(contrastive_loss(na, nb, margin, max_violation, norm, dtype))
    log.append(('Le' + name, v, na.shape[0]))
    total += scale * v
    ga, gb = contrastive_loss_backward(na, nb, margin, max_violation, norm, dtype)
    d[a] = d[a] + scale * l2_normalize_backward(raw[a], ga, dtype)
    d[b] = d[b] + scale * l2_normalize_backward(raw[b], gb, dtype)

  add_loss('vid', 'para', 1.0, '_vid')
  add_loss('vctx', 'pctx', 1.0, '_ctx_low_lvel')
  add_loss('vid', 'vid', 0.5, '_vid_inloss')
  add_loss('para', 'para', 0.5, '_para_inloss')
  if low_level_loss:
    add_loss('clip', 'cap', 1.0, '_low_lvel')
    add_loss('clip', 'clip', 0.5, '_clip_inloss')
    add_loss('cap', 'cap', 0.5, '_cap_inloss')

  def euclid(a, b, name):
    v = float(euclidean_loss(a, b, norm, dtype))
    log.append(('Le' + name, v, b.shape[0]))
    return v, euclidean_loss_backward(a, b, norm, dtype)

  v1, g_clip_recon = euclid(clip_recon, clip_emb, '_clip_recon')
  v2, g_cap_recon = euclid(cap_recon, cap_emb, '_cap_recon')
  total += (v1 + v2) * weight_recon
  g_clip_recon = g_clip_recon * weight_recon
  g_cap_recon = g_cap_recon * weight_recon
  grads = [None] * (8 if lowest else 6)
  if lowest:
    frames = np.concatenate([np.asarray(clips, dtype=dtype)[i, :l]
                             for i, l in enumerate(np.asarray(lengths_clip))], 0)
    words = np.concatenate([word[i, :l] for i, l in enumerate(np.asarray(lengths_cap))], 0)
    v3, g_fr = euclid(frame_recon, frames, '_reconstruct_frame_hier')
    v4, g_wd = euclid(sent_recon, words, '_reconstruct_word_hier')
    total += (v3 + v4) * lowest_weight_recon
    grads[6], d_cr = decoder_backward(d_cdec, g_fr * lowest_weight_recon)
    grads[7], d_pr = decoder_backward(d_sdec, g_wd * lowest_weight_recon)
    g_clip_recon = g_clip_recon + d_cr
    g_cap_recon = g_cap_recon + d_pr
  grads[4], d_vid_dec = decoder_backward(d_vdec, g_clip_recon)
  grads[5], d_para_dec = decoder_backward(d_tdec, g_cap_recon)
  d['vid'] = d['vid'] + d_vid_dec
  d['para'] = d['para'] + d_para_dec

  gather = lambda dxp, counts: np.concatenate([dxp[i, :c] for i, c in enumerate(counts)], 0)
  grads[2], dx_v2, dh0_v2 = pooled_gru_backward(c_v2, d['vid'])
  grads[3], dx_p2, dh0_p2 = pooled_gru_backward(c_p2, d['para'])
  g_clip, _, _ = pooled_gru_backward(c_clip, d['clip'] + gather(dx_v2, num_clips))
  g_vid, _, _ = pooled_gru_backward(c_vid, d['vctx'] + dh0_v2)
  g_cap, dx_cap, _ = pooled_gru_backward(c_cap, d['cap'] + gather(dx_p2, num_caps))
  g_par, dx_par, _ = pooled_gru_backward(c_par, d['pctx'] + dh0_p2)
  grads[0] = {k: g_clip[k] + g_vid[k] for k in g_clip}
  grads[1] = {k: g_cap[k] + g_par[k] for k in g_cap}
  dtable = np.zeros_like(table)
  for toks, lens, dxx in [(captions, lengths_cap, dx_cap), (paragraphs, lengths_paragraph, dx_par)]:
    toks = np.asarray(toks)
    for i, l in enumerate(np.asarray(lens)):
      np.add.at(dtable, toks[i, :l], dxx[i, :l])
  grads[1]['embed.weight'] = dtable
  return log, total, grads


# --------------------------------------------------------------------------------------------
# GroupWiseContrastiveLoss, /root/reference/loss.py:15-72
# --------------------------------------------------------------------------------------------
def _block_bounds(counts):
  off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
  return off


def groupwise_reduce(scores, num_clips, num_caps, max_violation):
  """loss.py:27-38: block max (max_violation) or mean of the clip x caption score matrix.
  Returns the reduced [B,B] matrix and, for max, the flat arg-max inside each block."""
  ro, co = _block_bounds(num_clips), _block_bounds(num_caps)
  B = len(num_clips)
  red = np.zeros((B, B), dtype=scores.dtype)
  arg = np.zeros((B, B, 2), dtype=np.int64)
  for i in range(B):
    for j in range(B):
      blk = scores[ro[i]:ro[i + 1], co[j]:co[j + 1]]
      if max_violation:
        k = np.unravel_index(np.argmax(blk), blk.shape)
        red[i, j] = blk[k]
        arg[i, j] = (ro[i] + k[0], co[j] + k[1])
      else:
        red[i, j] = blk.mean()
  return red, arg


def _hinge_from_scores(scores, margin, max_violation, norm, dtype):
  n = scores.shape[0]
  diag = np.diag(scores).reshape(n, 1)
  eye = np.eye(n, dtype=bool)
  cost_s = np.where(eye, 0, np.maximum(margin + scores - diag, 0))
  cost_im = np.where(eye, 0, np.maximum(margin + scores - diag.T, 0))
  if max_violation:
    g_s = np.zeros_like(scores); g_im = np.zeros_like(scores)
    js = cost_s.argmax(axis=1)
    g_s[np.arange(n), js] = (cost_s[np.arange(n), js] > 0)
    is_ = cost_im.argmax(axis=0)
    g_im[is_, np.arange(n)] = (cost_im[is_, np.arange(n)] > 0)
    loss = cost_s.max(axis=1).sum() + cost_im.max(axis=0).sum()
  else:
    g_s = (cost_s > 0).astype(dtype); g_im = (cost_im > 0).astype(dtype)
    loss = cost_s.sum() + cost_im.sum()
  G = g_s + g_im
  G[np.arange(n), np.arange(n)] -= g_s.sum(axis=1) + g_im.sum(axis=0)
  if norm:
    loss = loss / (n * n)
    G = G / (n * n)
  return loss, G


def groupwise_contrastive_loss(im, s, num_clips, num_caps, margin=0.0, max_violation=False,
                               norm=True, dtype=np.float64, want_grad=False):
  """GroupWiseContrastiveLoss.forward (loss.py:26-71) and, optionally, (d im, d s)."""
  im = np.asarray(im, dtype=dtype); s = np.asarray(s, dtype=dtype)
  scores = im @ s.T
  red, arg = groupwise_reduce(scores, num_clips, num_caps, max_violation)
  loss, G = _hinge_from_scores(red, margin, max_violation, norm, dtype)
  if not want_grad:
    return dtype(loss)
  ro, co = _block_bounds(num_clips), _block_bounds(num_caps)
  dS = np.zeros_like(scores)
  B = len(num_clips)
  for i in range(B):
    for j in range(B):
      if max_violation:
        dS[arg[i, j, 0], arg[i, j, 1]] += G[i, j]
      else:
        dS[ro[i]:ro[i + 1], co[j]:co[j + 1]] += G[i, j] / (num_clips[i] * num_caps[j])
  return dtype(loss), dS @ s, dS.T @ im
