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


class SeqInput(object):
  """Non-tensor description of how one packed batch is addressed (see cmhse_seq_batch):
    kind 'padded'  x is [S,T,I] features (layers.*.forward);
    kind 'tokens'  token ids [S,L] int64 + the embedding table (model.EncoderText.forward);
    kind 'rows'    x is [R,I] level-1 embeddings, sequence s = `counts[s]` consecutive rows
                   (VSE.structure_emb, model.py:238-255);
    kind 'repeat'  x is [S,I]; sequence s is x[s] repeated lens[s] times (the decoders' input,
                   VSE.reconstruct_emb, model.py:257-270) — never materialised;
    kind 'multi'   several padded feature tensors (`tensors`, not differentiated) or several token
                   tensors (`tokens` is a list) packed into ONE launch sequence: clips + whole
                   videos share clip_enc, sentences + paragraphs share txt_enc (model.py:319-320),
                   so one pass over max(T) steps serves both instead of two passes."""

  def __init__(self, kind, lens, pool, tokens=None, counts=None, tensors=None, sched=None,
               step_events=None, side=True):
    self.kind, self.lens, self.pool, self.tokens, self.counts = kind, lens, pool, tokens, counts
    self.tensors = tensors
    self.need_grad = False
    # a prebuilt ops.SeqSchedule for these sequences (built ahead of the step, or by the host pull
    # that fills `tensors` chunk by chunk: then `step_events` = {step: event} of ops.pull_steps);
    # side=False: this request gets no companion stream (it is taken by that pull)
    self.sched, self.step_events, self.side = sched, step_events, side


class _Saved(object):
  """What one request's backward needs from its forward (lives on the autograd ctx)."""
  __slots__ = ('fctx', 'spec', 'keep', 'x_shape', 'table_shape', 'has_hidden', 'saved_for_bwd',
               'row_starts')


def _fwd_request(spec, x, hidden, table, w_ih, w_hh, b_ih, b_hh, w_lin, b_lin, w_att):
  """The ops.gru_pool_fwd_multi request of one encoder call + its _Saved record."""
  pool = spec.pool
  weights = dict(w_ih=w_ih.detach(), w_hh=w_hh.detach(), b_ih=b_ih.detach(), b_hh=b_hh.detach())
  if pool == ops.POOL_ATTN:
    weights.update(w_lin=w_lin.detach(), b_lin=b_lin.detach(), w_att=w_att.detach().reshape(-1))
  H = w_hh.shape[1]
  sv = _Saved()
  sv.row_starts = None
  keep, x_ptrs, tok_ptrs, emb = [], None, None, None
  if spec.kind == 'tokens' or (spec.kind == 'multi' and spec.tokens is not None):
    toks = spec.tokens if spec.kind == 'multi' else [spec.tokens]
    ptr_list = []
    for tok in toks:   # padded [S, L] ids, or ops.Ragged (a packed batch: no padding stored)
      ops._require_cuda(tok, 'tokens')
      tok = ops.seq_keep(tok, torch.int64)
      keep.append(tok)
      ptr_list.append(ops.seq_row_ptrs(tok))
    emb = table.detach().float().contiguous()
    I = emb.shape[1]
    tok_ptrs = np.concatenate(ptr_list)
    device = keep[0].device
    keep.append(emb)
  elif spec.kind == 'multi':
    ptr_list = []
    for t in spec.tensors:   # padded [S, T, I] features, or ops.Ragged
      ops._require_cuda(t, 'x')
      tc = ops.seq_keep(t, torch.float32)
      keep.append(tc)
      ptr_list.append(ops.seq_row_ptrs(tc))
    I = keep[0].shape[2]
    x_ptrs = np.concatenate(ptr_list)
    device = keep[0].device
  else:
    ops._require_cuda(x, 'x')
    xc = x.detach().float().contiguous()
    device = xc.device
    keep.append(xc)
    if spec.kind == 'padded':
      I = xc.shape[2]
      x_ptrs = ops.padded_row_ptrs(xc)
    elif spec.kind == 'repeat':
      I = xc.shape[1]
      x_ptrs = ops.padded_row_ptrs(xc)
    else:
      I = xc.shape[1]
      counts = np.asarray(spec.counts, dtype=np.int64)
      starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
      x_ptrs = np.uint64(xc.data_ptr()) + starts * np.uint64(I * 4)
      sv.row_starts = starts
  h0_ptrs = None
  if hidden is not None:
    ops._require_cuda(hidden, 'hidden')
    h0 = hidden.detach().float().contiguous()
    keep.append(h0)
    h0_ptrs = ops.padded_row_ptrs(h0)
  need_grad = spec.need_grad   # decided by the caller: grad mode is off inside Function.forward
  req = dict(weights=weights, pool_mode=pool, lens=spec.lens, I=I, H=H, device=device,
             x_ptrs=x_ptrs, tok_ptrs=tok_ptrs, emb_table=emb, h0_ptrs=h0_ptrs,
             save_for_backward=need_grad, constant_input=(spec.kind == 'repeat'))
  if spec.sched is not None:
    req.update(sched=spec.sched, step_events=spec.step_events)
  if not spec.side:
    req['side'] = False
  sv.spec, sv.keep = spec, keep
  sv.x_shape = None if x is None else tuple(x.shape)
  sv.table_shape = None if table is None else tuple(table.shape)
  sv.has_hidden = hidden is not None
  sv.saved_for_bwd = need_grad
  return req, sv


def _bwd_request(sv, grad_out, need):
  """The ops.gru_pool_bwd_multi request of one encoder call; `need` = needs_input_grad of its ten
  inputs (x, hidden, table, weights...).  Returns (request, dx, dtable)."""
  if not sv.saved_for_bwd:
    raise RuntimeError('cmhse_amd: forward ran without saving state for backward')
  spec, fctx = sv.spec, sv.fctx
  device = fctx['device']
  dx = dtable = dx_ptrs = None
  if spec.kind not in ('tokens', 'multi') and need[0]:
    dx = torch.zeros(sv.x_shape, dtype=torch.float32, device=device)
    if spec.kind in ('padded', 'repeat'):
      dx_ptrs = ops.padded_row_ptrs(dx)      # 'repeat': the kernel accumulates over the steps
    else:
      dx_ptrs = np.uint64(dx.data_ptr()) + sv.row_starts * np.uint64(sv.x_shape[1] * 4)
  if sv.table_shape is not None and need[2]:
    dtable = torch.zeros(sv.table_shape, dtype=torch.float32, device=device)
  req = dict(fctx=fctx, dout=grad_out, dx_ptrs=dx_ptrs, d_emb_table=dtable,
             want_dh0=sv.has_hidden and need[1])
  return req, dx, dtable


def _grads_tuple(grads, dx, dh0, dtable):
  w_att_g = grads.get('w_att')
  return (dx, dh0, dtable, grads['w_ih'], grads['w_hh'], grads['b_ih'], grads['b_hh'],
          grads.get('w_lin'), grads.get('b_lin'), None if w_att_g is None else w_att_g.reshape(1, -1))


class _PackedGRUPoolFn(torch.autograd.Function):
  """Forward = cmhse_gru_pool_fwd, backward = cmhse_gru_pool_bwd (BPTT on the HIP path)."""

  @staticmethod
  def forward(ctx, spec, x, hidden, table, w_ih, w_hh, b_ih, b_hh, w_lin, b_lin, w_att):
    req, sv = _fwd_request(spec, x, hidden, table, w_ih, w_hh, b_ih, b_hh, w_lin, b_lin, w_att)
    out, sv.fctx = ops.gru_pool_fwd(**req)
    ctx.sv = sv
    return out

  @staticmethod
  def backward(ctx, grad_out):
    req, dx, dtable = _bwd_request(ctx.sv, grad_out, ctx.needs_input_grad[1:])
    (grads, dh0), = ops.gru_pool_bwd_multi([req])
    return (None,) + _grads_tuple(grads, dx, dh0, dtable)


class _GroupedGRUPoolFn(torch.autograd.Function):
  """Several INDEPENDENT encoder calls as one autograd node: forward = cmhse_gru_pool_fwd_multi,
  backward = cmhse_gru_pool_bwd_multi — step t of every encoder shares one launch in both
  directions (the two towers of VSE.train_emb).  Inputs: `specs` (list of SeqInput), then the ten
  tensors of _PackedGRUPoolFn per spec, flattened; outputs: one tensor per spec."""

  @staticmethod
  def forward(ctx, specs, streams, *flat):
    reqs, svs = [], []
    for i, spec in enumerate(specs):
      req, sv = _fwd_request(spec, *flat[10 * i:10 * i + 10])
      reqs.append(req)
      svs.append(sv)
    results = ops.gru_pool_fwd_multi(reqs, job_streams=streams)
    for sv, (_, fctx) in zip(svs, results):
      sv.fctx = fctx
    ctx.svs, ctx.streams = svs, streams
    return tuple(out for out, _ in results)

  @staticmethod
  def backward(ctx, *grad_outs):
    need = ctx.needs_input_grad[1:]      # (after `streams`)
    reqs, extra = [], []
    for i, (sv, g) in enumerate(zip(ctx.svs, grad_outs)):
      if g is None:      # this encoder's output did not reach the loss
        fctx = sv.fctx
        n_out = fctx['sched'].sum_T if fctx['pool_mode'] == ops.POOL_ALL else fctx['sched'].S
        g = torch.zeros(n_out, fctx['H'], dtype=torch.float32, device=fctx['device'])
      req, dx, dtable = _bwd_request(sv, g, need[1 + 10 * i:1 + 10 * i + 10])
      reqs.append(req)
      extra.append((dx, dtable))
    out = [None, None]
    for (grads, dh0), (dx, dtable) in zip(ops.gru_pool_bwd_multi(reqs, job_streams=ctx.streams), extra):
      out.extend(_grads_tuple(grads, dx, dh0, dtable))
    return tuple(out)


def run_grouped(calls, streams=None):
  """`calls`: list of (layer, SeqInput, x, hidden, table) for independent encoders; runs them as
  ONE autograd node and returns their outputs in order.  Without `streams` their time steps share
  launches on the current stream; with `streams` (one torch stream per call) every encoder is a
  chain of its own on its stream, forward and backward, and the host queues the launches of all
  of them step by step — side by side from the first step."""
  flat, specs = [], []
  for layer, spec, x, hidden, table in calls:
    w_lin, b_lin, w_att = layer._extra_weights()
    tensors = (x, hidden, table, layer.rnn.weight_ih_l0, layer.rnn.weight_hh_l0,
               layer.rnn.bias_ih_l0, layer.rnn.bias_hh_l0, w_lin, b_lin, w_att)
    spec.need_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                     for t in tensors)
    specs.append(spec)
    flat.extend(tensors)
  if len({s.need_grad for s in specs}) > 1:    # mixed: keep them apart (not a training-step case)
    return [_PackedGRUPoolFn.apply(spec, *flat[10 * i:10 * i + 10]) for i, spec in enumerate(specs)]
  return list(_GroupedGRUPoolFn.apply(specs, None if streams is None else list(streams), *flat))


# tools/train_profile.py --timeline sets this to a function(name, stream): a mark on a tower's stream
MARK = None


def _mark(name, streams):
  if MARK is not None:
    for i, st in enumerate(streams):
      MARK('%s[%d]' % (name, i), st)


class _TowersFn(torch.autograd.Function):
  """The two levels of several INDEPENDENT towers as ONE autograd node (VSE.train_emb: clip_enc ->
  vid_seq_enc beside txt_enc -> txt_seq_enc, model.py:319-331).  Every tower lives on its own
  stream from its first level-1 step to its last level-2 step, and back: level 2 of a tower starts
  right behind its level 1 without waiting for the other tower (CMHSE_NO_JOIN), and in the backward
  pass the gradient of the level-1 outputs — what the loss sends plus what level 2 sends — is summed
  on the tower's stream, so nothing between the two levels runs on, or waits for, the caller's
  stream.  As two grouped nodes the towers met on the caller's stream twice per direction (0.3 ms
  forward, 0.45 ms backward of a 10 ms step: event hops and a dozen small autograd kernels).
  Inputs: `meta` = per tower (level-1 SeqInput, rows of the first level-1 block, level-2 SeqInput),
  `streams`, then twenty tensors per tower (the ten of _PackedGRUPoolFn for level 1, then for
  level 2 with x / hidden / table None).  Outputs per tower: level-1 rows [:n], level-1 rows [n:],
  level-2 output."""

  @staticmethod
  def forward(ctx, meta, streams, *flat):
    hold = []
    reqs1, svs1, reqs2, svs2 = [], [], [], []
    for i, (spec1, n, spec2) in enumerate(meta):
      req, sv = _fwd_request(spec1, *flat[20 * i:20 * i + 10])
      # level 1 writes into a buffer known now, so that level 2's tables (addresses of its rows
      # and initial states) are built and uploaded BEFORE the first chain launch: a copy queued on
      # the caller's stream behind running chains was seen to wait for them
      out1 = torch.empty(len(spec1.lens), req['H'], dtype=torch.float32, device=req['device'])
      req['out'] = out1
      reqs1.append(req)
      svs1.append(sv)
      req, sv = _fwd_request(spec2, out1[:n], out1[n:], None, *flat[20 * i + 13:20 * i + 20])
      req['sched'] = ops.SeqSchedule(req['lens'], req['device'], req['x_ptrs'], None,
                                     req['h0_ptrs'])
      reqs2.append(req)
      svs2.append(sv)
    if MARK is not None:
      MARK('fwd:queued', None)     # on the caller's stream: everything in front of the first chain launch
    res1 = ops.gru_pool_fwd_multi(reqs1, job_streams=streams, join=False, hold=hold)
    _mark('fwd:level1', streams)
    for i in range(len(meta)):
      svs1[i].fctx = res1[i][1]
    res2 = ops.gru_pool_fwd_multi(reqs2, job_streams=streams)   # joins everything queued above
    _mark('fwd:level2', streams)
    outs = []
    for i, (_, n, _) in enumerate(meta):
      svs2[i].fctx = res2[i][1]
      outs.extend([res1[i][0][:n], res1[i][0][n:], res2[i][0]])
    ctx.svs1, ctx.svs2, ctx.streams, ctx.meta = svs1, svs2, streams, meta
    del hold
    return tuple(outs)

  @staticmethod
  def backward(ctx, *grads):
    need = ctx.needs_input_grad[2:]
    streams, hold = ctx.streams, []
    inner = (True, True) + (False,) * 8       # level 2: wrt its rows and its initial state
    # Everything that is queued on the caller's stream — zero fills, table uploads — for BOTH levels
    # first: the towers' streams fork from it once, and nothing is left to wait for in between.
    reqs2, dxs, reqs1, extra, douts = [], [], [], [], []
    for i, (sv1, sv2) in enumerate(zip(ctx.svs1, ctx.svs2)):
      g = grads[3 * i + 2]
      if g is None:
        g = torch.zeros(sv2.fctx['sched'].S, sv2.fctx['H'], dtype=torch.float32,
                        device=sv2.fctx['device'])
      req, dx, _ = _bwd_request(sv2, g, inner)
      reqs2.append(req)
      dxs.append(dx)
      dout = torch.empty(sv1.fctx['sched'].S, sv1.fctx['H'], dtype=torch.float32,
                         device=sv1.fctx['device'])
      req, dx, dtable = _bwd_request(sv1, dout, need[20 * i:20 * i + 10])
      reqs1.append(req)
      extra.append((dx, dtable))
      douts.append(dout)
    prep2, prep1 = ops.prepare_bwd(reqs2), ops.prepare_bwd(reqs1)
    _mark('bwd:start', streams)
    res2 = ops.gru_pool_bwd_multi(reqs2, job_streams=streams, join=False, hold=hold, prepared=prep2)
    _mark('bwd:level2', streams)
    for i, dout in enumerate(douts):
      n = ctx.meta[i][1]
      g_first, g_rest = grads[3 * i], grads[3 * i + 1]
      dx2, dh0 = dxs[i], res2[i][1]
      with torch.cuda.stream(streams[i]):   # the gradient of level 1's output, on the tower's stream
        if g_first is None:
          dout[:n].copy_(dx2)
        else:
          torch.add(g_first, dx2, out=dout[:n])
        if g_rest is None:
          dout[n:].copy_(dh0)
        else:
          torch.add(g_rest, dh0, out=dout[n:])
    res1 = ops.gru_pool_bwd_multi(reqs1, job_streams=streams, prepared=prep1)   # joins everything
    _mark('bwd:level1', streams)
    out = [None, None]
    for i in range(len(ctx.svs1)):
      out.extend(_grads_tuple(res1[i][0], extra[i][0], res1[i][1], extra[i][1]))
      out.extend(_grads_tuple(res2[i][0], None, None, None))
    del hold
    return tuple(out)


def run_towers(towers, streams):
  """`towers`: list of (level-1 call, n, level-2 layer, counts): the level-1 call is a
  (layer, SeqInput, x, hidden, table) description whose output rows [:n] are the rows of level 2
  (sequence s = counts[s] consecutive rows) and rows [n:] its initial hidden states (one per
  sequence).  Runs both levels of every tower as one autograd node on `streams` (one torch stream
  per tower); returns per tower (rows [:n], rows [n:], level-2 output) — or None when the layers do
  not agree on requires_grad (a frozen encoder), which one node cannot express."""
  flat, meta, grad_modes = [], [], set()
  for (layer, spec, x, hidden, table), n, layer2, counts in towers:
    counts = np.asarray(counts, dtype=np.int64)
    if int(counts.sum()) != n:
      raise ValueError('run_towers: level-2 counts must cover the first %d level-1 rows' % n)
    spec2 = SeqInput('rows', counts, layer2.POOL, counts=counts)
    t1 = (x, hidden, table, layer.rnn.weight_ih_l0, layer.rnn.weight_hh_l0, layer.rnn.bias_ih_l0,
          layer.rnn.bias_hh_l0) + tuple(layer._extra_weights())
    t2 = (None, None, None, layer2.rnn.weight_ih_l0, layer2.rnn.weight_hh_l0,
          layer2.rnn.bias_ih_l0, layer2.rnn.bias_hh_l0) + tuple(layer2._extra_weights())
    g1 = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in t1)
    g2 = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in t2)
    spec.need_grad, spec2.need_grad = g1, g1 or g2
    grad_modes.update([g1, g1 or g2])
    meta.append((spec, n, spec2))
    flat.extend(t1 + t2)
  if len(grad_modes) > 1:
    return None      # e.g. one frozen encoder: the caller runs the levels as separate nodes
  outs = _TowersFn.apply(meta, list(streams), *flat)
  return [tuple(outs[3 * i:3 * i + 3]) for i in range(len(towers))]


def _lens_numpy(q_len):
  if isinstance(q_len, torch.Tensor):
    return q_len.detach().cpu().numpy().astype(np.int64)   # lengths live on the host (layers.py:97)
  return np.asarray(q_len, dtype=np.int64)


class _GRUPoolBase(nn.Module):
  POOL = None
  # What `rnn_bidirectional=True` means upstream differs per layer (every reference call site passes
  # False, model.py:107-114): Seq2Seq builds a bidirectional nn.GRU and concatenates the two final
  # states (layers.py:31-34,58-59); Attention builds one too, but its `lin` stays H -> H, so its
  # forward fails on the 2H-wide states (layers.py:75-80,105); Maxout stores the flag and builds a
  # unidirectional GRU regardless (layers.py:167-172).  The same here, state-dict keys included
  # (`rnn.weight_ih_l0_reverse`, ...).
  BIDIRECTIONAL_GRU = True

  def __init__(self, embedding_features, rnn_features, rnn_bidirectional=False):
    super(_GRUPoolBase, self).__init__()
    self.bidirectional = rnn_bidirectional
    self.features = rnn_features
    self.rnn = nn.GRU(input_size=embedding_features, hidden_size=rnn_features, num_layers=1,
                      batch_first=True, bidirectional=bool(rnn_bidirectional) and self.BIDIRECTIONAL_GRU)
    self._build_extra(rnn_features)
    self._init_rnn(self.rnn.weight_ih_l0)
    self._init_rnn(self.rnn.weight_hh_l0)
    self.rnn.bias_ih_l0.data.zero_()
    self.rnn.bias_hh_l0.data.zero_()

  def _two_directions(self):
    return self.rnn.bidirectional

  def _no_bidirectional(self, what):
    if self._two_directions():
      raise RuntimeError('cmhse_amd: %s of a bidirectional %s is not supported (upstream only '
                         'Seq2Seq.forward(q_emb, q_len) works with rnn_bidirectional=True)'
                         % (what, type(self).__name__))

  def _build_extra(self, rnn_features):
    pass

  def _init_rnn(self, weight):
    # xavier-uniform per gate chunk, layers.py:41-43
    for w in weight.chunk(3, 0):
      init.xavier_uniform_(w)

  def _extra_weights(self):
    return None, None, None

  def _run(self, spec, x, hidden, table, reverse=False):
    self._no_bidirectional('this entry point') if (self._two_directions() and spec.kind != 'padded') else None
    w_lin, b_lin, w_att = self._extra_weights()
    sfx = '_reverse' if reverse else ''
    gru = [getattr(self.rnn, n + sfx) for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]
    spec.need_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in [x, hidden, table] + gru + [w_lin, b_lin, w_att])
    return _PackedGRUPoolFn.apply(spec, x, hidden, table, gru[0], gru[1], gru[2], gru[3], w_lin, b_lin, w_att)

  def forward(self, q_emb, q_len, hidden=None):
    if self._two_directions():
      return self._forward_bidirectional(q_emb, q_len, hidden)
    if isinstance(q_emb, ops.Ragged):   # a packed (un-padded) batch: plain data, no gradient wrt it
      if hidden is not None:
        raise ValueError('a Ragged input takes no initial hidden state (level-1 encoders only)')
      return self.forward_multi([q_emb], [q_len])
    return self._run(SeqInput('padded', _lens_numpy(q_len), self.POOL), q_emb, hidden, None)

  def _forward_bidirectional(self, q_emb, q_len, hidden):
    raise RuntimeError('cmhse_amd: %s.forward with rnn_bidirectional=True fails upstream as well (its '
                       'pooling is built for H-wide states, layers.py:75-80,105)' % type(self).__name__)

  def forward_rows(self, rows, counts, hidden=None):
    """Level-2 form (VSE.structure_emb): sequence s is `counts[s]` consecutive rows of `rows`;
    differentiable wrt `rows` and `hidden`."""
    self._no_bidirectional('forward_rows')
    counts = np.asarray(counts, dtype=np.int64)
    return self._run(SeqInput('rows', counts, self.POOL, counts=counts), rows, hidden, None)

  def _weights(self):
    self._no_bidirectional('the fused inference entry points')
    w_lin, b_lin, w_att = self._extra_weights()
    weights = dict(w_ih=self.rnn.weight_ih_l0.detach(), w_hh=self.rnn.weight_hh_l0.detach(),
                   b_ih=self.rnn.bias_ih_l0.detach(), b_hh=self.rnn.bias_hh_l0.detach())
    if self.POOL == ops.POOL_ATTN:
      weights.update(w_lin=w_lin.detach(), b_lin=b_lin.detach(), w_att=w_att.detach().reshape(-1))
    return weights

  # -- call descriptions for run_grouped(): (layer, SeqInput, x, hidden, table) -------------------
  def call_multi(self, tensors, lens_list, sched=None, step_events=None):
    self._no_bidirectional('a grouped call')
    lens = np.concatenate([_lens_numpy(l) for l in lens_list])
    return (self, SeqInput('multi', lens, self.POOL, tensors=list(tensors), sched=sched,
                           step_events=step_events), None, None, None)

  def call_tokens_multi(self, token_tensors, lens_list, table, sched=None, side=True):
    self._no_bidirectional('a grouped call')
    lens = np.concatenate([_lens_numpy(l) for l in lens_list])
    return (self, SeqInput('multi', lens, self.POOL, tokens=list(token_tensors), sched=sched,
                           side=side), None, None, table)

  def call_rows(self, rows, counts, hidden=None):
    self._no_bidirectional('a grouped call')
    counts = np.asarray(counts, dtype=np.int64)
    return (self, SeqInput('rows', counts, self.POOL, counts=counts), rows, hidden, None)

  def call_repeat(self, rows, counts):
    self._no_bidirectional('a grouped call')
    counts = np.asarray(counts, dtype=np.int64)
    return (self, SeqInput('repeat', counts, self.POOL), rows, None, None)

  def forward_multi(self, tensors, lens_list):
    """Several padded feature batches [S_i, T_i, I] through this encoder in one packed pass;
    returns the [sum S_i, H] outputs in order (differentiable wrt the weights)."""
    lens = np.concatenate([_lens_numpy(l) for l in lens_list])
    return self._run(SeqInput('multi', lens, self.POOL, tensors=list(tensors)), None, None, None)

  def forward_tokens_multi(self, token_tensors, lens_list, table):
    """Several token batches through this encoder in one packed pass (differentiable wrt the
    weights and the embedding table)."""
    lens = np.concatenate([_lens_numpy(l) for l in lens_list])
    return self._run(SeqInput('multi', lens, self.POOL, tokens=list(token_tensors)), None, None,
                     table)

  def forward_ptrs(self, lens, in_dim, device, x_ptrs=None, tok_ptrs=None, table=None,
                   h0_ptrs=None, out=None, step_plan=None):
    """Inference-only entry used by the fused paths (EncoderText, structure_emb, encode_data):
    sequences are given by base address (numpy uint64, input order), so padded batches, several
    loader batches at once and consecutive-row level-2 inputs are consumed in place."""
    out, _ = ops.gru_pool_fwd(**self.request_ptrs(lens, in_dim, device, x_ptrs, tok_ptrs, table,
                                                  h0_ptrs, out, step_plan=step_plan))
    return out

  def request_ptrs(self, lens, in_dim, device, x_ptrs=None, tok_ptrs=None, table=None,
                   h0_ptrs=None, out=None, sched=None, step_events=None, step_plan=None):
    """The arguments of forward_ptrs as one request of ops.gru_pool_fwd_multi, which runs
    independent encoders (the visual and the text tower) in shared per-step launches."""
    return dict(weights=self._weights(), pool_mode=self.POOL, lens=lens, I=in_dim,
                H=self.rnn.weight_hh_l0.shape[1], device=device, x_ptrs=x_ptrs,
                tok_ptrs=tok_ptrs, emb_table=table, h0_ptrs=h0_ptrs, out=out, sched=sched,
                step_events=step_events, step_plan=step_plan)

  def forward_tokens(self, tokens, q_len, table):
    """Fused embedding-lookup + encoder (model.EncoderText.forward, model.py:92-99): the word
    vectors are gathered inside the GRU operand load and never materialised."""
    return self._run(SeqInput('tokens', _lens_numpy(q_len), self.POOL, tokens=tokens), None,
                       None, table)


def _reverse_valid_steps(q_emb, lens):
  """q_emb [S, T, I] with step t of sequence s moved to step len_s - 1 - t (padding stays where it
  is): what the reverse direction of a bidirectional nn.GRU over a packed batch consumes.  An index
  gather — data movement, differentiable."""
  S, T = q_emb.shape[0], q_emb.shape[1]
  t = torch.arange(T)[None, :]
  ln = torch.as_tensor(lens, dtype=torch.int64)[:, None]
  idx = torch.where(t < ln, ln - 1 - t, t).to(q_emb.device)
  return torch.gather(q_emb, 1, idx[:, :, None].expand(S, T, q_emb.shape[2]))


class Seq2Seq(_GRUPoolBase):
  """/root/reference/layers.py:26-66 — final hidden state; with rnn_bidirectional=True the final
  states of the two directions side by side ([S, 2H], layers.py:58-59): the reverse direction is
  the same packed GRU kernel over the time-reversed valid steps with the `_reverse` weights."""
  POOL = ops.POOL_LAST

  def _forward_bidirectional(self, q_emb, q_len, hidden):
    if hidden is not None:      # upstream hands nn.GRU a (1, N, H) state where it expects (2, N, H)
      raise RuntimeError('Expected hidden size (2, %d, %d), got [1, %d, %d]'
                         % (hidden.shape[0], hidden.shape[1], hidden.shape[0], hidden.shape[1]))
    if isinstance(q_emb, ops.Ragged):
      raise RuntimeError('cmhse_amd: a bidirectional Seq2Seq takes padded [S, T, I] input')
    lens = _lens_numpy(q_len)
    fwd = self._run(SeqInput('padded', lens, self.POOL), q_emb, None, None)
    bwd = self._run(SeqInput('padded', lens, self.POOL), _reverse_valid_steps(q_emb, lens), None, None,
                    reverse=True)
    return torch.cat([fwd, bwd], dim=1)


class Maxout(_GRUPoolBase):
  """/root/reference/layers.py:164-204 — per-sequence max over valid steps (always one direction:
  upstream builds its GRU with bidirectional=False whatever the flag says, :167-172)."""
  POOL = ops.POOL_MAX
  BIDIRECTIONAL_GRU = False


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
