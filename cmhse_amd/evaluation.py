"""Host-side mirror of the reference's `evaluation.py` on the MI355X hot path.

Exports the names train.py imports (train.py:10): i2t, t2i, AverageMeter, LogCollector,
encode_data, LogReporter — with the reference's signatures and return values
(/root/reference/evaluation.py:18-213).

MI355X-first differences in HOW (not WHAT) things are computed:
  * encode_data fuses many loader batches into one "super-batch" per encoder launch sequence
    (clips and whole-video streams of all those batches go through `clip_enc` together, sentences
    and paragraphs through `txt_enc` together), keeps every embedding on the device, and still
    reports the per-loader-batch 'Letest' loss (evaluation.py:129) and returns the same 8-tuple;
  * i2t / t2i never build the N x N matrix or sort: ranks and arg-max come out of the fused
    similarity kernel (cmhse_sim_rank).
"""
from __future__ import annotations

import contextlib
import gc
import hashlib
import os
import time
from collections import OrderedDict

import numpy
import numpy as np
import torch

from . import ops


class AverageMeter(object):
  """Running value + weighted mean of one logged quantity — the contract of
  /root/reference/evaluation.py:18-45: `update(val, n=0)` (a weight of 0 only refreshes `val`),
  `avg = sum / (1e-4 + count)` (so an empty meter reads 0, not NaN), and the printed form."""
  __slots__ = ('val', 'avg', 'sum', 'count')

  def __init__(self):
    self.reset()

  def reset(self):
    self.val = self.avg = self.sum = self.count = 0

  def update(self, val, n=0):
    self.val = val
    self.sum += n * val
    self.count += n
    self.avg = self.sum / (self.count + 1e-4)

  def __str__(self):
    return str(self.val) if self.count == 0 else '%.4f (%.4f)' % (self.val, self.avg)


class LogCollector(object):
  """Named AverageMeters in first-use order (/root/reference/evaluation.py:48-72): what
  VSE.forward_loss writes to (`model.logger.update('Le'+name, value, n)`, model.py:291) and what
  train.py prints / sends to tensorboard.

  Values may arrive late: VSE.train_emb hands the step's loss values over as a copy that is still
  in flight (`defer`), so that the host can queue the next step instead of waiting for this one.
  Everything that READS the collector (`meters`, str()) first settles what is outstanding, and
  `update` / `tb_log` calls made in the meantime wait in the same queue, so every reader — and the
  tensorboard sink — sees exactly the reference's sequence of updates."""

  def __init__(self):
    self._meters = OrderedDict()
    self._deferred = []

  def _update(self, k, v, n=0):
    self._meters.setdefault(k, AverageMeter()).update(v, n)

  def update(self, k, v, n=0):
    if self._deferred:
      self._deferred.append(lambda: self._update(k, v, n))
    else:
      self._update(k, v, n)

  def defer(self, thunk):
    """Queue `thunk()` (which calls `_update`) to run before the next read."""
    self._deferred.append(thunk)

  def settle(self):
    while self._deferred:
      self._deferred.pop(0)()

  @property
  def meters(self):
    self.settle()
    return self._meters

  @meters.setter
  def meters(self, value):      # `collector.meters = OrderedDict()`: a reset
    # what is still queued belongs to the meters being replaced: deliver it first (the reference's
    # tb_log would already have emitted those rows), then reset
    self.settle()
    self._meters = value

  def flush(self):
    """Deliver everything still queued (late loss values, tb_log rows): call after the LAST step of
    a loop that no reader follows (a reader — `meters`, str(), train_start / val_start — does it
    implicitly)."""
    self.settle()

  def __str__(self):
    return '  '.join('%s %s' % (k, m) for k, m in self.meters.items())

  # A collector that still holds late values (closures over device events) cannot be pickled or
  # deep-copied as it is: both first settle what is outstanding, so the copy carries plain meters.
  def __getstate__(self):
    self.settle()
    return {'_meters': self._meters, '_deferred': []}

  def __deepcopy__(self, memo):
    import copy
    self.settle()
    other = LogCollector()
    other._meters = copy.deepcopy(self._meters, memo)
    return other

  def tb_log(self, tb_logger, prefix='', step=None):
    """evaluation.py:68-72.  train.py calls this after EVERY step (train.py:215).  While the step's
    loss values are still in flight the call takes its place in the queue instead of waiting for
    them: the sink receives exactly the (key, value, step) triples it would have, a step later in
    wall time, and the host goes on queueing the next step."""
    if self._deferred:
      self._deferred.append(lambda: self._tb_log_now(tb_logger, prefix, step))
    else:
      self._tb_log_now(tb_logger, prefix, step)

  def _tb_log_now(self, tb_logger, prefix, step):
    for k, m in self._meters.items():
      tb_logger.log_value(prefix + k, m.val, step=step)


def LogReporter(tb_logger, result, epoch, name):
  """/root/reference/evaluation.py:74-78: one tensorboard scalar per report entry."""
  for key, value in result.items():
    tb_logger.log_value(name + key, value, step=epoch)


# ---------------------------------------------------------------------------------------------
# encode_data
# ---------------------------------------------------------------------------------------------
def _to_dev(t, device):
  return t if t.is_cuda else t.to(device, non_blocking=True)


# Schedule of encode_group (module attributes; all give bit-identical embeddings, tested):
#   GROUP_TOWERS  step t of both level-1 encoders in shared launches (default);
#   EARLY_POOL    the visual attention pass beside the text chain's few-sequence tail (default);
#   TWO_STREAMS   instead: the two towers on two HIP streams (+3 % wall, measured; off).
TWO_STREAMS = [False]
GROUP_TOWERS = [True]
EARLY_POOL = [True]


def _tail_stream(device):
  return ops.stream_set(device)[0]


def _side_streams(device):
  st = ops.stream_set(device)
  return st[0], st[1]


PIPELINE_UPLOAD = [True]      # pinned host batches are pulled chunk by chunk under the step pipeline
UPLOAD_CHUNK = [8]             # time steps per pull chunk once the pipeline is full (4 and 8 level, 12+ slower: DESIGN §7b)


def _copy_stream(device):
  # a stream on a hardware queue of its own (ops.stream_set): a copy stream that shares the compute
  # stream's hardware queue runs in submission order with it — no overlap at all, measured
  return ops.stream_set(device)[1]


# Pinned float32 blocks for the deferred 'Letest' values, recycled by flush(): page-locking a fresh
# block while the GPU is busy costs milliseconds (see ops.upload).
_HOST_SCALARS = []
_HOST_EVENTS = []


def _host_scalars(n):
  """A pinned float32 BLOCK of at least n elements (callers slice it for use and hand the whole
  block back to the pool, so its capacity never shrinks)."""
  for i, h in enumerate(_HOST_SCALARS):
    if h.numel() >= n:
      return _HOST_SCALARS.pop(i)
  return torch.empty(max(n, 1024), dtype=torch.float32).pin_memory()


_HOST_ROWS, _HOST_ROW_EVENTS = [], []


def _hold_host_rows(stream, tensors, keep=3):
  while _HOST_ROWS and (_HOST_ROWS[0][0].query() or len(_HOST_ROWS) >= keep):
    if not _HOST_ROWS[0][0].query():
      _HOST_ROWS[0][0].synchronize()
    _HOST_ROW_EVENTS.append(_HOST_ROWS.pop(0)[0])
  ev = _HOST_ROW_EVENTS.pop() if _HOST_ROW_EVENTS else torch.cuda.Event()
  ev.record(stream)
  _HOST_ROWS.append((ev, tensors))


def _pinned_f32(t):
  return (isinstance(t, (torch.Tensor, ops.Ragged)) and not t.is_cuda and t.dtype == torch.float32
          and t.is_contiguous() and t.is_pinned())


def _empty_like_on(t, device):
  """Uninitialised device storage of the same layout as loader tensor `t` (padded or Ragged)."""
  if isinstance(t, ops.Ragged):
    return ops.Ragged(torch.empty(t.data.shape, dtype=torch.float32, device=device), t.lens)
  return torch.empty(t.shape, dtype=torch.float32, device=device)


def _lens_i64(x):
  """A batch's length member (host int64 tensor from the loader, list, or array) as int64 NumPy;
  the tensor case without the detour through `__array__` (616 of these per pass of the full split)."""
  if type(x) is torch.Tensor and x.dtype == torch.int64 and not x.is_cuda:
    return x.numpy()
  return np.asarray(x, dtype=np.int64)


def _dev_seq(t, device, dtype):
  """Loader tensor (padded or Ragged; host or device) -> contiguous `dtype` storage on `device`."""
  if type(t) is torch.Tensor and t.is_cuda and t.dtype == dtype and t.is_contiguous():
    return t          # already in place (616 of these per pass of the full split: keep it cheap)
  if isinstance(t, ops.Ragged):
    return ops.seq_keep(t if t.is_cuda else t.to(device, non_blocking=True), dtype)
  return ops.seq_keep(_to_dev(t, device), dtype)


def _plan_key(group):
  """What a cached plan of `group` is valid for: EVERY batch's four length arrays and clip /
  sentence counts (hashed in full) and the base addresses of its four feature / token tensors.  An
  in-place edit of any batch's lengths, a swapped or re-uploaded batch anywhere in the list, or a
  different number of batches gives a different key, and encode_group rebuilds the plan
  (tests/test_gpu_pipeline.py::test_plan_key_*).  What the key cannot see is an in-place edit of the
  feature VALUES — those are read afresh on every pass anyway (a plan holds addresses and
  schedules, not data).  A plan keeps its group's device tensors alive (`keep`): the whole resident
  split, 14.7 GB for ActivityNet-ICEP val; drop the plan dict to release them."""
  h = hashlib.blake2b(digest_size=16)
  ptrs = np.empty(4 * len(group), dtype=np.uint64)
  for i, b in enumerate(group):
    for k in (4, 5, 6, 7, 8, 9):
      v = b[k]
      a = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
      h.update(np.ascontiguousarray(a, dtype=np.int64).data)
      h.update(b'|')
    for k in range(4):
      ptrs[4 * i + k] = _base_ptr(b[k])
  h.update(ptrs.data)
  return (len(group), h.hexdigest())


def _base_ptr(t):
  return t.data.data_ptr() if isinstance(t, ops.Ragged) else t.data_ptr()


TOWERS = ('v1', 't1', 'v2', 't2')   # clips + whole videos | sentences + paragraphs | clips per video | sentences per paragraph


def length_histograms(batches):
  """{tower: np.bincount of the sequence lengths} of the four encoders over `batches` (12-tuples
  with their length members; a stub batch contributes nothing).  Histograms add over any partition
  of a split, which is how ranks that hold different batches agree on split_step_plan."""
  acc = {k: [] for k in TOWERS}
  for b in batches:
    if b[4] is None:
      continue
    acc['v1'] += [_lens_i64(b[4]), _lens_i64(b[6])]
    acc['t1'] += [_lens_i64(b[5]), _lens_i64(b[7])]
    acc['v2'].append(np.asarray(b[8], dtype=np.int64))
    acc['t2'].append(np.asarray(b[9] if b[9] is not None else b[8], dtype=np.int64))
  return {k: (np.bincount(np.concatenate(v)) if v else np.zeros(1, dtype=np.int64)).astype(np.int64)
          for k, v in acc.items()}


def plan_from_histograms(hists):
  """{tower: #{s : len_s > t} for t = 0 .. Tmax - 1} from length histograms."""
  return {k: (int(h.sum()) - np.cumsum(h)[:-1]).astype(np.int64) for k, h in hists.items()}


def split_step_plan(batches):
  """The step plan of a validation split: for each of the four encoders the number of sequences of
  the WHOLE split still active at time step t.  Handed to every encoder call that encodes a share
  of that split (one super-batch of several, one rank's batches: cmhse_seq_batch.step_plan_host), it
  pins which kernel serves each time step, so every sequence is encoded bit for bit as in a
  single call over the whole split — and the integer ranks do not depend on how the split was cut
  ("ranks must be identical for G in {1,2,4,8}")."""
  return plan_from_histograms(length_histograms(batches))


def encode_group(model, group, contextual_model=True, device=None, plan=None, step_plan=None, stage=None):
  """Encode a list of loader batches (12-tuples) as ONE super-batch.  Returns a dict of device
  tensors: the six un-normalised embedding matrices plus their L2-normalised versions, rows in
  loader order.  Arithmetic per sequence is identical to per-batch encoding (sequences are
  independent), only the launch granularity differs.
  `plan`: a dict the caller keeps between passes over the SAME resident batches (a validation set
  held in HBM across epochs; bench.py): the level-1 schedules (sort, step counts, the uploaded
  pointer tables) are built on the first pass and reused afterwards — 2 ms of host work in front
  of the first launch, 5 % of a rank's 45 ms share of the split.
  `stage`: a HostStage (encode_data): the level-1 matrices of each tower are normalised on ITS copy
  stream and start for the host as soon as that tower's level-1 output is final — the visual
  tower's while the text tower's few-sequence tail still steps, the text tower's under level 2 —
  instead of after the whole pass (the values do not depend on it)."""
  device = device or torch.device('cuda', torch.cuda.current_device())
  sp = step_plan or {}
  if plan is not None and plan.get('key') == _plan_key(group) and GROUP_TOWERS[0] and not TWO_STREAMS[0]:
    return _encode_group_planned(model, group, contextual_model, device, plan, sp, stage)
  clips_l, caps_l, vids_l, pars_l = [], [], [], []
  len_clip, len_cap, len_vid, len_par = [], [], [], []
  num_clips, num_caps = [], []
  # The loader hands over pinned HOST tensors (activity_net/data.py:157-162, pin_memory=True).  In
  # the grouped schedule their features are not copied up front: device buffers are only allocated
  # here, and the rows travel time-chunk by time-chunk on a copy stream while earlier GRU steps
  # already compute (ops.pull_steps; only valid steps cross PCIe, padding stays uninitialised and
  # is never read).
  pull = (PIPELINE_UPLOAD[0] and GROUP_TOWERS[0] and not TWO_STREAMS[0] and
          all(_pinned_f32(b[0]) and _pinned_f32(b[2]) for b in group))
  v_sched = v_events = None
  if pull:
    # features first: the copy stream starts pulling step 0 before anything else of this group is
    # queued (schedule metadata travels on the copy stream too; step 0's event orders it)
    for b in group:
      clips_l.append(_empty_like_on(b[0], device))
      vids_l.append(_empty_like_on(b[2], device))
    main, copy = torch.cuda.current_stream(device), _copy_stream(device)
    copy.wait_stream(main)         # the fresh device buffers may recycle blocks still in use
    with torch.cuda.stream(copy):
      v_sched = ops.SeqSchedule(
          np.concatenate([_lens_i64(b[4]) for b in group] + [_lens_i64(b[6]) for b in group]), device,
          x_ptrs=ops.seq_row_ptrs_many(clips_l + vids_l),
          src_ptrs=ops.seq_row_ptrs_many([b[0] for b in group] + [b[2] for b in group]))
      v_events = ops.pull_steps(v_sched, int(group[0][0].shape[2]), copy, UPLOAD_CHUNK[0])
      # the pull kernels read the pinned tensors by address, possibly after this function has
      # returned: they stay referenced here until an event behind the last chunk has completed (a
      # caller that drops its loader batches right away would otherwise hand the blocks back to
      # torch's pinned-memory allocator while they are still being read)
      _hold_host_rows(copy, [b[0] for b in group] + [b[2] for b in group])
    for t in clips_l + vids_l:
      t.record_stream(copy)        # allocated on the caller's stream, written on the copy stream
    v_sched.meta.record_stream(main)   # the other way round
  for b in group:
    if not pull:
      clips_l.append(_dev_seq(b[0], device, torch.float32))
      vids_l.append(_dev_seq(b[2], device, torch.float32))
    caps_l.append(_dev_seq(b[1], device, torch.int64))
    pars_l.append(_dev_seq(b[3], device, torch.int64))
    len_clip.append(_lens_i64(b[4]))
    len_cap.append(_lens_i64(b[5]))
    len_vid.append(_lens_i64(b[6]))
    len_par.append(_lens_i64(b[7]))
    num_clips.extend(b[8])
    num_caps.extend(b[9])
  n_clip = int(sum(len(l) for l in len_clip))
  n_cap = int(sum(len(l) for l in len_cap))
  n_vid = len(num_clips)
  if n_clip != sum(num_clips) or n_cap != sum(num_caps):
    raise ValueError('batch contract violated: sum(num_clips) != number of clip rows')

  clip_rnn, txt_rnn = model.clip_enc.rnn, model.txt_enc.rnn
  H1v = clip_rnn.rnn.weight_hh_l0.shape[1]
  H1t = txt_rnn.rnn.weight_hh_l0.shape[1]
  img_dim = clips_l[0].shape[2]
  table = model.txt_enc.embed.weight.detach()

  def level2(enc, rows, counts, ctx_rows, Hin, tower):
    counts = np.asarray(counts, dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
    x_ptrs = np.uint64(rows.data_ptr()) + starts * np.uint64(Hin * 4)
    h0 = ops.padded_row_ptrs(ctx_rows) if contextual_model else None
    return enc.rnn.forward_ptrs(counts, Hin, device, x_ptrs=x_ptrs, h0_ptrs=h0, step_plan=sp.get(tower))

  n = ops.l2norm_rows

  def visual_tower():
    # level 1: clips of all batches, then whole-video streams of all batches (same weights)
    ptrs = ops.seq_row_ptrs_many(clips_l + vids_l)
    lens = np.concatenate(len_clip + len_vid)
    vis = clip_rnn.forward_ptrs(lens, img_dim, device, x_ptrs=ptrs, step_plan=sp.get('v1'))
    clip_emb, vid_ctx = vis[:n_clip], vis[n_clip:]
    # level 2: each video's clips are consecutive rows of clip_emb -> addressed in place
    vid_emb = level2(model.vid_seq_enc, clip_emb, num_clips, vid_ctx, H1v, 'v2')
    return n(vid_emb), n(clip_emb), n(vid_ctx)

  def text_tower():
    # level 1: sentences, then paragraphs (embedding lookup fused into the operand load)
    ptrs = ops.seq_row_ptrs_many(caps_l + pars_l)
    lens = np.concatenate(len_cap + len_par)
    txt = txt_rnn.forward_ptrs(lens, table.shape[1], device, tok_ptrs=ptrs, table=table,
                               step_plan=sp.get('t1'))
    cap_emb, para_ctx = txt[:n_cap], txt[n_cap:]
    para_emb = level2(model.txt_seq_enc, cap_emb, num_caps, para_ctx, H1t, 't2')
    return n(para_emb), n(cap_emb), n(para_ctx)

  if TWO_STREAMS[0]:
    # The two towers are independent until the loss.  On two HIP streams the dispatcher can fill
    # the partial last round of one tower's step grid, and the CUs idle during the other's short
    # ragged-tail launches, with the other tower's workgroups.
    main = torch.cuda.current_stream(device)
    s_vis, s_txt = _side_streams(device)
    s_vis.wait_stream(main)
    s_txt.wait_stream(main)
    with torch.cuda.stream(s_txt):
      out_txt = text_tower()
    with torch.cuda.stream(s_vis):
      out_vis = visual_tower()
    main.wait_stream(s_vis)
    main.wait_stream(s_txt)
    for t in out_vis + out_txt:
      t.record_stream(main)      # allocated on a side stream, consumed on the caller's stream
  elif GROUP_TOWERS[0]:
    # The two towers are independent: step t of both level-1 encoders shares one launch
    # (cmhse_gru_pool_fwd_multi), then step t of both level-2 encoders.
    v_ptrs = ops.seq_row_ptrs_many(clips_l + vids_l)
    t_ptrs = ops.seq_row_ptrs_many(caps_l + pars_l)
    # The visual chain (<= 80 frames) ends long before the text chain (paragraphs of hundreds of
    # tokens, a handful of sequences per step by then): the text tail continues on a separate (own hardware queue; ops.stream_set)
    # side stream while the visual attention pass runs on this one, instead of after it.
    tail = _tail_stream(device) if EARLY_POOL[0] else None
    v_lens, t_lens = np.concatenate(len_clip + len_vid), np.concatenate(len_cap + len_par)
    t_sched = None
    if plan is not None and not pull:     # resident batches: the schedules outlive this pass
      v_sched = ops.SeqSchedule(v_lens, device, x_ptrs=v_ptrs)
      t_sched = ops.SeqSchedule(t_lens, device, tok_ptrs=t_ptrs)
      plan.clear()
      plan.update(key=_plan_key(group), v_sched=v_sched, t_sched=t_sched, v_lens=v_lens,
                  t_lens=t_lens, v_ptrs=v_ptrs, t_ptrs=t_ptrs, img_dim=img_dim, n_clip=n_clip,
                  n_cap=n_cap, n_vid=n_vid, num_clips=np.asarray(num_clips, dtype=np.int64),
                  num_caps=np.asarray(num_caps, dtype=np.int64),
                  batch_sizes=[len(b[8]) for b in group], keep=(clips_l, vids_l, caps_l, pars_l))
    ready = None if stage is None else stage.tower_events()
    (vis, _), (txt, _) = ops.gru_pool_fwd_multi([
        clip_rnn.request_ptrs(v_lens, img_dim, device, x_ptrs=v_ptrs,
                              sched=v_sched, step_events=v_events, step_plan=sp.get('v1')),
        txt_rnn.request_ptrs(t_lens, table.shape[1], device,
                             tok_ptrs=t_ptrs, table=table, sched=t_sched,
                             step_plan=sp.get('t1'))], tail_stream=tail, ready_events=ready)
    clip_emb, vid_ctx = vis[:n_clip], vis[n_clip:]
    cap_emb, para_ctx = txt[:n_cap], txt[n_cap:]
    lvl1 = None if stage is None else stage.level1(ready, clip_emb, cap_emb, vid_ctx, para_ctx)

    def level2_request(enc, rows, counts, ctx_rows, Hin, tower):
      counts = np.asarray(counts, dtype=np.int64)
      starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
      x_ptrs = np.uint64(rows.data_ptr()) + starts * np.uint64(Hin * 4)
      h0 = ops.padded_row_ptrs(ctx_rows) if contextual_model else None
      return enc.rnn.request_ptrs(counts, Hin, device, x_ptrs=x_ptrs, h0_ptrs=h0,
                                  step_plan=sp.get(tower))

    (vid_emb, _), (para_emb, _) = ops.gru_pool_fwd_multi([
        level2_request(model.vid_seq_enc, clip_emb, num_clips, vid_ctx, H1v, 'v2'),
        level2_request(model.txt_seq_enc, cap_emb, num_caps, para_ctx, H1t, 't2')])
    if lvl1 is not None:
      out_vis = (n(vid_emb), lvl1['clip_emb'], lvl1['vid_ctx'])
      out_txt = (n(para_emb), lvl1['cap_emb'], lvl1['para_ctx'])
    else:
      out_vis = (n(vid_emb), n(clip_emb), n(vid_ctx))
      out_txt = (n(para_emb), n(cap_emb), n(para_ctx))
  else:
    out_vis = visual_tower()
    out_txt = text_tower()
  return dict(vid_emb=out_vis[0], para_emb=out_txt[0], clip_emb=out_vis[1], cap_emb=out_txt[1],
              vid_ctx=out_vis[2], para_ctx=out_txt[2], n_vid=n_vid,
              batch_sizes=[len(b[8]) for b in group])


def _encode_group_planned(model, group, contextual_model, device, plan, sp, stage=None):
  """encode_group's grouped schedule with the level-1 schedules of an earlier pass over the same
  batches (encode_group(plan=...)): same launches, same values."""
  clip_rnn, txt_rnn = model.clip_enc.rnn, model.txt_enc.rnn
  H1v = clip_rnn.rnn.weight_hh_l0.shape[1]
  H1t = txt_rnn.rnn.weight_hh_l0.shape[1]
  table = model.txt_enc.embed.weight.detach()
  n_clip, n_cap = plan['n_clip'], plan['n_cap']
  tail = _tail_stream(device) if EARLY_POOL[0] else None
  ready = None if stage is None else stage.tower_events()
  (vis, _), (txt, _) = ops.gru_pool_fwd_multi([
      clip_rnn.request_ptrs(plan['v_lens'], plan['img_dim'], device, x_ptrs=plan['v_ptrs'],
                            sched=plan['v_sched'], step_plan=sp.get('v1')),
      txt_rnn.request_ptrs(plan['t_lens'], table.shape[1], device, tok_ptrs=plan['t_ptrs'],
                           table=table, sched=plan['t_sched'], step_plan=sp.get('t1'))],
      tail_stream=tail, ready_events=ready)
  clip_emb, vid_ctx = vis[:n_clip], vis[n_clip:]
  cap_emb, para_ctx = txt[:n_cap], txt[n_cap:]
  lvl1 = None if stage is None else stage.level1(ready, clip_emb, cap_emb, vid_ctx, para_ctx)

  def level2_request(enc, rows, counts, ctx_rows, Hin, tower):
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
    x_ptrs = np.uint64(rows.data_ptr()) + starts * np.uint64(Hin * 4)
    h0 = ops.padded_row_ptrs(ctx_rows) if contextual_model else None
    return enc.rnn.request_ptrs(counts, Hin, device, x_ptrs=x_ptrs, h0_ptrs=h0,
                                step_plan=sp.get(tower))

  (vid_emb, _), (para_emb, _) = ops.gru_pool_fwd_multi([
      level2_request(model.vid_seq_enc, clip_emb, plan['num_clips'], vid_ctx, H1v, 'v2'),
      level2_request(model.txt_seq_enc, cap_emb, plan['num_caps'], para_ctx, H1t, 't2')])
  n = ops.l2norm_rows
  if lvl1 is None:
    lvl1 = dict(clip_emb=n(clip_emb), cap_emb=n(cap_emb), vid_ctx=n(vid_ctx), para_ctx=n(para_ctx))
  return dict(vid_emb=n(vid_emb), para_emb=n(para_emb), n_vid=plan['n_vid'],
              batch_sizes=plan['batch_sizes'], **lvl1)


@contextlib.contextmanager
def _no_gc_pause():
  """No cyclic garbage collection while a pass's launches are being queued: the GPU has nothing
  to do until the first of them arrives, and a full collection over a training process's objects
  was measured at 70-90 ms in that spot (tools/pass_jitter.py; a pass is 281).  The collector is
  switched back on right after (the pass allocates a few thousand short-lived objects)."""
  was = gc.isenabled()
  gc.disable()
  try:
    yield
  finally:
    if was:
      gc.enable()


def _group_batches(batches, max_bytes):
  """Split the loader's batches into super-batches of at most `max_bytes` of padded features."""
  groups, cur, cur_bytes = [], [], 0
  for b in batches:
    nbytes = b[0].numel() * 4 + b[2].numel() * 4
    if cur and cur_bytes + nbytes > max_bytes:
      groups.append(cur)
      cur, cur_bytes = [], 0
    cur.append(b)
    cur_bytes += nbytes
  if cur:
    groups.append(cur)
  return groups


def encode_data_device(opt, model, data_loader, log_step=10, logging=print, contextual_model=True,
                       superbatch_bytes=48 << 30, defer_logging=False, plan=None, step_plan=None,
                       stage=None):
  """Device-resident core of encode_data: returns (dict of six [N,*] normalised embedding tensors
  on the GPU, num_clips_total, cur_vid_total).  With `defer_logging` the per-batch 'Letest' values
  travel to the host asynchronously and a fourth return value, `finish()`, replays the logger
  updates: a caller that scores the embeddings right away (bench.py, parallel_eval) queues its
  ranking kernels first and calls finish() afterwards, so the GPU does not idle through that
  device-to-host round trip.  `plan`: a dict kept by a caller that encodes the SAME resident
  batches again and again (encode_group): schedules built once.  `step_plan`: split_step_plan() of
  the whole split when `data_loader` is only a share of it (parallel_eval passes the one all ranks
  agree on); by default the plan of `data_loader` itself, so that cutting it into several
  super-batches (`superbatch_bytes`) does not change a bit of any embedding.  `stage`: a HostStage
  that receives every group's six matrices as soon as they are queued (encode_data)."""
  batch_time = AverageMeter()
  val_logger = LogCollector()
  model.val_start(opt)
  end = time.time()
  outs = {k: [] for k in ['vid_emb', 'para_emb', 'clip_emb', 'cap_emb', 'vid_ctx', 'para_ctx']}
  num_clips_total, cur_vid_total = [], []
  batches = list(data_loader)
  n_loader = len(batches)
  pending, state = [], {'i': 0, 'end': end}

  def flush():
    """Replay model.logger.update('Letest', value, batch size) (model.py:291) for the groups whose
    loss values have reached the host, batch by batch in loader order."""
    while pending:
      group, sizes, host, ev, block = pending.pop(0)
      if ev is not None:
        ev.synchronize()
      values = host.tolist()
      if ev is not None:
        _HOST_SCALARS.append(block)    # read: the pinned block may carry the next group's values
        _HOST_EVENTS.append(ev)        # ... and its event the next group's copy
      for b, bs, lv in zip(group, sizes, values):
        val_logger.update('Letest', lv, bs)
        batch_time.update(time.time() - state['end'])
        state['end'] = time.time()
        if state['i'] % log_step == 0:
          logging('Test: [{0}/{1}]\t{e_log}\tTime {batch_time.val:.3f} ({batch_time.avg:.3f})\t'
                  .format(state['i'], n_loader, batch_time=batch_time, e_log=str(val_logger)))
        state['i'] += 1

  with torch.no_grad(), _no_gc_pause():
    groups = _group_batches(batches, superbatch_bytes)
    if step_plan is None and len(groups) > 1:       # (one group: its own step counts ARE the split's)
      step_plan = split_step_plan(batches)
    for gi, group in enumerate(groups):
      model.logger = val_logger                     # evaluation.py:99
      enc = encode_group(model, group, contextual_model,
                         plan=None if plan is None else plan.setdefault(gi, {}), step_plan=step_plan,
                         stage=stage)
      if stage is not None:
        stage.put({k: enc[k] for k in outs})     # (what left early is skipped)
      for k in outs:
        outs[k].append(enc[k])
      # per-loader-batch 'Letest' loss (evaluation.py:129), all on device, one sync per group
      crit = model.criterion
      loss_dev = ops.contrastive_blocks_fwd(enc['vid_emb'], enc['para_emb'], enc['batch_sizes'],
                                            crit.margin, crit.max_violation, crit.norm)
      for b in group:
        num_clips_total.extend(b[8])
        cur_vid_total.extend(b[11])
      if defer_logging:
        block = _host_scalars(loss_dev.numel())
        host = block[:loss_dev.numel()]
        host.copy_(loss_dev.reshape(-1), non_blocking=True)   # (stream-ordered after the loss kernels)
        ev = _HOST_EVENTS.pop() if _HOST_EVENTS else torch.cuda.Event()
        ev.record()
        pending.append((group, enc['batch_sizes'], host, ev, block))
      else:
        pending.append((group, enc['batch_sizes'], loss_dev.cpu(), None, None))
        flush()
  cat = {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in outs.items()}
  if defer_logging:
    return cat, num_clips_total, cur_vid_total, flush
  return cat, num_clips_total, cur_vid_total


MATRICES = ('vid_emb', 'para_emb', 'clip_emb', 'cap_emb', 'vid_ctx', 'para_ctx')
PUSH_KERNEL = [True]     # device-to-host staging by cmhse_push_rows (a few waves) instead of hipMemcpyAsync


_STAGE_EVENTS = []


class HostStage(object):
  """The device-to-host side of encode_data: six page-locked float32 matrices (what the reference
  builds row by row with `.data.cpu()` + list.extend, evaluation.py:120-125,139-144) filled by
  asynchronous copies on the package's copy stream while the encoders and the ranking still run.
  `put(mats)` is called with the CURRENT stream having queued the producers of `mats`; a matrix a
  group has already handed over (the level-1 rows, early) is skipped the second time.  The pinned
  blocks come from torch's caching host allocator: the arrays encode_data returns view them, and
  they go back to the cache when the caller drops those arrays."""

  def __init__(self, rows, dims, device):
    self.device = device
    self.host = {k: torch.empty((rows[k], dims[k]), dtype=torch.float32, pin_memory=True)
                 for k in MATRICES}
    self.filled = {k: 0 for k in MATRICES}
    self.seen = set()
    self.copy = _copy_stream(device)
    self._keep = []      # the staged device tensors: their ids stay unique until wait()
    self._events = _STAGE_EVENTS     # recycled between groups and passes (creating an event beside a
                                     # busy GPU was seen to stall the host for milliseconds, ops.upload)

  def tower_events(self):
    """Two events (visual, text) for gru_pool_fwd_multi(ready_events=): recorded once here so that
    they own a handle; the library re-records each where its tower's level-1 output is final."""
    main = torch.cuda.current_stream(self.device)
    evs = []
    for _ in range(2):
      ev = self._events.pop() if self._events else torch.cuda.Event()
      ev.record(main)
      evs.append(ev)
    return evs

  def level1(self, ready, clip_emb, cap_emb, vid_ctx, para_ctx):
    """Normalise the four level-1 matrices on the copy stream, each tower behind ITS ready event,
    and send them off; returns them ({name: normalised device tensor}).  The caller's stream is
    made to wait for the normalisations (not for the copies) before it returns."""
    main = torch.cuda.current_stream(self.device)
    out = {}
    normed = self._events.pop() if self._events else torch.cuda.Event()
    towers = ((ready[0], (('clip_emb', clip_emb), ('vid_ctx', vid_ctx))),
              (ready[1], (('cap_emb', cap_emb), ('para_ctx', para_ctx))))
    for i, (ev, pair) in enumerate(towers):
      self.copy.wait_event(ev)
      with torch.cuda.stream(self.copy):
        for k, raw in pair:
          raw.record_stream(self.copy)
          out[k] = ops.l2norm_rows(raw)
      if i == len(towers) - 1:
        normed.record(self.copy)      # (behind the first tower's copies, in front of the last one's)
      self._send([(k, out[k]) for k, _ in pair])
    main.wait_event(normed)           # consumers of the returned tensors on the caller's stream
    for t in out.values():
      t.record_stream(main)
    self._events += list(ready) + [normed]
    return out

  def put(self, mats):
    todo = [(k, t) for k, t in mats.items() if k in self.host and id(t) not in self.seen]
    if not todo:
      return
    self.copy.wait_stream(torch.cuda.current_stream(self.device))
    self._send(todo, ops.PUSH_WORKGROUPS_LATE[0])

  def _send(self, todo, workgroups=None):
    for k, t in todo:
      n = t.shape[0]
      dst = self.host[k][self.filled[k]:self.filled[k] + n]
      if PUSH_KERNEL[0]:
        tc = t.contiguous()
        self._keep.append(tc)
        ops.push_rows(tc, dst, self.copy, workgroups)
      else:             # the runtime's copy: a chip-wide blit kernel on this image (A/B only)
        with torch.cuda.stream(self.copy):
          dst.copy_(t, non_blocking=True)
      t.record_stream(self.copy)
      self.filled[k] += n
      self.seen.add(id(t))
    self._keep += [t for _, t in todo]

  def wait(self):
    """Block until every queued copy has landed; returns {name: pinned tensor}."""
    for k in MATRICES:
      if self.filled[k] != self.host[k].shape[0]:
        raise RuntimeError('encode_data: %s received %d of %d rows' % (k, self.filled[k], self.host[k].shape[0]))
    self.copy.synchronize()
    self._keep = []
    return self.host


def _stage_for(model, batches, device):
  """A HostStage sized for `batches` (12-tuples with their count members) and `model`'s widths."""
  n_vid = sum(len(b[8]) for b in batches)
  n_clip = sum(int(sum(b[8])) for b in batches)
  n_cap = sum(int(sum(b[9] if b[9] is not None else b[8])) for b in batches)
  h = lambda enc: int(enc.rnn.rnn.weight_hh_l0.shape[1])
  rows = dict(vid_emb=n_vid, para_emb=n_vid, clip_emb=n_clip, cap_emb=n_cap, vid_ctx=n_vid, para_ctx=n_vid)
  dims = dict(vid_emb=h(model.vid_seq_enc), para_emb=h(model.txt_seq_enc), clip_emb=h(model.clip_enc),
              cap_emb=h(model.txt_enc), vid_ctx=h(model.clip_enc), para_ctx=h(model.txt_enc))
  return HostStage(rows, dims, device)


# What the last encode_data left on the device for the i2t / t2i calls that follow it in
# train.validate (train.py:225-236): the two [N, D] matrices themselves and — queued speculatively
# behind the encoders, under the tail of the device-to-host copies — the ranks of both directions.
# i2t / t2i serve a call from it only when they are handed the very ndarray OBJECTS encode_data
# returned AND the bytes those arrays hold now still equal the device copies (the arrays view pinned
# memory: cmhse_rows_differ reads them in place over PCIe, 0.7 ms for the two 20 MB matrices).  One
# entry, replaced by the next encode_data, dropped when either array is garbage-collected.
TRACE = None           # tools/api_path_profile.py: a list that receives (phase, perf_counter) marks
SPECULATE_RANKS = [True]
SUPERBATCH_BYTES = [48 << 30]    # padded feature bytes per super-batch of encode_data (288 GB of HBM)
_LAST_ENCODE = [None]
CACHE_STATS = {'hits': 0, 'stale': 0}


def _forget_last_encode(_ref=None):
  _LAST_ENCODE[0] = None


def _mark(name):
  if TRACE is not None:
    TRACE.append((name, time.perf_counter()))


def encode_data(opt, model, data_loader, log_step=10, logging=print, contextual_model=True):
  """/root/reference/evaluation.py:80-158: returns (vid_embs, para_embs, clip_embs, cap_embs,
  vid_contexts, para_contexts, num_clips_total, cur_vid_total); the six arrays are float32 NumPy
  like the reference's.  They leave the device through page-locked memory on the copy stream: the
  level-1 matrices (clips, sentences, contexts: 4/5 of the bytes) while level 2 runs, the two
  level-2 matrices under the ranking of both directions, which is queued here already
  (SPECULATE_RANKS) for the i2t / t2i calls train.validate makes next."""
  import weakref
  batches = list(data_loader)
  device = torch.device('cuda', torch.cuda.current_device())
  stage = _stage_for(model, batches, device)
  _mark('pinned matrices allocated')
  cat, num_clips_total, cur_vid_total, finish_log = encode_data_device(
      opt, model, batches, log_step, logging, contextual_model, superbatch_bytes=SUPERBATCH_BYTES[0],
      defer_logging=True, stage=stage)
  _mark('encoders queued')
  ranks_host = ranks_event = None
  if SPECULATE_RANKS[0] and cat['vid_emb'].shape == cat['para_emb'].shape:
    r_i, t_i = ops.sim_rank(cat['vid_emb'], cat['para_emb'])
    r_t, t_t = ops.sim_rank(cat['para_emb'], cat['vid_emb'])
    ranks_host = torch.empty((4, r_i.shape[0]), dtype=torch.int32, pin_memory=True)
    ranks_host.copy_(torch.stack([r_i, t_i, r_t, t_t]), non_blocking=True)
    ranks_event = _HOST_EVENTS.pop() if _HOST_EVENTS else torch.cuda.Event()
    ranks_event.record()
  _mark('ranking queued')
  finish_log()          # the per-batch 'Letest' meters (evaluation.py:129), behind everything queued
  _mark('Letest values on the host (encoders done)')
  host = stage.wait()
  _mark('six matrices on the host')
  arrays = {k: host[k].numpy() for k in MATRICES}
  if ranks_event is not None:
    ranks_event.synchronize()
    _HOST_EVENTS.append(ranks_event)
    _mark('ranks on the host')
    _LAST_ENCODE[0] = dict(
        vid=weakref.ref(arrays['vid_emb'], _forget_last_encode),
        para=weakref.ref(arrays['para_emb'], _forget_last_encode),
        vid_host=host['vid_emb'], para_host=host['para_emb'],
        vid_dev=cat['vid_emb'], para_dev=cat['para_emb'], ranks=ranks_host.numpy())
  else:
    _LAST_ENCODE[0] = None
  return tuple(arrays[k] for k in MATRICES) + (num_clips_total, cur_vid_total)


# ---------------------------------------------------------------------------------------------
# i2t / t2i
# ---------------------------------------------------------------------------------------------
def report_from_ranks(ranks):
  """The six numbers of evaluation.py:173-184 from the 0-based ranks of the matching item:
  recall at 1 / 5 / 50 in percent (the key 'r10' IS Recall@50 upstream: `ranks < 50`, :175),
  median and mean rank 1-based, and the recall sum train.py uses as its model-selection score."""
  ranks = numpy.asarray(ranks)
  recall = {key: 100.0 * numpy.count_nonzero(ranks < k) / len(ranks)
            for key, k in (('r1', 1), ('r5', 5), ('r10', 50))}
  recall['medr'] = numpy.floor(numpy.median(ranks)) + 1
  recall['meanr'] = ranks.mean() + 1
  recall['sum'] = recall['r1'] + recall['r5'] + recall['r10']
  return recall


def _as_device(x, device=None):
  if isinstance(x, torch.Tensor):
    return x if x.is_cuda else x.cuda()
  if not torch.cuda.is_available():
    raise RuntimeError('cmhse_amd.evaluation needs an MI355X (no CPU fallback)')
  t = torch.from_numpy(numpy.ascontiguousarray(x, dtype=numpy.float32))
  # an array that views page-locked memory (encode_data's own) goes up without the pageable stop-over
  return t.to(device or 'cuda', non_blocking=t.is_pinned())


def _from_last_encode(images, captions):
  """The ranks [4, N] int32 the last encode_data queued, when `images` / `captions` are the arrays
  that call returned and still hold what it wrote; else None."""
  e = _LAST_ENCODE[0]
  if e is None or e['vid']() is not images or e['para']() is not captions:
    return None
  _mark('content check queued')
  # the arrays view page-locked memory: the device compares them in place with its own copies (one pass
  # over PCIe, nothing staged)
  differ = ops.rows_differ([(e['vid_host'], e['vid_dev']), (e['para_host'], e['para_dev'])])
  _mark('content check done')
  if not differ:
    CACHE_STATS['hits'] += 1
    return e['ranks']
  CACHE_STATS['stale'] += 1
  _LAST_ENCODE[0] = None         # edited in place since: never again served from here
  return None


def _rank_report(queries, gallery):
  rank, top1 = ops.sim_rank(_as_device(queries), _as_device(gallery))
  ranks = rank.cpu().numpy().astype(numpy.float64)      # the reference stores ranks in float64
  top1 = top1.cpu().numpy().astype(numpy.float64)
  return report_from_ranks(ranks), top1, ranks


def _served(ranks, direction):
  r = ranks[2 * direction].astype(numpy.float64)
  return report_from_ranks(r), ranks[2 * direction + 1].astype(numpy.float64), r


def i2t_t2i(images, captions):
  """Both directions of train.validate's scoring (train.py:234-236: i2t then t2i on the same two
  matrices) with ONE device-to-host copy: the two ranking launches are queued back to back and
  their four int32 vectors come down together.  Returns ((report, top1, ranks) of i2t, same of
  t2i), each exactly what i2t / t2i return."""
  v, p = _as_device(images), _as_device(captions)
  r_i, t_i = ops.sim_rank(v, p)
  r_t, t_t = ops.sim_rank(p, v)
  host = torch.stack([r_i, t_i, r_t, t_t]).cpu().numpy().astype(numpy.float64)
  return ((report_from_ranks(host[0]), host[1], host[0]),
          (report_from_ranks(host[2]), host[3], host[2]))


def i2t(images, captions, npts=None, measure='cosine'):
  """/root/reference/evaluation.py:160-185 (video -> paragraph).  `npts`, `measure` are ignored
  upstream too (:161).  Accepts NumPy arrays (reference contract) or GPU tensors.  Handed the
  arrays the last encode_data returned, unchanged, it reports the ranks that call already
  computed on the device copies (bit-identical: same kernel, same operands)."""
  ranks = _from_last_encode(images, captions)
  if ranks is not None:
    return _served(ranks, 0)
  return _rank_report(images, captions)


def t2i(images, captions, npts=None, measure='cosine'):
  """/root/reference/evaluation.py:188-213 (paragraph -> video)."""
  ranks = _from_last_encode(images, captions)
  if ranks is not None:
    return _served(ranks, 1)
  return _rank_report(captions, images)
