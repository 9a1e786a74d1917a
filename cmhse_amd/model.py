"""Host-side mirror of the reference's `model.py` (VSE and its encoders) on the MI355X hot path.

Class names, constructor arguments, attributes and method signatures follow
/root/reference/model.py:21-99 (encoders) and :102-369 (VSE) so a train.py-style driver
(train.py:124-172,193; evaluation.py:97-129) runs unchanged; every arithmetic step is a HIP
kernel behind include/cmhse_hip.h (forward AND backward), including the reconstruction decoders
(--reconstruct_loss / --lowest_reconstruct_loss) and the weak (group-wise) loss.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .decoder import DecoderSequence, EuclideanLoss
from .layers import Attention, Maxout, Seq2Seq
from .loss import (ContrastiveLoss, GroupWiseContrastiveLoss, contrastive_losses, normalize,
                   step_losses)


def _make_rnn(rnn_type, in_dim, embed_size, bidirectional):
  if rnn_type == 'attention':
    return Attention(in_dim, embed_size, rnn_bidirectional=bidirectional)
  if rnn_type == 'seq2seq':
    return Seq2Seq(in_dim, embed_size, rnn_bidirectional=bidirectional)
  if rnn_type == 'maxout':
    return Maxout(in_dim, embed_size, rnn_bidirectional=bidirectional)
  raise ValueError('Unsupported RNN type')      # model.py:34


class EncoderImage(nn.Module):
  """/root/reference/model.py:21-42."""

  def __init__(self, img_dim, embed_size, bidirectional=False, rnn_type='maxout'):
    super(EncoderImage, self).__init__()
    self.embed_size = embed_size
    self.bidirectional = bidirectional
    self.rnn = _make_rnn(rnn_type, img_dim, embed_size, bidirectional)

  def forward(self, x, lengths):
    return self.rnn(x, lengths)


class EncoderSequence(nn.Module):
  """/root/reference/model.py:44-65 — level-2 encoder, optional initial hidden state."""

  def __init__(self, img_dim, embed_size, bidirectional=False, rnn_type='maxout'):
    super(EncoderSequence, self).__init__()
    self.embed_size = embed_size
    self.bidirectional = bidirectional
    self.rnn = _make_rnn(rnn_type, img_dim, embed_size, bidirectional)

  def forward(self, x, lengths, hidden=None):
    return self.rnn(x, lengths, hidden)


class EncoderText(nn.Module):
  """/root/reference/model.py:67-99.  The word table is loaded from
  `vocab/<data_name>_w2v_total.npz` relative to the cwd when that file exists (model.py:89-90);
  otherwise it keeps nn.Embedding's N(0,1) init (synthetic runs)."""

  def __init__(self, vocab_size, word_dim, embed_size, bidirectional=False, rnn_type='maxout',
               data_name='anet_precomp'):
    super(EncoderText, self).__init__()
    self.embed_size = embed_size
    self.bidirectional = bidirectional
    self.embed = nn.Embedding(vocab_size, word_dim)
    self.rnn = _make_rnn(rnn_type, word_dim, embed_size, bidirectional)
    self.init_weights(data_name)

  def init_weights(self, data_name):
    path = 'vocab/{}_w2v_total.npz'.format(data_name)
    if os.path.exists(path):
      self.embed.weight.data = torch.from_numpy(
          np.load(path)['arr_0'].astype(float)).float()

  def forward(self, x, lengths, return_word=True):
    """Returns (outputs, cap_emb) like model.py:92-99.  The lookup is fused into the GRU's
    operand load; the word tensor is only materialised when `return_word`."""
    outputs = self.rnn.forward_tokens(x, lengths, self.embed.weight)
    cap_emb = None
    if return_word:
      cap_emb = _word_rows(self.embed.weight.detach(), x)
    return outputs, cap_emb


def _base_ptr(t):
  return t.data.data_ptr() if isinstance(t, ops.Ragged) else t.data_ptr()


def _word_rows(table, tokens):
  """table[tokens]: [S, L, word_dim] for padded ids; for a packed batch (ops.Ragged) the word
  vectors of the valid tokens only, back to back, as a Ragged of [sum(lens), word_dim]."""
  if isinstance(tokens, ops.Ragged):
    return ops.Ragged(ops.gather_rows(table, tokens.data), tokens.lens)
  return ops.gather_rows(table, tokens)


# A training step's two towers (encoders and decoders alike) are independent until the losses, and
# at training batch sizes every GRU time step is a short, latency-bound launch.  How the two
# chains are scheduled (TRAIN_SCHEDULE[0]; all five give the same values, and
# bit-identical ones wherever the backward pass has no float atomics — tested):
#   'interleaved'  (default) every tower is a chain on a stream of its own, and the host queues
#                  step t of both before step t + 1 of either: side by side from the first step,
#                  both directions.  The two encoder levels of both towers are ONE autograd node
#                  (layers.run_towers): a tower goes from its level 1 to its level 2, and back,
#                  without meeting the other tower or the caller's stream in between;
#   'levels'       the same, but each level is a node of its own that joins the caller's stream
#                  (what 'interleaved' was before: +0.3 ms forward, +0.45 ms backward of glue);
#   'towers'       one call per tower and level, each tower on its own stream (round 2): the second
#                  tower starts only when the host has queued the whole first one (~100 launches
#                  forward, ~250 backward);
#   'grouped'      step t of both towers in one shared launch on one stream (half the dependent
#                  launches, twice the workgroups each: 15.4 against 14.9 ms per step in round 2 —
#                  a step kernel at a training batch is bound by operand bandwidth per CU);
#   'serial'       one stream, one tower after the other.
TRAIN_SCHEDULE = ['interleaved']
# the 4-7 contrastive losses of a step as one launch set (loss.contrastive_losses); False = one by one
BATCHED_LOSSES = [True]
# normalize + every contrastive term + the weighted total as ONE autograd node
# (loss.step_losses); False = the operator-by-operator form above
FUSED_LOSSES = [True]
# torch.optim.Adam(fused=True): one launch for the whole update (VSE.__init__)
FUSED_ADAM = [True]
# train_emb on the loader's pinned HOST tensors (train.py unchanged: DataLoader(pin_memory=True),
# activity_net/data.py:157-162): the frame features are not copied in front of the step; device
# buffers are allocated unfilled and the rows cross PCIe time-chunk by time-chunk under the visual
# chain (ops.pull_steps on the text tower's companion stream; the projection chunk that covers a
# range of steps waits for exactly its rows).  False = `.cuda(non_blocking=True)` in front of the
# step, as the reference does (model.py:225-227).  A loop that can look one batch ahead should wrap
# its loader in collate.DevicePrefetcher instead: the next batch is then resident before its step
# starts and neither path is taken.
HOST_PULL = [True]
HOST_PULL_CHUNK = [8]
HOST_PULL_STREAM = [None]     # None = ops.copy_stream(device); tools/host_lead.py tries others
# ... and WHICH hand-over a pinned batch gets (HOST_PULL on):
#   'auto'   (default) when the host runs ahead of the GPU — the previous step has not finished when
#            this one is being queued, the normal state of a loop that does not read the logger
#            every step (a step is queued after a third of its run time) — the whole
#            batch is copied by the DMA engine on the copy stream into one of two persistent device
#            slots (collate.DeviceStager) and is resident long before the GPU reaches this step:
#            the look-ahead of collate.DevicePrefetcher without touching the loop; when the GPU is
#            already waiting for the host (a loop that synchronises every step), the chunked pull
#            under the visual chain, which starts the chain after the first steps' rows;
#   'ahead'  always the former;  'pull'  always the latter.
HOST_FEED = ['auto']


def _tower_streams(device):
  """The two towers' streams: [0] and [1] of the package's stream set (ops.stream_set)."""
  st = ops.stream_set(device)
  return st[0], st[1]


# tools/train_timeline.py sets this to a list: (name, host perf_counter, event on the current
# stream) at the phase boundaries of a training step.  None (default) = no instrumentation.
TRACE = None


def _tick(name):
  if TRACE is not None:
    import time
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    TRACE.append((name, time.perf_counter(), ev))


class VSE(object):
  """/root/reference/model.py:102-369."""

  def __init__(self, opt):
    self.reconstruct_loss = bool(getattr(opt, 'reconstruct_loss', False))
    self.lowest_reconstruct_loss = bool(getattr(opt, 'lowest_reconstruct_loss', False))
    if self.lowest_reconstruct_loss and not self.reconstruct_loss:
      raise ValueError('--lowest_reconstruct_loss needs --reconstruct_loss (model.py:324-326)')
    if not torch.cuda.is_available():
      raise RuntimeError('cmhse_amd.VSE needs an MI355X (no CPU fallback for the hot path)')
    self.norm = opt.norm
    self.grad_clip = opt.grad_clip
    self.clip_enc = EncoderImage(opt.img_dim, opt.img_first_size, rnn_type=opt.rnn_type)
    self.txt_enc = EncoderText(opt.vocab_size, opt.word_dim, opt.cap_first_size,
                               rnn_type=opt.rnn_type, data_name=opt.data_name)
    self.vid_seq_enc = EncoderSequence(opt.img_first_size, opt.embed_size, rnn_type=opt.rnn_type)
    self.txt_seq_enc = EncoderSequence(opt.cap_first_size, opt.embed_size, rnn_type=opt.rnn_type)
    for enc in (self.clip_enc, self.txt_enc, self.vid_seq_enc, self.txt_seq_enc):
      enc.cuda()

    self.criterion = ContrastiveLoss(margin=opt.margin, measure=opt.measure,
                                     max_violation=opt.max_violation, norm=self.norm)
    self.weak_criterion = GroupWiseContrastiveLoss(margin=opt.margin, measure=opt.measure,
                                                   max_violation=opt.max_violation,
                                                   norm=self.norm)      # model.py:127-129
    params = list(self.txt_enc.parameters())
    params += list(self.clip_enc.parameters())
    params += list(self.vid_seq_enc.parameters())
    params += list(self.txt_seq_enc.parameters())
    decode_rnn_type = getattr(opt, 'decode_rnn_type', 'seq2seq')
    if self.reconstruct_loss:       # model.py:138-149
      self.vid_seq_dec = DecoderSequence(opt.embed_size, opt.img_first_size,
                                         rnn_type=decode_rnn_type).cuda()
      self.txt_seq_dec = DecoderSequence(opt.embed_size, opt.cap_first_size,
                                         rnn_type=decode_rnn_type).cuda()
      self.criterion_Euclid_Distance = EuclideanLoss(norm=self.norm)
      params += list(self.vid_seq_dec.parameters())
      params += list(self.txt_seq_dec.parameters())
    if self.lowest_reconstruct_loss:   # model.py:151-158
      self.clip_seq_dec = DecoderSequence(opt.embed_size, opt.img_dim,
                                          rnn_type=decode_rnn_type).cuda()
      self.sent_seq_dec = DecoderSequence(opt.embed_size, opt.word_dim,
                                          rnn_type=decode_rnn_type).cuda()
      params += list(self.clip_seq_dec.parameters())
      params += list(self.sent_seq_dec.parameters())
    self.params = params
    # model.py:160 — torch.optim.Adam(params, lr).  Same optimizer, same state and param_groups
    # (train.py:269-270 mutates the LR through them); `fused=True` selects torch's single-launch
    # implementation of the same update rule (15 small launches less at the end of every step:
    # 12.4 -> 12.2 ms).  It is equal to the default implementation within rounding, not guaranteed
    # bit for bit, and keeps state['step'] as a device tensor; FUSED_ADAM[0] = False selects torch's
    # default (what the parity tests against the reference's optimiser trajectory use).
    fused = FUSED_ADAM[0] and all(p.is_cuda for p in params)
    self.optimizer = (torch.optim.Adam(params, lr=opt.learning_rate, fused=True) if fused
                      else torch.optim.Adam(params, lr=opt.learning_rate))
    ops.stream_set(next(iter(params)).device)     # bind the package's streams to hardware queues now
    self.Eiters = 0
    self.logger = None
    self._pending_log = None
    self._log_slot, self._log_pinned, self._log_owner = 0, [None, None], [None, None]
    self._loss_weights = {}     # weight vectors of the batched losses, by number of terms
    self._mid_step_hook = None  # called between the forward and backward passes (collate.DevicePrefetcher)
    # host-fed steps (HOST_FEED): the device slots of the batch hand-over, the slot this step reads,
    # and an event behind every step's last launch (is the GPU still busy when the next is queued?)
    self._stager, self._stage_slot, self._step_done, self._step_done_pool = None, None, None, []
    self._host_rows, self._host_ev_pool = [], []    # (event behind the last pull chunk, the pinned tensors the pulls read)

  # -- checkpoint contract: a LIST of 4 / 6 / 8 state-dicts (model.py:166-191) ------------------
  def _modules(self):
    mods = [self.clip_enc, self.txt_enc, self.vid_seq_enc, self.txt_seq_enc]
    if self.reconstruct_loss:
      mods += [self.vid_seq_dec, self.txt_seq_dec]
    if self.lowest_reconstruct_loss:
      mods += [self.clip_seq_dec, self.sent_seq_dec]
    return mods

  def state_dict(self, opt=None):
    return [m.state_dict() for m in self._modules()]

  def load_state_dict(self, state_dict, opt=None):
    for m, sd in zip(self._modules(), state_dict):
      m.load_state_dict(sd)

  def train_start(self, opt=None):
    self._settle_logger()
    for m in self._modules():
      m.train()

  def val_start(self, opt=None):
    self._settle_logger()
    for m in self._modules():
      m.eval()

  def _settle_logger(self):
    """Mode switches are where a train.py-style loop changes loggers (train.py:189, evaluation.py:99):
    whatever the current one still expects (the last step's loss values, a queued tb_log) is
    delivered first."""
    settle = getattr(self.logger, 'settle', None)
    if settle is not None:
      settle()

  # -- forward --------------------------------------------------------------------------------
  def forward_emb(self, clips, captions, lengths_clip, lengths_cap, return_word=False):
    """model.py:222-236."""
    clips = clips.cuda(non_blocking=True)
    captions = captions.cuda(non_blocking=True)
    clip_emb = self.clip_enc(clips, lengths_clip)
    cap_emb, word = self.txt_enc(captions, lengths_cap, return_word=return_word)
    if return_word:
      return clip_emb, cap_emb, word
    return clip_emb, cap_emb

  def structure_emb(self, clip_emb, cap_emb, num_clips, num_caps, vid_context=None,
                    para_context=None):
    """model.py:238-255.  The reference scatters consecutive rows of clip_emb into a zero-padded
    [B, max(num_clips), H] tensor with a Python loop; here each video's sequence is addressed in
    place (its rows are already consecutive), so no copy is made."""
    vid_emb = self._level2(self.vid_seq_enc, clip_emb, num_clips, vid_context)
    para_emb = self._level2(self.txt_seq_enc, cap_emb, num_caps, para_context)
    return vid_emb, para_emb

  @staticmethod
  def _level2(enc, rows, counts, context):
    return enc.rnn.forward_rows(rows, counts, context)

  def reconstruct_emb(self, vid_emb, para_emb, num_clips, num_caps):
    """model.py:257-270: decode every video / paragraph embedding back into its clips /
    sentences.  The repeated-embedding input tensor of the reference is never built."""
    clip_emb = self.vid_seq_dec.forward_repeat(vid_emb, num_clips)
    sent_emb = self.txt_seq_dec.forward_repeat(para_emb, num_caps)
    return clip_emb, sent_emb

  def lowest_reconstruct_emb(self, vid_emb, para_emb, num_clips, num_caps):
    """model.py:272-285 (called with the reconstructed clip / sentence embeddings and the
    frame / word counts)."""
    frame_emb = self.clip_seq_dec.forward_repeat(vid_emb, num_clips)
    word_emb = self.sent_seq_dec.forward_repeat(para_emb, num_caps)
    return frame_emb, word_emb

  # -- logger protocol (model.py:291: `self.logger.update('Le'+name, loss.item(), n)`) -------------
  # The reference pays one host sync per loss (3-9 per step).  Inside train_emb the (name, device
  # scalar, n) triples are queued instead and leave the device as ONE copy into pinned memory,
  # issued once the backward pass and the Adam update have been queued.  A LogCollector of this
  # package takes the copy while it is still in flight (LogCollector.defer) and replays it in the
  # reference's order before anything reads the collector, so the host goes on to queue the next
  # step while this one runs (the GPU was idle 0.45 ms per step waiting for the host otherwise);
  # any other logger object gets the values before train_emb returns.
  def _log(self, key, loss, n):
    if self._pending_log is not None:
      self._pending_log.append((key, loss.detach(), n))
    else:
      self.logger.update(key, loss.item(), n)

  def _flush_log(self):
    pending, self._pending_log = self._pending_log, None
    if not pending:
      return
    logger = self.logger
    stacked = torch.stack([v.reshape(()) for _, v, _ in pending])
    if not hasattr(logger, 'defer'):
      for (key, _, n), v in zip(pending, stacked.cpu().tolist()):
        logger.update(key, v, n)
      return
    logger.settle()            # the previous step's copy: long done, the GPU is a step behind us
    slot = self._log_slot = (self._log_slot + 1) % 2
    bufs, owners = self._log_pinned, self._log_owner
    if owners[slot] is not None and owners[slot] is not logger:
      owners[slot].settle()    # model.logger was swapped since: its copy still reads this buffer
    owners[slot] = logger
    if bufs[slot] is None or bufs[slot].numel() < len(pending):
      bufs[slot] = torch.empty(max(16, len(pending)), dtype=torch.float32, pin_memory=True)
    host = bufs[slot][:len(pending)]
    host.copy_(stacked, non_blocking=True)
    done = torch.cuda.Event()
    done.record()

    def replay():
      done.synchronize()
      for (key, _, n), v in zip(pending, host.tolist()):
        logger._update(key, v, n)
    logger.defer(replay)

  def forward_weak_loss(self, clip_emb, cap_emb, num_clips, num_caps, name, **kwargs):
    """model.py:294-299."""
    loss = self.weak_criterion(clip_emb, cap_emb, num_clips, num_caps)
    self._log('Le' + name, loss, clip_emb.size(0))
    return loss

  def forward_reconstruct_loss(self, clip_recon, clip_emb, name, **kwargs):
    """model.py:301-306."""
    loss = self.criterion_Euclid_Distance(clip_recon, clip_emb)
    self._log('Le' + name, loss, clip_emb.size(0))
    return loss

  def forward_loss(self, clip_emb, cap_emb, name, **kwargs):
    """model.py:287-292."""
    loss = self.criterion(clip_emb, cap_emb)
    self._log('Le' + name, loss, clip_emb.size(0))
    return loss

  def prepare_batch(self, batch):
    """Host work of a training step that depends on the batch alone, done AHEAD of the step
    (collate.DevicePrefetcher(loader, prepare=model.prepare_batch) calls it one step early): the
    level-1 schedules of both towers — sort by length, step counts, per-sequence address tables,
    their upload (layers.py:94-97 does this inside every forward).  They ride on the batch's first
    member and train_losses picks them up; a batch that is not device-resident is left alone.
    Returns `batch`."""
    clips, captions, videos, paragraphs = batch[:4]
    if not all(isinstance(t, (torch.Tensor, ops.Ragged)) and t.is_cuda
               for t in (clips, captions, videos, paragraphs)):
      return batch
    if clips.dtype != torch.float32 or videos.dtype != torch.float32 or \
        captions.dtype != torch.int64 or paragraphs.dtype != torch.int64 or \
        not all(t.is_contiguous() for t in (clips, captions, videos, paragraphs)):
      return batch
    lens = lambda a, b: np.concatenate([np.asarray(a, dtype=np.int64).reshape(-1),
                                        np.asarray(b, dtype=np.int64).reshape(-1)])
    device = clips.device
    v_lens, t_lens = lens(batch[4], batch[6]), lens(batch[5], batch[7])
    prep = dict(
        v_sched=ops.SeqSchedule(v_lens, device, x_ptrs=ops.seq_row_ptrs_many([clips, videos])),
        t_sched=ops.SeqSchedule(t_lens, device, tok_ptrs=ops.seq_row_ptrs_many([captions, paragraphs])),
        v_lens=v_lens, t_lens=t_lens,
        ptrs=tuple(_base_ptr(t) for t in (clips, captions, videos, paragraphs)))
    clips._cmhse_prep = prep
    return batch

  def _stage_ahead(self, clips, captions, videos, paragraphs):
    """HOST_FEED 'auto' / 'ahead': the four pinned host members of the batch copied into a device
    slot on the copy stream (see HOST_FEED); returns the device members, or None when this
    hand-over does not apply (not pinned host tensors, HOST_PULL off, 'pull', or — 'auto' — the GPU
    has already finished the previous step and is waiting for this one)."""
    from .collate import DeviceStager
    from .evaluation import _pinned_f32
    mode = HOST_FEED[0]
    if mode not in ('auto', 'ahead', 'pull'):
      raise ValueError("model.HOST_FEED: 'auto' | 'ahead' | 'pull'")
    if not HOST_PULL[0] or mode == 'pull' or not (_pinned_f32(clips) and _pinned_f32(videos)):
      return None
    big = [clips, captions, videos, paragraphs]
    if not DeviceStager.stageable(big) or not all(t.is_pinned() for t in (captions, paragraphs)):
      return None
    if mode == 'auto' and (self._step_done is None or self._step_done.query()):
      return None                 # the GPU is idle: start the chain on the first rows instead
    device = next(iter(self.params)).device
    if self._stager is None:
      self._stager = DeviceStager()
    staged, ready, slot = self._stager.stage(big, device, HOST_PULL_STREAM[0] or ops.copy_stream(device))
    torch.cuda.current_stream(device).wait_event(ready)
    self._stage_slot = slot
    return staged

  def _release_host_rows(self, keep=6):
    """Drop the host tensors whose pulls have run (see _pull_visual); never hold more than `keep`."""
    rows = self._host_rows
    while rows and (rows[0][0].query() or len(rows) > keep):
      if not rows[0][0].query():
        rows[0][0].synchronize()
      self._host_ev_pool.append(rows.pop(0)[0])

  def _pull_visual(self, clips, videos, lengths_clip, lengths_video):
    """The hand-over of a host-fed training step (HOST_PULL): when `clips` and `videos` are the
    loader's pinned float32 host tensors (padded, or the ops.Ragged members of collate_packed),
    returns (device clips, device videos, schedule, {step: event}) — device storage allocated but
    not filled, the level-1 visual schedule built with the host rows as sources, and one
    cmhse_pull_steps launch per time chunk queued on the copy stream; None
    otherwise (resident tensors, pageable memory, HOST_PULL off).  The pulls go to the package's
    copy stream (ops.copy_stream)."""
    from .evaluation import _empty_like_on, _pinned_f32
    if not (HOST_PULL[0] and _pinned_f32(clips) and _pinned_f32(videos)):
      return None
    device = next(iter(self.params)).device
    main = torch.cuda.current_stream(device)
    copy = HOST_PULL_STREAM[0] or ops.copy_stream(device)
    lens = np.concatenate([np.asarray(lengths_clip, dtype=np.int64).reshape(-1),
                           np.asarray(lengths_video, dtype=np.int64).reshape(-1)])
    # Everything of the hand-over lives on the copy stream — the device buffers come from ITS pool —
    # so nothing here waits for the caller's stream: the host runs about a step ahead of the GPU,
    # and the rows of step k + 1 start crossing PCIe while step k's backward pass still computes.
    # (A block of that pool is reused only when the streams that read it have passed its release:
    # record_stream below; the towers' streams are joined into the caller's before the step ends.)
    self._release_host_rows()
    with torch.cuda.stream(copy):
      cd, vd = _empty_like_on(clips, device), _empty_like_on(videos, device)
      sched = ops.SeqSchedule(lens, device, x_ptrs=ops.seq_row_ptrs_many([cd, vd]),
                              src_ptrs=ops.seq_row_ptrs_many([clips, videos]))
      events = ops.pull_steps(sched, int(clips.shape[2]), copy, HOST_PULL_CHUNK[0])
    # The pull kernels read the loader's pinned tensors by ADDRESS, possibly after this call has
    # returned (the host runs ahead of the GPU) — torch's pinned-memory allocator knows nothing of
    # them and would hand the blocks to the loader's next batch as soon as the loop drops this one.
    # Keep the tensors until an event recorded behind the last chunk has completed.
    ev = self._host_ev_pool.pop() if self._host_ev_pool else torch.cuda.Event()
    ev.record(copy)
    self._host_rows.append((ev, clips, videos))
    for t in (cd, vd, sched.meta):
      t.record_stream(main)          # allocated on the copy stream, consumed on the caller's (and its forks)
    self._pull_on_s3 = copy.cuda_stream == ops.stream_set(device)[3].cuda_stream
    return cd, vd, sched, events

  def train_losses(self, opts, clips, captions, videos, paragraphs, lengths_clip, lengths_cap,
                   lengths_video, lengths_paragraph, num_clips, num_caps, ind=None, cur_vid=None,
                   *args):
    """Forward half of train_emb (model.py:319-344): embeddings and the 4-7 contrastive losses,
    logged exactly like the reference.  Returns the total loss tensor."""
    # model.py:319-320 run clip_enc on the clips and again on the whole-video streams (same for
    # txt_enc on sentences / paragraphs).  The sequences are independent, so both batches go
    # through each encoder in ONE packed pass: max(T) steps instead of T_clip + T_video.
    schedule = TRAIN_SCHEDULE[0]
    if schedule not in ('interleaved', 'levels', 'towers', 'grouped', 'serial'):
      raise ValueError('unknown training schedule %r' % (schedule,))
    v_sched = v_events = None
    pulled = None
    if schedule in ('interleaved', 'levels'):
      staged = self._stage_ahead(clips, captions, videos, paragraphs)
      if staged is not None:
        clips, captions, videos, paragraphs = staged
      else:
        pulled = self._pull_visual(clips, videos, lengths_clip, lengths_video)
    if pulled is not None:
      clips, videos, v_sched, v_events = pulled
    else:
      clips = clips.cuda(non_blocking=True)
      videos = videos.cuda(non_blocking=True)
    prep = getattr(clips, '_cmhse_prep', None) if pulled is None else None
    captions = captions.cuda(non_blocking=True)
    paragraphs = paragraphs.cuda(non_blocking=True)
    t_sched = None
    if prep is not None and schedule in ('interleaved', 'levels'):
      # schedules built a step ahead (prepare_batch): valid for exactly these tensors and lengths
      same = prep['ptrs'] == tuple(_base_ptr(t) for t in (clips, captions, videos, paragraphs)) and \
          np.array_equal(prep['v_lens'][:len(lengths_clip)], np.asarray(lengths_clip).reshape(-1)) and \
          np.array_equal(prep['v_lens'][len(lengths_clip):], np.asarray(lengths_video).reshape(-1)) and \
          np.array_equal(prep['t_lens'][:len(lengths_cap)], np.asarray(lengths_cap).reshape(-1)) and \
          np.array_equal(prep['t_lens'][len(lengths_cap):], np.asarray(lengths_paragraph).reshape(-1))
      if same:
        v_sched, t_sched = prep['v_sched'], prep['t_sched']
    n_clip, n_cap = clips.shape[0], captions.shape[0]
    lc = np.asarray(lengths_clip, dtype=np.int64)
    lw = np.asarray(lengths_cap, dtype=np.int64)

    def visual_tower():
      _tick('vis:start')
      vis = self.clip_enc.rnn.forward_multi([clips, videos], [lengths_clip, lengths_video])
      _tick('vis:level1')
      clip_emb, vid_context = vis[:n_clip], vis[n_clip:]
      vid_emb = self._level2(self.vid_seq_enc, clip_emb, num_clips, vid_context)
      _tick('vis:level2')
      clip_recon = frame_recon = None
      if self.reconstruct_loss:
        clip_recon = self.vid_seq_dec.forward_repeat(vid_emb, num_clips)
      if self.lowest_reconstruct_loss:
        frame_recon = self.clip_seq_dec.forward_repeat(clip_recon, lc)
      return clip_emb, vid_context, vid_emb, clip_recon, frame_recon

    def text_tower():
      _tick('txt:start')
      txt = self.txt_enc.rnn.forward_tokens_multi([captions, paragraphs],
                                                  [lengths_cap, lengths_paragraph],
                                                  self.txt_enc.embed.weight)
      _tick('txt:level1')
      cap_emb, para_context = txt[:n_cap], txt[n_cap:]
      word = (_word_rows(self.txt_enc.embed.weight.detach(), captions)
              if self.lowest_reconstruct_loss else None)
      para_emb = self._level2(self.txt_seq_enc, cap_emb, num_caps, para_context)
      cap_recon = sent_recon = None
      if self.reconstruct_loss:
        cap_recon = self.txt_seq_dec.forward_repeat(para_emb, num_caps)
      if self.lowest_reconstruct_loss:
        sent_recon = self.sent_seq_dec.forward_repeat(cap_recon, lw)
      return cap_emb, para_context, para_emb, cap_recon, sent_recon, word

    def grouped_towers(streams=None, one_node=False):
      from .layers import run_grouped as _run_grouped
      run_grouped = lambda calls: _run_grouped(calls, streams)
      _tick('vis:start')
      level1 = [self.clip_enc.rnn.call_multi([clips, videos], [lengths_clip, lengths_video],
                                             sched=v_sched, step_events=v_events),
                self.txt_enc.rnn.call_tokens_multi([captions, paragraphs],
                                                   [lengths_cap, lengths_paragraph],
                                                   self.txt_enc.embed.weight, sched=t_sched,
                                                   side=not (v_events is not None and self._pull_on_s3))]
      towers = None
      if one_node:
        # both levels of both towers as one node: each tower stays on its stream between its
        # levels, forward and backward (layers.run_towers)
        from .layers import run_towers
        towers = run_towers([(level1[0], n_clip, self.vid_seq_enc.rnn, num_clips),
                             (level1[1], n_cap, self.txt_seq_enc.rnn, num_caps)], streams)
      if towers is not None:
        (clip_emb, vid_context, vid_emb), (cap_emb, para_context, para_emb) = towers
        _tick('vis:level2')
      else:
        vis, txt = run_grouped(level1)
        _tick('vis:level1')
        clip_emb, vid_context = vis[:n_clip], vis[n_clip:]
        cap_emb, para_context = txt[:n_cap], txt[n_cap:]
        vid_emb, para_emb = run_grouped([
            self.vid_seq_enc.rnn.call_rows(clip_emb, num_clips, vid_context),
            self.txt_seq_enc.rnn.call_rows(cap_emb, num_caps, para_context)])
        _tick('vis:level2')
      word = (_word_rows(self.txt_enc.embed.weight.detach(), captions)
              if self.lowest_reconstruct_loss else None)
      clip_recon = cap_recon = frame_recon = sent_recon = None
      if self.reconstruct_loss:
        clip_recon, cap_recon = run_grouped([
            self.vid_seq_dec.rnn.call_repeat(vid_emb, num_clips),
            self.txt_seq_dec.rnn.call_repeat(para_emb, num_caps)])
      if self.lowest_reconstruct_loss:
        frame_recon, sent_recon = run_grouped([
            self.clip_seq_dec.rnn.call_repeat(clip_recon, lc),
            self.sent_seq_dec.rnn.call_repeat(cap_recon, lw)])
      return ((clip_emb, vid_context, vid_emb, clip_recon, frame_recon),
              (cap_emb, para_context, para_emb, cap_recon, sent_recon, word))

    if schedule in ('interleaved', 'levels'):
      out_v, out_t = grouped_towers(_tower_streams(clips.device), schedule == 'interleaved')
    elif schedule == 'grouped':
      out_v, out_t = grouped_towers()
    elif schedule == 'towers':
      # autograd runs each backward node on its forward's stream, so the two BPTT chains sit on
      # the towers' streams as well
      main = torch.cuda.current_stream()
      s_vis, s_txt = _tower_streams(clips.device)
      s_vis.wait_stream(main)
      s_txt.wait_stream(main)
      with torch.cuda.stream(s_vis):
        out_v = visual_tower()
      with torch.cuda.stream(s_txt):
        out_t = text_tower()
      main.wait_stream(s_vis)
      main.wait_stream(s_txt)
      for t in out_v + out_t:
        if t is not None:
          t.record_stream(main)
      for t in (clips, videos):
        t.record_stream(s_vis)
      for t in (captions, paragraphs):
        t.record_stream(s_txt)
    else:
      out_v = visual_tower()
      out_t = text_tower()
    clip_emb, vid_context, vid_emb, clip_recon, frame_recon = out_v
    cap_emb, para_context, para_emb, cap_recon, sent_recon, word = out_t
    _tick('towers:joined')
    n = normalize
    weak = opts.low_level_loss and getattr(opts, 'weak_low_level_loss', False)
    if FUSED_LOSSES[0] and not weak:
      # model.py:333-343 as one node: (name, a, b, weight in the total), a / b index `xs`
      xs = [vid_emb, para_emb, vid_context, para_context]
      terms = [('_vid', 0, 1, 1.0), ('_ctx_low_lvel', 2, 3, 1.0), ('_vid_inloss', 0, 0, 0.5),
               ('_para_inloss', 1, 1, 0.5)]
      if opts.low_level_loss:
        xs += [clip_emb, cap_emb]
        terms += [('_low_lvel', 4, 5, 1.0), ('_clip_inloss', 4, 4, 0.5), ('_cap_inloss', 5, 5, 0.5)]
      loss, values = step_losses(self.criterion, xs, [t[1:] for t in terms])
      for k, (name, a, _, _) in enumerate(terms):
        self._log('Le' + name, values[k], xs[a].size(0))
    elif BATCHED_LOSSES[0]:
      nv, npar = n(vid_emb), n(para_emb)
      # model.py:333-343 — the same 4-7 ContrastiveLoss evaluations with the same weights, as ONE
      # launch set forward and one backward (loss.contrastive_losses); logged in the reference's
      # order.  (name, a, b, weight in the total)
      terms = [('_vid', nv, npar, 1.0), ('_ctx_low_lvel', n(vid_context), n(para_context), 1.0),
               ('_vid_inloss', nv, nv, 0.5), ('_para_inloss', npar, npar, 0.5)]
      loss_2 = None
      if opts.low_level_loss:
        nc, ns = n(clip_emb), n(cap_emb)
        if not weak:
          terms.append(('_low_lvel', nc, ns, 1.0))
        terms += [('_clip_inloss', nc, nc, 0.5), ('_cap_inloss', ns, ns, 0.5)]
      values = contrastive_losses(self.criterion, [(a, b) for _, a, b, _ in terms])
      for k, (name, a, _, _) in enumerate(terms):
        if weak and name == '_clip_inloss':      # model.py:338-340: its place in the log
          loss_2 = self.forward_weak_loss(nc, ns, num_clips, num_caps, '_wlow_lvel')
        self._log('Le' + name, values[k], a.size(0))
      w = self._loss_weights.get(len(terms))
      if w is None or w.device != values.device:
        w = torch.tensor([t[3] for t in terms], dtype=torch.float32, device=values.device)
        self._loss_weights[len(terms)] = w
      loss = torch.dot(values, w)
      if loss_2 is not None:
        loss = loss + loss_2
    else:
      nv, npar = n(vid_emb), n(para_emb)
      loss_1 = self.forward_loss(nv, npar, '_vid')
      loss_3 = self.forward_loss(n(vid_context), n(para_context), '_ctx_low_lvel')
      loss_5 = (self.forward_loss(nv, nv, '_vid_inloss') +
                self.forward_loss(npar, npar, '_para_inloss')) / 2
      loss = loss_1 + loss_3 + loss_5
      if opts.low_level_loss:
        nc, ns = n(clip_emb), n(cap_emb)
        if weak:      # model.py:338-340
          loss_2 = self.forward_weak_loss(nc, ns, num_clips, num_caps, '_wlow_lvel')
        else:
          loss_2 = self.forward_loss(nc, ns, '_low_lvel')
        loss_6 = (self.forward_loss(nc, nc, '_clip_inloss') +
                  self.forward_loss(ns, ns, '_cap_inloss')) / 2
        loss = loss + loss_2 + loss_6
    if self.reconstruct_loss:        # model.py:346-348
      loss_recon = (self.forward_reconstruct_loss(clip_recon, clip_emb.detach(), '_clip_recon') +
                    self.forward_reconstruct_loss(cap_recon, cap_emb.detach(), '_cap_recon'))
      loss = loss + loss_recon * opts.weight_recon
    if self.lowest_reconstruct_loss:   # model.py:350-364: targets = valid frames / word vectors
      def valid_rows(t, lens):
        S, T = t.shape[0], t.shape[1]
        row_bytes = t.shape[2] * 4
        idx = np.concatenate([i * T + np.arange(l) for i, l in enumerate(lens)])
        return np.uint64(t.data_ptr()) + idx.astype(np.uint64) * np.uint64(row_bytes)

      def targets(t, lens):   # (addresses of the valid rows, the storage that owns them)
        if isinstance(t, ops.Ragged):   # a packed batch holds exactly the valid rows, in order
          d = ops.seq_keep(t, torch.float32).data
          return (np.uint64(d.data_ptr()) +
                  np.arange(d.shape[0], dtype=np.uint64) * np.uint64(d.shape[1] * 4)), d
        d = t.detach().float().contiguous()
        return valid_rows(d, lens), d
      (clip_rows, clips_c), (word_rows, word_c) = targets(clips, lc), targets(word, lw)
      crit = self.criterion_Euclid_Distance
      l_fr = crit.forward_rows(frame_recon, clip_rows, clips_c)
      self._log('Le_reconstruct_frame_hier', l_fr, int(lc.sum()))
      l_wd = crit.forward_rows(sent_recon, word_rows, word_c)
      self._log('Le_reconstruct_word_hier', l_wd, int(lw.sum()))
      loss = loss + (l_fr + l_wd) * opts.lowest_weight_recon
    return loss

  def _mark_step_done(self):
    """An event behind the step's last launch (two, alternated: no event is created per step), and
    the release point of the slot a staged batch was read from."""
    if len(self._step_done_pool) < 2:
      self._step_done_pool.append(torch.cuda.Event())
    ev = self._step_done_pool[self.Eiters % 2] if len(self._step_done_pool) == 2 else self._step_done_pool[-1]
    ev.record()
    self._step_done = ev
    self._release_host_rows()
    if self._stage_slot is not None:
      from .collate import DeviceStager
      DeviceStager.done(self._stage_slot, torch.cuda.current_stream())
      self._stage_slot = None

  def train_emb(self, opts, clips, captions, videos, paragraphs, lengths_clip, lengths_cap,
                lengths_video, lengths_paragraph, num_clips, num_caps, ind, cur_vid, *args):
    """model.py:309-369: one optimisation step.  Forward, losses and backward run on the HIP
    path; the parameter update is torch.optim.Adam as upstream (model.py:160,369)."""
    self.Eiters += 1
    self.logger.update('Eit', self.Eiters)
    self.logger.update('lr', self.optimizer.param_groups[0]['lr'])
    self.optimizer.zero_grad()
    _tick('step:start')
    self._pending_log = []
    try:
      loss = self.train_losses(opts, clips, captions, videos, paragraphs, lengths_clip,
                               lengths_cap, lengths_video, lengths_paragraph, num_clips, num_caps,
                               ind, cur_vid)
      _tick('losses:done')
      if self._mid_step_hook is not None:
        self._mid_step_hook()      # e.g. the next batch's upload: it runs under this backward pass
      loss.backward()
      _tick('backward:done')
      if self.grad_clip > 0:
        torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip)
      self.optimizer.step()
      _tick('adam:done')
      self._mark_step_done()
    except BaseException:
      # a failed step logs nothing: copying the queued loss values down here could raise a second
      # error (an asynchronous HIP fault) that hides the first, and would record a partial step
      self._pending_log = None
      if self._stage_slot is not None:      # whatever was queued may still read the batch's slot
        from .collate import DeviceStager
        DeviceStager.done(self._stage_slot, torch.cuda.current_stream())
        self._stage_slot = None
      raise
    self._flush_log()
