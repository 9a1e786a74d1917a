"""The loader's batch assembly: `collate_fn` of /root/reference/
activity_net/data.py:114-150 and didemo_dev/data.py:127-165.

Two forms, both taking the list of per-video samples `Dataset.__getitem__` returns
(activity_net/data.py:97-109):

    (clips, captions, video, paragraph, lengths_clip, lengths_cap, num_clip, num_caption, index,
     cur_vid | groups)
       clips      list of [len, img_dim] float tensors (the video's clip segments)
       captions   list of 1-D tensors of token ids (stored as floats by the reference)
       video      [frames <= 80, img_dim]          paragraph  1-D token ids

  collate_fn(samples)      the reference's 12-tuple: every ragged sequence copied into a zero-padded
                           [S, Tmax, ...] host tensor (same values, dtypes and member order).
  collate_packed(samples)  the MI355X form of the same batch: the four big members are `ops.Ragged`
                           views into ONE host buffer that holds the sequences back to back — no
                           padding is written, uploaded or read (clip_enc / txt_enc address each
                           sequence by pointer), and one pinned block crosses PCIe instead of four
                           tensors.  `Ragged.padded()` (cmhse_pad_rows, an index kernel) rebuilds
                           the reference's padded tensor on the device for a caller that wants it.
Both tuples feed VSE.train_emb / evaluation.encode_data unchanged.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def _lengths(samples):
  lengths_clip = torch.cat([torch.as_tensor(s[4]).long().reshape(-1) for s in samples], 0)
  lengths_cap = torch.cat([torch.as_tensor(s[5]).long().reshape(-1) for s in samples], 0)
  lengths_video = torch.tensor([len(s[2]) for s in samples], dtype=torch.int64)
  lengths_paragraph = torch.tensor([len(s[3]) for s in samples], dtype=torch.int64)
  return lengths_clip, lengths_cap, lengths_video, lengths_paragraph


def _tail(samples):
  """Members 8-11 of the 12-tuple: the per-sample counts / ids as tuples (zip(*batch)); DiDeMo's
  last member is the concatenation of the samples' `groups` tensors (didemo_dev/data.py:163)."""
  num_clip = tuple(s[6] for s in samples)
  num_caption = tuple(s[7] for s in samples)
  index = tuple(s[8] for s in samples)
  last = tuple(s[9] for s in samples)
  if len(last) and all(isinstance(g, torch.Tensor) for g in last):
    last = torch.cat(last)
  return num_clip, num_caption, index, last


def collate_fn(samples):
  """activity_net/data.py:114-150.  Row `_cur_ind` of `clips` is clip j of video i, rows in sample
  order; only the first lengths_clip[_cur_ind] steps of a clip are copied (:122), the rest stays 0."""
  lengths_clip, lengths_cap, lengths_video, lengths_paragraph = _lengths(samples)
  img_dim = samples[0][0][0].shape[1]
  clips = torch.zeros(len(lengths_clip), int(lengths_clip.max()), img_dim)
  row = 0
  for s in samples:
    for vid in s[0]:
      end = int(lengths_clip[row])
      clips[row, :end] = torch.as_tensor(vid)[:end]
      row += 1
  videos = torch.zeros(len(samples), int(lengths_video.max()), samples[0][2].shape[1])
  for i, s in enumerate(samples):
    videos[i, :int(lengths_video[i])] = torch.as_tensor(s[2])[:int(lengths_video[i])]
  captions = torch.zeros(len(lengths_cap), int(lengths_cap.max()), dtype=torch.int64)
  row = 0
  for s in samples:
    for cap in s[1]:
      end = int(lengths_cap[row])
      captions[row, :end] = torch.as_tensor(cap)[:end].long()
      row += 1
  paragraphs = torch.zeros(len(samples), int(lengths_paragraph.max()), dtype=torch.int64)
  for i, s in enumerate(samples):
    paragraphs[i, :int(lengths_paragraph[i])] = torch.as_tensor(s[3])[:int(lengths_paragraph[i])].long()
  return (clips, captions, videos, paragraphs, lengths_clip, lengths_cap, lengths_video,
          lengths_paragraph) + _tail(samples)


def collate_packed(samples, pin=False):
  """Same batch, no padding.  Layout of the one host block (8-byte aligned sections):
       [ clip frames | video frames ]  float32 rows of img_dim     [ caption ids | paragraph ids ]  int64
  Members 0-3 of the returned 12-tuple are ops.Ragged views into it; members 4-11 are exactly
  collate_fn's.  A clip contributes its first lengths_clip steps, like collate_fn's copy.
  `pin=True` page-locks the block here (main-process loaders; it needs the HIP runtime, so not
  inside DataLoader worker processes — there leave it False and let DataLoader(pin_memory=True)
  pin the members through Ragged.pin_memory(), as it does for the reference's tensors)."""
  lengths_clip, lengths_cap, lengths_video, lengths_paragraph = _lengths(samples)
  img_dim = int(samples[0][0][0].shape[1])
  lc, lw = lengths_clip.numpy(), lengths_cap.numpy()
  lv, lp = lengths_video.numpy(), lengths_paragraph.numpy()
  n_frames = int(lc.sum() + lv.sum())
  n_tok = int(lw.sum() + lp.sum())
  frame_bytes = n_frames * img_dim * 4
  tok_off = (frame_bytes + 7) // 8 * 8
  block = torch.empty(tok_off + n_tok * 8, dtype=torch.uint8)
  if pin and torch.cuda.is_available():
    block = block.pin_memory()
  frames = block[:frame_bytes].view(torch.float32).view(n_frames, img_dim)
  tokens = block[tok_off:].view(torch.int64)
  row = k = 0
  for s in samples:
    for vid in s[0]:
      frames[row:row + lc[k]] = torch.as_tensor(vid)[:lc[k]]
      row += int(lc[k])
      k += 1
  for i, s in enumerate(samples):
    frames[row:row + lv[i]] = torch.as_tensor(s[2])[:lv[i]]
    row += int(lv[i])
  pos = k = 0
  for s in samples:
    for cap in s[1]:
      tokens[pos:pos + lw[k]] = torch.as_tensor(cap)[:lw[k]].long()
      pos += int(lw[k])
      k += 1
  for i, s in enumerate(samples):
    tokens[pos:pos + lp[i]] = torch.as_tensor(s[3])[:lp[i]].long()
    pos += int(lp[i])
  n_cf, n_ct = int(lc.sum()), int(lw.sum())
  clips = ops.Ragged(frames[:n_cf], lc)
  videos = ops.Ragged(frames[n_cf:], lv)
  captions = ops.Ragged(tokens[:n_ct], lw)
  paragraphs = ops.Ragged(tokens[n_ct:], lp)
  return (clips, captions, videos, paragraphs, lengths_clip, lengths_cap, lengths_video,
          lengths_paragraph) + _tail(samples)


def upload_packed(batch, device=None, non_blocking=True):
  """The `.cuda()` of a collate_packed batch as ONE host-to-device copy of its block; returns the
  12-tuple with device-resident Ragged members (the small members stay on the host, as in the
  reference: lengths are read by the host-side schedule)."""
  clips, captions, videos, paragraphs = batch[:4]
  device = device or torch.device('cuda', torch.cuda.current_device())
  base = clips.data.untyped_storage()
  same = all(t.data.untyped_storage().data_ptr() == base.data_ptr()
             for t in (captions, videos, paragraphs))
  if not same:   # not produced by collate_packed: member by member
    return tuple(t.to(device, non_blocking=non_blocking) for t in batch[:4]) + tuple(batch[4:])
  host = torch.empty(0, dtype=torch.uint8).set_(base)
  dev = host.to(device, non_blocking=non_blocking)

  def view(t):
    off = t.data.data_ptr() - base.data_ptr()
    nbytes = t.data.numel() * t.data.element_size()
    return ops.Ragged(dev[off:off + nbytes].view(t.data.dtype).view(t.data.shape), t.lens)
  return (view(clips), view(captions), view(videos), view(paragraphs)) + tuple(batch[4:])


class DeviceStager(object):
  """Two persistent device slots for the four big members of a loader batch (clips, captions,
  videos, paragraphs), alternated: `stage` copies a batch's host tensors into slot k % 2 on a copy
  stream and returns device views of it, `done` marks the point on the consumer's stream behind
  which the slot may be overwritten.  A training loop then allocates nothing and creates no HIP
  event per step (either can stall the host for tens of ms while the GPU is busy, see ops.upload).
  Padded 12-tuples: four copies; collate_packed batches (ops.Ragged members of ONE block): one."""

  def __init__(self):
    self._slots = [None, None]
    self._k = 0

  def _slot(self, k, device, copy, nbytes):
    slot = self._slots[k % 2]
    if slot is None:
      slot = self._slots[k % 2] = {'buf': None, 'ready': torch.cuda.Event(), 'done': None}
    if slot['buf'] is None or slot['buf'].numel() < nbytes:
      with torch.cuda.stream(copy):
        old, slot['buf'] = slot['buf'], torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
      if old is not None:
        old.record_stream(torch.cuda.current_stream(device))
    return slot

  @staticmethod
  def stageable(big):
    return all(isinstance(t, (torch.Tensor, ops.Ragged)) and not t.is_cuda for t in big)

  def stage(self, big, device, copy):
    """`big`: the four host members.  Returns (device members, ready event, slot): the consumer's
    stream waits for `ready`, and calls done(slot, stream) when its last reader has been queued."""
    k, self._k = self._k, self._k + 1
    datas = [t.data if isinstance(t, ops.Ragged) else t for t in big]
    if not all(d.is_contiguous() for d in datas):
      datas = [d.contiguous() for d in datas]
    sizes = [d.numel() * d.element_size() for d in datas]
    offs, off = [], 0
    for n in sizes:
      offs.append(off)
      off += (n + 255) // 256 * 256
    slot = self._slot(k, device, copy, off)
    with torch.cuda.stream(copy):
      if slot['done'] is not None:
        copy.wait_event(slot['done'])      # the step that last read this slot has been passed
      base = datas[0].untyped_storage()
      one_block = all(d.untyped_storage().data_ptr() == base.data_ptr() for d in datas)
      views = []
      if one_block:                        # collate_packed: ONE copy of the span the members cover
        lo = min(d.data_ptr() for d in datas)
        hi = max(d.data_ptr() + n for d, n in zip(datas, sizes))
        host = torch.empty(0, dtype=torch.uint8).set_(base)[lo - base.data_ptr():hi - base.data_ptr()]
        if slot['buf'].numel() < hi - lo:
          slot = self._slot(k, device, copy, hi - lo)
        slot['buf'][:hi - lo].copy_(host, non_blocking=True)
        offs = [d.data_ptr() - lo for d in datas]
      for d, o, n in zip(datas, offs, sizes):
        v = slot['buf'][o:o + n].view(d.dtype).view(d.shape)
        if not one_block:
          v.copy_(d, non_blocking=True)
        views.append(v)
      slot['ready'].record(copy)
    staged = [ops.Ragged(v, t.lens) if isinstance(t, ops.Ragged) else v for v, t in zip(views, big)]
    return staged, slot['ready'], slot

  @staticmethod
  def done(slot, stream):
    if slot['done'] is None:
      slot['done'] = torch.cuda.Event()
    slot['done'].record(stream)

  def release(self, stream):
    """The slots are about to be dropped while the last steps may still read them."""
    for slot in self._slots:
      if slot is not None and slot['buf'] is not None:
        slot['buf'].record_stream(stream)


class DevicePrefetcher(object):
  """One-batch look-ahead over a loader of 12-tuples: while the training step of batch k runs, batch
  k + 1 is already crossing PCIe on a copy stream, so `train_emb` always receives device-resident
  tensors and runs at resident speed.  What a train.py-style loop changes is one line:

      for i, train_data in enumerate(DevicePrefetcher(train_loader)):      # train.py:185
        model.train_emb(opt, *train_data)                                  # train.py:193

  The reference uploads inside the step (model.py:225-227: `.cuda()` of the loader's pinned tensors,
  activity_net/data.py:157-162); that copy — 79 MB packed / 100 MB padded per batch-32 ICEP step,
  1.4-1.75 ms at the 57 GB/s of the link — would otherwise stand in front of the first kernel.
  Works with collate_fn batches (four tensors) and collate_packed batches (one block, one copy:
  upload_packed).  The four big members are uploaded; members 4-11 (lengths, counts, ids) stay on
  the host, where the schedule reads them.  `prepare(batch)` (optional; VSE.prepare_batch) runs
  right after the upload is queued, one step ahead of its use.
  The device tensors of a yielded batch are views of one of TWO persistent upload slots that
  alternate: a batch stays valid until the loop has drawn two more (train.py's loop keeps none;
  clone what must live longer).  Work the loop body queues on streams other than the current one
  must be joined into it before the body ends (train_emb does).
  The copy stream is the package's own (ops.copy_stream): a stream beside the four of the stream
  set — on [3], which shares the null stream's hardware queue, an upload still in flight at a step
  boundary stands in front of the next step's first launches (+9 % per step, tools/host_lead.py)."""

  def __init__(self, loader, device=None, prepare=None, model=None, when='before', stream=None):
    """`model`: a VSE — shorthand for prepare=model.prepare_batch, and needed for when='mid'.
    `when`: 'before' queues the upload of batch k + 1 before step k is queued (it then runs as soon
    as the copy stream is free, typically under the END of step k - 1); 'mid' queues it from inside
    step k, between its forward and backward passes (VSE.train_emb calls the hook), so that it
    runs under step k's backward pass.  `stream`: the copy stream (default: ops.copy_stream)."""
    self.loader, self.prepare = loader, prepare
    if model is not None and prepare is None:
      self.prepare = model.prepare_batch
    if when not in ('before', 'mid') or (when == 'mid' and model is None):
      raise ValueError("DevicePrefetcher: when = 'before' | 'mid' (the latter needs model=)")
    self.model, self.when, self.stream = model, when, stream
    self._stager = DeviceStager()
    self.device = torch.device(device) if device is not None else None

  def __len__(self):
    return len(self.loader)

  def _stage(self, batch, device, copy):
    big = list(batch[:4])
    if not DeviceStager.stageable(big):
      return batch, None, None     # already resident (or not tensors): handed through
    views, ready, slot = self._stager.stage(big, device, copy)
    staged = tuple(views) + tuple(batch[4:])
    if self.prepare is not None:
      self.prepare(staged)
    return staged, ready, slot

  def __iter__(self):
    device = self.device or torch.device('cuda', torch.cuda.current_device())
    copy = self.stream or ops.copy_stream(device)
    it = iter(self.loader)
    state = {'nxt': None, 'asked': True}

    def stage_next():
      if state['asked']:
        return
      state['asked'] = True
      for following in it:
        state['nxt'] = self._stage(following, device, copy)
        break

    state['asked'] = False
    stage_next()
    try:
      if self.when == 'mid':
        self.model._mid_step_hook = stage_next
      while state['nxt'] is not None:
        (cur, ready, slot), state['nxt'] = state['nxt'], None
        state['asked'] = False
        if self.when == 'before':
          stage_next()               # queue the NEXT upload before this batch's step is queued
        main = torch.cuda.current_stream(device)
        if ready is not None:
          main.wait_event(ready)
        yield cur
        if slot is not None:         # everything the loop body queued reads this slot before here
          DeviceStager.done(slot, torch.cuda.current_stream(device))
        stage_next()                 # 'mid': the step did not call the hook (no train_emb on this batch)
    finally:
      if self.when == 'mid' and self.model._mid_step_hook is stage_next:
        self.model._mid_step_hook = None
      self._stager.release(torch.cuda.current_stream(device))


def split_samples(batch):
  """Inverse of collate_fn for a padded 12-tuple: the per-video samples a Dataset would have
  returned (used by the tests and by synthetic loaders to exercise both collate forms)."""
  clips, captions, videos, paragraphs, lc, lw, lv, lp, num_clips, num_caps, index, last = batch
  lc, lw = np.asarray(lc, dtype=np.int64), np.asarray(lw, dtype=np.int64)
  lv, lp = np.asarray(lv, dtype=np.int64), np.asarray(lp, dtype=np.int64)
  samples, r = [], 0
  per_sample_last = last if not isinstance(last, torch.Tensor) else None
  for i, (nc, nw) in enumerate(zip(num_clips, num_caps)):
    if nc != nw:
      raise ValueError('batch contract: num_clips != num_caps')
    cl = [clips[r + j, :lc[r + j]].clone() for j in range(nc)]
    cp = [captions[r + j, :lw[r + j]].clone().float() for j in range(nc)]   # reference: float ids
    samples.append((cl, cp, videos[i, :lv[i]].clone(), paragraphs[i, :lp[i]].clone().float(),
                    torch.as_tensor(lc[r:r + nc]).long(), torch.as_tensor(lw[r:r + nc]).long(),
                    nc, nw, index[i], per_sample_last[i] if per_sample_last is not None else i))
    r += nc
  return samples
