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
