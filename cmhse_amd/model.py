"""Host-side mirror of the reference's `model.py` (VSE and its encoders) on the MI355X hot path.

Class names, constructor arguments, attributes and method signatures follow
/root/reference/model.py:21-99 (encoders) and :102-369 (VSE) so a train.py-style driver
(train.py:124-172,193; evaluation.py:97-129) runs unchanged; every arithmetic step is a HIP
kernel behind include/cmhse_hip.h (forward AND backward).  Out of scope here: the
reconstruction decoders and the weak (group-wise) loss.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .layers import Attention, Maxout, Seq2Seq
from .loss import ContrastiveLoss, normalize


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
      cap_emb = ops.gather_rows(self.embed.weight.detach(), x)
    return outputs, cap_emb


class VSE(object):
  """/root/reference/model.py:102-369."""

  def __init__(self, opt):
    if getattr(opt, 'reconstruct_loss', False) or getattr(opt, 'lowest_reconstruct_loss', False):
      raise NotImplementedError('reconstruction decoders: SURVEY.md §8(f) row 2 (not built yet)')
    if getattr(opt, 'weak_low_level_loss', False):
      raise NotImplementedError('GroupWiseContrastiveLoss: SURVEY.md §8(f) row 4 (not built yet)')
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
    params = list(self.txt_enc.parameters())
    params += list(self.clip_enc.parameters())
    params += list(self.vid_seq_enc.parameters())
    params += list(self.txt_seq_enc.parameters())
    self.params = params
    self.optimizer = torch.optim.Adam(params, lr=opt.learning_rate)
    self.Eiters = 0
    self.logger = None

  # -- checkpoint contract: a LIST of state-dicts (model.py:166-191) --------------------------
  def state_dict(self, opt=None):
    return [self.clip_enc.state_dict(), self.txt_enc.state_dict(),
            self.vid_seq_enc.state_dict(), self.txt_seq_enc.state_dict()]

  def load_state_dict(self, state_dict, opt=None):
    self.clip_enc.load_state_dict(state_dict[0])
    self.txt_enc.load_state_dict(state_dict[1])
    self.vid_seq_enc.load_state_dict(state_dict[2])
    self.txt_seq_enc.load_state_dict(state_dict[3])

  def train_start(self, opt=None):
    for enc in (self.clip_enc, self.txt_enc, self.vid_seq_enc, self.txt_seq_enc):
      enc.train()

  def val_start(self, opt=None):
    for enc in (self.clip_enc, self.txt_enc, self.vid_seq_enc, self.txt_seq_enc):
      enc.eval()

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

  def forward_loss(self, clip_emb, cap_emb, name, **kwargs):
    """model.py:287-292."""
    loss = self.criterion(clip_emb, cap_emb)
    self.logger.update('Le' + name, loss.item(), clip_emb.size(0))
    return loss

  def train_losses(self, opts, clips, captions, videos, paragraphs, lengths_clip, lengths_cap,
                   lengths_video, lengths_paragraph, num_clips, num_caps, ind=None, cur_vid=None,
                   *args):
    """Forward half of train_emb (model.py:319-344): embeddings and the 4-7 contrastive losses,
    logged exactly like the reference.  Returns the total loss tensor."""
    clip_emb, cap_emb = self.forward_emb(clips, captions, lengths_clip, lengths_cap)
    vid_context, para_context = self.forward_emb(videos, paragraphs, lengths_video,
                                                 lengths_paragraph)
    vid_emb, para_emb = self.structure_emb(clip_emb, cap_emb, num_clips, num_caps, vid_context,
                                           para_context)
    n = normalize
    nv, npar = n(vid_emb), n(para_emb)
    loss_1 = self.forward_loss(nv, npar, '_vid')
    loss_3 = self.forward_loss(n(vid_context), n(para_context), '_ctx_low_lvel')
    loss_5 = (self.forward_loss(nv, nv, '_vid_inloss') +
              self.forward_loss(npar, npar, '_para_inloss')) / 2
    loss = loss_1 + loss_3 + loss_5
    if opts.low_level_loss:
      nc, ns = n(clip_emb), n(cap_emb)
      loss_2 = self.forward_loss(nc, ns, '_low_lvel')
      loss_6 = (self.forward_loss(nc, nc, '_clip_inloss') +
                self.forward_loss(ns, ns, '_cap_inloss')) / 2
      loss = loss + loss_2 + loss_6
    return loss

  def train_emb(self, opts, clips, captions, videos, paragraphs, lengths_clip, lengths_cap,
                lengths_video, lengths_paragraph, num_clips, num_caps, ind, cur_vid, *args):
    """model.py:309-369: one optimisation step.  Forward, losses and backward run on the HIP
    path; the parameter update is torch.optim.Adam as upstream (model.py:160,369)."""
    self.Eiters += 1
    self.logger.update('Eit', self.Eiters)
    self.logger.update('lr', self.optimizer.param_groups[0]['lr'])
    self.optimizer.zero_grad()
    loss = self.train_losses(opts, clips, captions, videos, paragraphs, lengths_clip, lengths_cap,
                             lengths_video, lengths_paragraph, num_clips, num_caps, ind, cur_vid)
    loss.backward()
    if self.grad_clip > 0:
      torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip)
    self.optimizer.step()
