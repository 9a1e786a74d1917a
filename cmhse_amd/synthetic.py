"""Synthetic stand-in for the reference's dataset loaders.

The reference's `activity_net/data.py` / `didemo_dev/data.py` need 30-60 GB of
precomputed features that are not available.  This
module emits batches that honour the same 12-tuple contract as the reference
`collate_fn` (/root/reference/activity_net/data.py:114-150):

    clips f32[sumC, Tc, I] zero-padded     captions  i64[sumC, Lc] zero-padded
    videos f32[B, Tv, I]  zero-padded      paragraphs i64[B, Lp]   zero-padded
    lengths_clip i64[sumC]  lengths_cap i64[sumC]
    lengths_video i64[B]    lengths_paragraph i64[B]
    num_clips tuple[int]*B  num_caps tuple[int]*B   ind tuple[int]  cur_vid tuple[str]

Sizes follow the statistics SURVEY.md §8(d) measured on the reference's caption
JSONs (clips/video histogram of val_1.json, <=80 frames per clip, paragraph =
concatenation of the sentences, activity_net/data.py:54-78).
"""
from __future__ import annotations

import numpy as np
import torch

# clips-per-video histogram of data/captions/val_1.json
ANET_CLIP_HIST = {2: 1018, 3: 2171, 4: 890, 5: 357, 6: 185, 7: 134, 8: 69, 9: 35,
                  10: 19, 11: 14, 12: 12, 13: 6, 15: 3, 18: 1, 21: 2, 25: 1}
# DiDeMo
DIDEMO_CLIP_HIST = {1: 35, 2: 3, 3: 287, 4: 301, 5: 356, 6: 19, 7: 3}
# words/sentence empirical quantiles (probability, value) for ActivityNet
ANET_WORD_Q = [(0.0, 4), (0.10, 8), (0.25, 10), (0.50, 14), (0.75, 18), (0.90, 23),
               (0.99, 38), (1.0, 91)]
DIDEMO_WORD_Q = [(0.0, 3), (0.25, 5), (0.50, 7), (0.75, 10), (0.99, 25), (1.0, 51)]

ANET_VOCAB = 13058    # len(vocab) of anet_total
DIDEMO_VOCAB = 7205


def _sample_hist(rng, hist, n):
  keys = np.array(sorted(hist.keys()))
  p = np.array([hist[k] for k in keys], dtype=np.float64)
  return rng.choice(keys, size=n, p=p / p.sum())


def _sample_quantiles(rng, q, n):
  u = rng.uniform(0.0, 1.0, size=n)
  ps = np.array([a for a, _ in q])
  vs = np.array([b for _, b in q], dtype=np.float64)
  return np.maximum(1, np.rint(np.interp(u, ps, vs))).astype(np.int64)


class SplitSpec(object):
  """Ragged description of a whole split: everything except the feature values."""

  def __init__(self, num_clips, frames_per_clip, frames_per_video, words_per_sent):
    self.num_clips = [int(c) for c in num_clips]            # per video
    self.frames_per_clip = [int(f) for f in frames_per_clip]  # per clip, flat
    self.frames_per_video = [int(f) for f in frames_per_video]
    self.words_per_sent = [int(w) for w in words_per_sent]   # per sentence, flat

  @property
  def n_videos(self):
    return len(self.num_clips)

  def totals(self):
    words_par = sum(self.words_per_sent)
    return dict(videos=self.n_videos, clips=len(self.frames_per_clip),
                clip_frame_steps=sum(self.frames_per_clip),
                video_frame_steps=sum(self.frames_per_video),
                word_steps=sum(self.words_per_sent) + words_par)


def anet_like_spec(n_videos, seed=0, dataset='anet'):
  """ActivityNet/DiDeMo-shaped ragged sizes."""
  rng = np.random.RandomState(seed)
  if dataset == 'didemo':
    nclips = _sample_hist(rng, DIDEMO_CLIP_HIST, n_videos)
    total = int(nclips.sum())
    fpc = np.full(total, 80, dtype=np.int64)
    wps = _sample_quantiles(rng, DIDEMO_WORD_Q, total)
  else:
    nclips = _sample_hist(rng, ANET_CLIP_HIST, n_videos)
    total = int(nclips.sum())
    capped = rng.uniform(size=total) < 0.53
    fpc = np.where(capped, 80, rng.randint(1, 80, size=total))
    wps = _sample_quantiles(rng, ANET_WORD_Q, total)
  fpv = np.full(n_videos, 80, dtype=np.int64)
  return SplitSpec(nclips, fpc, fpv, wps)


def uniform_spec(n_videos, clips=4, frames=10, words=12):
  """BASELINE config 0: every video `clips` clips x `frames` frames, `words` words/sentence;
  the whole-video stream is the concatenation of its clips."""
  return SplitSpec([clips] * n_videos, [frames] * (clips * n_videos),
                   [clips * frames] * n_videos, [words] * (clips * n_videos))


def ragged_spec(n_videos, seed=0, max_clips=5, max_frames=9, max_words=7, max_video=12):
  """Small fully-ragged split used by the parity tests (includes length-1 items)."""
  rng = np.random.RandomState(seed)
  nclips = rng.randint(1, max_clips + 1, size=n_videos)
  total = int(nclips.sum())
  fpc = rng.randint(1, max_frames + 1, size=total)
  wps = rng.randint(1, max_words + 1, size=total)
  fpv = rng.randint(1, max_video + 1, size=n_videos)
  return SplitSpec(nclips, fpc, fpv, wps)


def batch_lengths(spec, batch_size):
  """Per loader batch: (lengths_clip, lengths_video, lengths_cap, lengths_paragraph) as the
  collate_fn would report them — the sizes alone, without materialising any feature."""
  out, clip_pos = [], 0
  for b0 in range(0, spec.n_videos, batch_size):
    b1 = min(spec.n_videos, b0 + batch_size)
    nclips = spec.num_clips[b0:b1]
    sumC = sum(nclips)
    fpc = spec.frames_per_clip[clip_pos:clip_pos + sumC]
    wps = spec.words_per_sent[clip_pos:clip_pos + sumC]
    clip_pos += sumC
    par, j = [], 0
    for c in nclips:
      par.append(sum(wps[j:j + c]))
      j += c
    out.append((np.asarray(fpc), np.asarray(spec.frames_per_video[b0:b1]), np.asarray(wps),
                np.asarray(par)))
  return out


def make_batches(spec, batch_size, img_dim, vocab_size, seed=0, feat='normal',
                 device='cpu', dtype=torch.float32):
  """Materialise `spec` as a list of 12-tuples (one per loader batch).

  feat: 'normal' -> N(0,1) (C3D-like); 'relu' -> |N(0,0.5)| (ICEP-like).
  Token ids are drawn from U{4..V-1} (ids 0-3 are PAD/<start>/<end>/UNK in anet_vocab.py:67-70).
  """
  gen = torch.Generator(device='cpu')
  gen.manual_seed(seed)
  batches = []
  clip_pos = 0
  n = spec.n_videos
  for b0 in range(0, n, batch_size):
    b1 = min(n, b0 + batch_size)
    B = b1 - b0
    nclips = spec.num_clips[b0:b1]
    sumC = sum(nclips)
    fpc = spec.frames_per_clip[clip_pos:clip_pos + sumC]
    wps = spec.words_per_sent[clip_pos:clip_pos + sumC]
    fpv = spec.frames_per_video[b0:b1]
    clip_pos += sumC

    Tc, Lc, Tv = max(fpc), max(wps), max(fpv)
    clips = torch.zeros(sumC, Tc, img_dim)
    captions = torch.zeros(sumC, Lc, dtype=torch.int64)
    for i in range(sumC):
      x = torch.randn(fpc[i], img_dim, generator=gen)
      if feat == 'relu':
        x = (0.5 * x).abs()
      clips[i, :fpc[i]] = x
      captions[i, :wps[i]] = torch.randint(4, vocab_size, (wps[i],), generator=gen)
    videos = torch.zeros(B, Tv, img_dim)
    par_len = []
    j = 0
    for v in range(B):
      x = torch.randn(fpv[v], img_dim, generator=gen)
      if feat == 'relu':
        x = (0.5 * x).abs()
      videos[v, :fpv[v]] = x
      par_len.append(sum(wps[j:j + nclips[v]]))
      j += nclips[v]
    Lp = max(par_len)
    paragraphs = torch.zeros(B, Lp, dtype=torch.int64)
    j = 0
    for v in range(B):
      toks = torch.cat([captions[j + c, :wps[j + c]] for c in range(nclips[v])])
      paragraphs[v, :par_len[v]] = toks       # activity_net/data.py:78
      j += nclips[v]

    batch = (clips.to(device=device, dtype=dtype), captions.to(device),
             videos.to(device=device, dtype=dtype), paragraphs.to(device),
             torch.tensor(fpc, dtype=torch.int64), torch.tensor(wps, dtype=torch.int64),
             torch.tensor(fpv, dtype=torch.int64), torch.tensor(par_len, dtype=torch.int64),
             tuple(nclips), tuple(nclips), tuple(range(b0, b1)),
             tuple('v_%06d' % k for k in range(b0, b1)))
    batches.append(batch)
  return batches


class ListLoader(object):
  """Minimal object with the DataLoader surface `encode_data` uses (iteration + len)."""

  def __init__(self, batches):
    self.batches = list(batches)

  def __iter__(self):
    return iter(self.batches)

  def __len__(self):
    return len(self.batches)


def correlated_embeddings(n, dim=1024, sigma=3.0, seed=0):
  """Scoring-only inputs of SURVEY.md §8d S5: normalize(z + sigma*eps) pairs; with sigma=3,
  n=4917, dim=1024 this gives R@1 ~ 33 %, medr 4 (a non-trivial rank distribution)."""
  rng = np.random.RandomState(seed)
  z = rng.standard_normal((n, dim))
  a = z + sigma * rng.standard_normal((n, dim))
  b = z + sigma * rng.standard_normal((n, dim))
  a /= np.linalg.norm(a, axis=1, keepdims=True)
  b /= np.linalg.norm(b, axis=1, keepdims=True)
  return a.astype(np.float32), b.astype(np.float32)


def dataset_samples(seed, img_dim, n_videos, didemo=False):
  """Per-video samples in the shape Dataset.__getitem__ returns them (activity_net/data.py:97-109):
  ragged clips, float-typed token ids, a whole-video stream, counts; seeded."""
  rng = np.random.RandomState(seed)
  samples = []
  for i in range(n_videos):
    n = int(rng.randint(1, 5))
    lc = rng.randint(1, 9, size=n)
    lw = rng.randint(1, 7, size=n)
    clips = [torch.from_numpy(rng.standard_normal((int(l), img_dim)).astype(np.float32)) for l in lc]
    caps = [torch.Tensor(rng.randint(1, 50, size=int(l)).tolist()) for l in lw]   # float ids (:80)
    video = torch.from_numpy(rng.standard_normal((int(rng.randint(2, 11)), img_dim)).astype(np.float32))
    paragraph = torch.cat(caps, 0)
    last = torch.full((n,), i, dtype=torch.int64) if didemo else 'v_%03d' % i
    samples.append((clips, caps, video, paragraph, torch.Tensor(lc.tolist()).long(),
                    torch.Tensor(lw.tolist()).long(), n, n, 100 + i, last))
  return samples
