"""Workloads of bench.py (the configurations BASELINE.json names) and the synthetic loader batches they
run on, generated directly in HBM.  Shared by bench.py, its supplementary legs (bench_legs.py), the
tools and the full-size GPU tests."""
import argparse

import numpy as np
import torch

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (no sparsity)

WORKLOADS = {
    'anet_c3d_val': dict(n_videos=4917, batch=32, img_dim=500, feat='normal', vocab=13058,
                         dataset='anet'),
    'anet_icep_val': dict(n_videos=4917, batch=32, img_dim=2048, feat='relu', vocab=13058,
                          dataset='anet'),
    'didemo_icep_val': dict(n_videos=1004, batch=32, img_dim=2048, feat='relu', vocab=7205,
                            dataset='didemo'),
    'plumbing': dict(n_videos=64, batch=16, img_dim=500, feat='normal', vocab=13058,
                     dataset='uniform'),
}


def make_opt(wl, rnn_type, embed):
  return argparse.Namespace(
      margin=0.2, word_dim=300, embed_size=embed, grad_clip=0.0, learning_rate=0.001,
      max_violation=False, img_dim=wl['img_dim'], measure='cosine', rnn_type=rnn_type,
      img_first_size=embed, cap_first_size=embed, low_level_loss=False, weak_low_level_loss=False,
      reconstruct_loss=False, lowest_reconstruct_loss=False, norm=False,
      data_name='anet_precomp', vocab_size=wl['vocab'])


def device_batch(spec, b0, b1, clip_pos, img_dim, vocab, feat, gen, device):
  """One loader batch of the 12-tuple contract, generated directly in HBM."""
  nclips = spec.num_clips[b0:b1]
  sumC = sum(nclips)
  fpc = torch.tensor(spec.frames_per_clip[clip_pos:clip_pos + sumC], dtype=torch.int64)
  wps = torch.tensor(spec.words_per_sent[clip_pos:clip_pos + sumC], dtype=torch.int64)
  fpv = torch.tensor(spec.frames_per_video[b0:b1], dtype=torch.int64)
  B = b1 - b0

  def feats(lens):
    T = int(lens.max())
    x = torch.randn(len(lens), T, img_dim, generator=gen, device=device)
    if feat == 'relu':
      x = (0.5 * x).abs_()
    mask = torch.arange(T, device=device)[None, :] < lens.to(device)[:, None]
    return x * mask[:, :, None]

  clips, videos = feats(fpc), feats(fpv)
  Lc = int(wps.max())
  caps = torch.randint(4, vocab, (sumC, Lc), generator=gen, device=device)
  caps = caps * (torch.arange(Lc, device=device)[None, :] < wps.to(device)[:, None])
  starts = np.concatenate([[0], np.cumsum(nclips)])
  par_len = torch.tensor([int(wps[starts[v]:starts[v + 1]].sum()) for v in range(B)],
                         dtype=torch.int64)
  pars = torch.zeros(B, int(par_len.max()), dtype=torch.int64, device=device)
  caps_h, wps_l = caps.cpu(), wps.tolist()
  for v in range(B):
    toks = torch.cat([caps_h[j, :wps_l[j]] for j in range(starts[v], starts[v + 1])])
    pars[v, :len(toks)] = toks.to(device)
  return (clips, caps, videos, pars, fpc, wps, fpv, par_len, tuple(nclips), tuple(nclips),
          tuple(range(b0, b1)), tuple('v_%06d' % k for k in range(b0, b1)))


def build_loader(spec, wl, device, own_lo, own_hi=None, seed=0):
  """All loader batches of the split; only the batches this rank owns — [own_lo, own_hi), or the
  index collection `own_lo` when `own_hi` is None — are materialised (the others carry just
  num_clips, which is all parallel_eval needs from them)."""
  own = set(range(own_lo, own_hi)) if own_hi is not None else set(own_lo)
  gen = torch.Generator(device=device)
  batches, clip_pos = [], 0
  n, bs = spec.n_videos, wl['batch']
  for bi, b0 in enumerate(range(0, n, bs)):
    b1 = min(n, b0 + bs)
    nclips = spec.num_clips[b0:b1]
    if bi in own:
      gen.manual_seed(seed * 100003 + bi)
      batches.append(device_batch(spec, b0, b1, clip_pos, wl['img_dim'], wl['vocab'], wl['feat'],
                                  gen, device))
    else:
      stub = [None] * 12
      stub[8] = tuple(nclips)
      batches.append(tuple(stub))
    clip_pos += sum(nclips)
  return batches


def gru_flops_per_step(I, H):
  """algorithmic FLOPs of one GRU (sequence, timestep)."""
  return 2 * 3 * H * I + 2 * 3 * H * H + 14 * H

