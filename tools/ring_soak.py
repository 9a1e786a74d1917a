#!/usr/bin/env python3
"""Full-size race screen of the bf16x3 LDS-DMA ring (nt_phase_bf3_ring): the whole validation split encoded
`--passes` times in bf16x3 mode with an exact-fp32 pass in between every `--every` passes; every pass's six
embedding matrices must equal the first pass's bit for bit.  A fragment read placed before the DMA that
fills its stage has landed would not be deterministic: it shows up as a pass that differs.

  python tools/ring_soak.py [--passes 40] [--every 5] [--n_videos 0]
"""
import argparse
import os
import sys
import zlib

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

import bench  # noqa: E402
from cmhse_amd import ops, synthetic  # noqa: E402
from cmhse_amd.evaluation import encode_data_device  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402

KEYS = ['vid_emb', 'para_emb', 'clip_emb', 'cap_emb', 'vid_ctx', 'para_ctx']


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--passes', type=int, default=40)
  ap.add_argument('--every', type=int, default=5)
  ap.add_argument('--n_videos', type=int, default=0)
  ap.add_argument('--rnn_type', default='attention')
  args = ap.parse_args()
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(0)
  wl = dict(bench.WORKLOADS['anet_icep_val'])
  if args.n_videos:
    wl['n_videos'] = args.n_videos
  opt = bench.make_opt(wl, args.rnn_type, 1024)
  torch.manual_seed(1)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(wl['n_videos'], seed=0, dataset=wl['dataset'])
  nb = (spec.n_videos + wl['batch'] - 1) // wl['batch']
  batches = bench.build_loader(spec, wl, dev, 0, nb)
  quiet = lambda *a, **k: None

  def one(mode):
    ops.set_math_mode(mode)
    try:
      cat, _, _ = encode_data_device(opt, model, batches, logging=quiet)
    finally:
      ops.set_math_mode('fp32')
    return cat

  first = {k: v.clone() for k, v in one('bf16x3').items() if k in KEYS}
  exact = one('fp32')
  dev_max = max(float((first[k] - exact[k]).abs().max()) for k in KEYS)
  bad = 0
  for i in range(1, args.passes):
    if i % args.every == 0:
      one('fp32')
    cat = one('bf16x3')
    same = all(torch.equal(cat[k], first[k]) for k in KEYS)
    bad += 0 if same else 1
    if not same:
      print('pass %d differs from the first: %s' % (i, [k for k in KEYS if not torch.equal(cat[k], first[k])]))
  crc = 0
  for k in KEYS:
    crc = zlib.crc32(first[k].cpu().numpy().tobytes(), crc)
  print('%d videos, %s pooling: %d bf16x3 passes, %d differ from the first; crc32 of the six matrices %d; '
        'max |bf16x3 - fp32| %.3g' % (wl['n_videos'], args.rnn_type, args.passes, bad, crc, dev_max))
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
