#!/usr/bin/env python3
"""Per-step cost of the small-batch GRU step kernels: one encoder call (Seq2Seq pooling, all
sequences of full length T) at S sequences, with the mid-size kernel on (default) and off
(tune.mid_max_seqs=0 -> tiny / tiled kernels), arms interleaved (tools/_arms.py)."""
import argparse
import os
import statistics
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from cmhse_amd import layers  # noqa: E402
sys.path.insert(0, os.path.join(REPO, 'tools'))
import _arms  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--T', type=int, default=40)
  ap.add_argument('--H', type=int, default=1024)
  ap.add_argument('--sizes', default='1,8,32,64,152,320,512,1024,1536,2048')
  ap.add_argument('--dims', default='500,300,1024')
  ap.add_argument('--arms', default='tune.mid_max_seqs=0;tune.mid_max_seqs=4096')
  args = ap.parse_args()
  dev = torch.device('cuda', 0)
  arms = _arms.parse(args.arms)
  print('%6s %6s ' % ('I', 'S') + ' '.join('%28s' % a for a in args.arms.split(';')) + '   (us per step)')
  for I in [int(x) for x in args.dims.split(',')]:
    torch.manual_seed(0)
    layer = layers.Seq2Seq(I, args.H).to(dev)
    for S in [int(x) for x in args.sizes.split(',')]:
      x = torch.randn(S, args.T, I, device=dev)
      lens = torch.full((S,), args.T, dtype=torch.int64)
      res = [[] for _ in arms]
      outs = []
      for rnd in range(4):
        for i, a in enumerate(arms):
          _arms.apply(a)
          with torch.no_grad():
            layer(x, lens)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
              y = layer(x, lens)
            e1.record()
            torch.cuda.synchronize()
          if rnd == 0:
            outs.append(y)
          else:
            res[i].append(e0.elapsed_time(e1) / 3 / args.T * 1e3)
      err = max(float((o - outs[0]).abs().max()) for o in outs)
      print('%6d %6d ' % (I, S) + ' '.join('%28.1f' % statistics.median(r) for r in res) +
            '   max|diff| %.2g' % err)


if __name__ == '__main__':
  main()
