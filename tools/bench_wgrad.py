#!/usr/bin/env python3
"""Stand-alone rate of the weight-gradient products (gemm_tn_rows_kernel): one encoder call with
S sequences of ONE step, so its backward pass is a single BPTT launch plus one product over S packed
rows.  Run under `rocprofv3 --kernel-trace --stats` to read the kernel's own duration; the printed
figure is the whole backward call (events), which bounds it from above.

  python tools/bench_wgrad.py --S 9600 --I 2048 --H 1024 --pool seq2seq
"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from cmhse_amd import layers, ops  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--S', type=int, default=9600)
  ap.add_argument('--T', type=int, default=1)
  ap.add_argument('--I', type=int, default=2048)
  ap.add_argument('--H', type=int, default=1024)
  ap.add_argument('--pool', default='seq2seq', choices=['seq2seq', 'attention', 'maxout'])
  ap.add_argument('--reps', type=int, default=5)
  ap.add_argument('--tn_rows_bm', type=int, default=0, help='tile height of the products: 128, 192, 0 = by shape')
  args = ap.parse_args()
  dev = torch.device('cuda', 0)
  ops.tune('tn_rows_bm', args.tn_rows_bm)
  cls = {'seq2seq': layers.Seq2Seq, 'attention': layers.Attention, 'maxout': layers.Maxout}[args.pool]
  torch.manual_seed(0)
  layer = cls(args.I, args.H).to(dev)
  x = torch.randn(args.S, args.T, args.I, device=dev)
  lens = torch.full((args.S,), args.T, dtype=torch.int64)
  rows = args.S * args.T
  flops = rows * 2.0 * 3 * args.H * (args.I + args.H)
  for rep in range(args.reps + 1):
    out = layer(x, lens)
    g = torch.ones_like(out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out.backward(g)
    e1.record()
    torch.cuda.synchronize()
    if rep:
      ms = e0.elapsed_time(e1)
      print('S %d T %d I %d H %d %s: backward %.3f ms; dW_ih + dW_hh = %.1f GFLOP -> >= %.1f TFLOP/s'
            % (args.S, args.T, args.I, args.H, args.pool, ms, flops / 1e9, flops / ms / 1e9))
    layer.zero_grad()


if __name__ == '__main__':
  main()
