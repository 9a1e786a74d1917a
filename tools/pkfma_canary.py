#!/usr/bin/env python3
"""Lost-update canary beside the GRU step kernels (profiles/r05_bf16x6_rate.txt, "bystander").

tools/microbench/canary.hip::pkfma_canary_kernel repeats attn_pool_kernel's inner loop on data whose
sums are exact and re-derives them in integer arithmetic.  It runs on the caller's stream while a
neighbour encoder call (96 sequences x 300 steps, every step forced onto the LDS-tiled kernel, one
launch per step) runs on a side stream in the math mode under test.  Prints, per mode, how many
(thread, component) sums came out wrong and what the first of them look like.

  python tools/pkfma_canary.py [--modes fp32,bf16x3] [--reps 10] [--neighbour steps|chain|mid] [--flag6 0x800]
(builds tools/microbench/canary.so itself.)  --flag6: the mode_flags bit of a library that still carries the
withdrawn bf16x6 tile (CMHSE_HIP_LIB=...).  What it found: profiles/r05_bf16_mfma_bystander.txt.
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from cmhse_amd import _lib, ops  # noqa: E402


sys.path.insert(0, os.path.join(R, 'tools'))
from canary_build import CANARY_LIB, CANARY_SRC, build_canary  # noqa: E402,F401


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--modes', default='none,fp32,bf16x3')
  ap.add_argument('--reps', type=int, default=10)
  ap.add_argument('--blocks', type=int, default=4096)
  ap.add_argument('--iters', type=int, default=24)
  ap.add_argument('--len', type=int, default=80)
  ap.add_argument('--flag6', type=lambda x: int(x, 0), default=0)
  ap.add_argument('--neighbour', default='steps', choices=['steps', 'chain', 'mid'],
                  help='steps: 96 x 300, one LDS-tiled launch per step; chain: 2048 x 80 in step chains (the validation pass); mid: 152 x 80 on the small-batch kernels (a training batch)')
  args = ap.parse_args()
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(0)
  can = ctypes.CDLL(build_canary())
  rows = 24576
  hs = torch.empty(rows, 1024, dtype=torch.float32, device=dev)
  main_s = torch.cuda.current_stream()
  assert can.pkfma_canary_fill(ctypes.c_void_p(hs.data_ptr()), ctypes.c_uint32(rows), ctypes.c_void_p(main_s.cuda_stream)) == 0
  torch.cuda.synchronize()

  g = torch.Generator().manual_seed(8)
  H, SB, TB, IB = 1024, 96, 300, 300
  w = dict(w_ih=torch.randn(3 * H, IB, generator=g).mul_(0.05), w_hh=torch.randn(3 * H, H, generator=g).mul_(0.05),
           b_ih=torch.randn(3 * H, generator=g).mul_(0.1), b_hh=torch.randn(3 * H, generator=g).mul_(0.1),
           w_lin=torch.randn(H, H, generator=g).mul_(0.05), b_lin=torch.randn(H, generator=g).mul_(0.1),
           w_att=torch.randn(1, H, generator=g).mul_(0.2))
  w = {k: v.to(dev) for k, v in w.items()}
  xB = torch.randn(SB, TB, IB, generator=g).to(dev)
  rB = dict(weights=w, pool_mode=ops.POOL_LAST, lens=np.full(SB, TB), I=IB, H=H, device=dev, x_ptrs=ops.padded_row_ptrs(xB))
  side = torch.cuda.Stream()
  if args.neighbour == 'steps':
    ops.tune('tiny_max_seqs', 0)
    ops.tune('mid_max_seqs', 0)
    ops.tune('chain_min_steps', 0)
  else:
    SB, TB, IB = (2048, 80, 2048) if args.neighbour == 'chain' else (152, 80, 2048)
    w['w_ih'] = torch.randn(3 * H, IB, generator=g).mul_(0.02).to(dev)
    xB = torch.randn(SB, TB, IB, generator=g).to(dev)
    lens = np.sort(np.random.RandomState(5).randint(TB // 2, TB + 1, size=SB))[::-1].copy()
    rB = dict(weights=w, pool_mode=ops.POOL_ATTN, lens=lens, I=IB, H=H, device=dev, x_ptrs=ops.padded_row_ptrs(xB))

  orig = ops._prepare_fwd

  def prep6(*a, **k):
    job, meta = orig(*a, **k)
    job['mode_flags'] |= args.flag6
    lib = _lib.load()
    b = job['b']
    nbytes = lib.cmhse_gru_pool_workspace(b.S, b.Tmax, job['ctx']['sched'].sum_T, b.I, b.H, job['mode_flags'])
    job['ws'] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    job['ws_bytes'] = nbytes
    job['ctx']['ws'] = job['ws']
    return job, meta

  print('library: %s' % _lib.LIB_PATH)
  for mode in [m for m in args.modes.split(',') if m]:
    report = torch.zeros(8 + 8 * 14, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for rep in range(args.reps):
      if mode != 'none':
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
          if mode == 'bf16x6':
            assert args.flag6, '--flag6 and a library that has the mode'
            ops._prepare_fwd = prep6
          else:
            ops.set_math_mode(mode)
          try:
            keep = ops.gru_pool_fwd(**rB)
          finally:
            ops._prepare_fwd = orig
            ops.set_math_mode('fp32')
      assert can.pkfma_canary_launch(ctypes.c_void_p(report.data_ptr()), ctypes.c_void_p(hs.data_ptr()), rows, args.len,
                                     args.blocks, args.iters, ctypes.c_void_p(main_s.cuda_stream)) == 0
      torch.cuda.synchronize()
    r = report.cpu().numpy().astype(np.int64) & 0xffffffff
    sums = int(r[0]) * 256 * 4
    print('neighbour %-7s: %d wrong sums of %d (%d workgroup-iterations of %d steps)' % (mode, int(r[1]), sums, int(r[0]), args.len))
    for i in range(min(int(r[2]), 14)):
      q = r[8 + 8 * i: 16 + 8 * i]
      got = float(np.array([q[4]], dtype=np.uint32).view(np.float32)[0])
      print('   block %d iteration %d thread %d (wave %d lane %d) component %d: got %.1f want %d (deficit %.1f)'
            % (q[0], q[1], q[2], q[2] // 64, q[2] % 64, q[3], got, q[5], q[5] - got))


if __name__ == '__main__':
  main()
