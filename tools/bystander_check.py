#!/usr/bin/env python3
"""Does a call on one stream change a bit of a concurrent, unrelated call on another stream?
The victim: a complete attention-pooled encoder call (step chain + attention projection + pooling of 426
sequences, exact fp32) on the current stream.  The neighbour: another encoder's per-step launches on a
second stream, in every math mode the library has.  Written while tracking down why an (abandoned) bf16x6
mode changed a few pooled rows of its neighbours (profiles/r05_bf16x6_rate.txt; the cause, a gfx950 hazard
between double-rate matrix instructions and v_pk_fma_f32, and the far more sensitive form of this check:
profiles/r05_bf16_mfma_bystander.txt, tools/pkfma_canary.py); kept as an end-to-end regression check for
the modes that ship: every repetition must report 0 rows.

  python tools/bystander_check.py [--reps 100]
"""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cmhse_amd import ops  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--reps', type=int, default=100)
  args = ap.parse_args()
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(0)
  g = torch.Generator().manual_seed(8)

  def weights(I, H):
    w = dict(w_ih=torch.randn(3 * H, I, generator=g).mul_(0.05), w_hh=torch.randn(3 * H, H, generator=g).mul_(0.05),
             b_ih=torch.randn(3 * H, generator=g).mul_(0.1), b_hh=torch.randn(3 * H, generator=g).mul_(0.1),
             w_lin=torch.randn(H, H, generator=g).mul_(0.05), b_lin=torch.randn(H, generator=g).mul_(0.1),
             w_att=torch.randn(1, H, generator=g).mul_(0.2))
    return {k: v.to(dev) for k, v in w.items()}

  rng = np.random.RandomState(3)
  H = 1024
  SA, TA, IA = 426, 80, 2048
  SB, TB, IB = 96, 300, 300
  lensA = rng.randint(1, TA + 1, size=SA)
  lensA[:200] = TA
  xA = torch.randn(SA, TA, IA, generator=g).to(dev)
  xB = torch.randn(SB, TB, IB, generator=g).to(dev)
  rA = dict(weights=weights(IA, H), pool_mode=ops.POOL_ATTN, lens=lensA, I=IA, H=H, device=dev, x_ptrs=ops.padded_row_ptrs(xA))
  rB = dict(weights=weights(IB, H), pool_mode=ops.POOL_LAST, lens=np.full(SB, TB), I=IB, H=H, device=dev,
            x_ptrs=ops.padded_row_ptrs(xB))
  side = torch.cuda.Stream()
  bad_total = 0
  with ops.tuned(tiny_max_seqs=0, mid_max_seqs=0):
    base, _ = ops.gru_pool_fwd(**rA)
    base = base.clone()
    torch.cuda.synchronize()
    for mode in ('fp32', 'bf16x3'):
      bad_reps = bad_rows = 0
      for _ in range(args.reps):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
          ops.set_math_mode(mode)
          try:
            with ops.tuned(chain_min_steps=0):            # the neighbour: one launch per time step
              keep = ops.gru_pool_fwd(**rB)
          finally:
            ops.set_math_mode('fp32')
        out, _ = ops.gru_pool_fwd(**rA)                   # the victim, concurrently
        torch.cuda.synchronize()
        n = int(((out - base).abs().amax(1) > 0).sum())
        bad_reps += n > 0
        bad_rows += n
        del keep
      print('neighbour in %-7s: %d of %d repetitions changed the victim (%d rows in all)' % (mode, bad_reps, args.reps, bad_rows))
      bad_total += bad_rows
  sys.exit(1 if bad_total else 0)


if __name__ == '__main__':
  main()
