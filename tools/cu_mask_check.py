#!/usr/bin/env python3
"""The multi-step kernels on a chip that gives them fewer CUs than it reports (run this under
HSA_CU_MASK=0:0-31: the queue's workgroups are confined to 32 CUs while the device still reports 256 —
what another tenant or a partition would do to the assumptions of csrc/grid_sync.hpp).

  1. the step chain (one workgroup per task, started in index order, no co-residency requirement) must
     stay bit-identical to per-step launches and record no timeout;
  2. the resident tail kernels of a training step (H / 16 = 64 workgroups that must ALL be on the chip,
     one per CU) cannot fit: the library must surface CMHSE_ERR_TIMEOUT (never return wrong gradients
     silently), and after the caller acknowledges it the same step must run on per-step launches and
     give the reference gradients.
Prints one line per check and exits 0 when all hold.
"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cmhse_amd import layers, ops  # noqa: E402


def main():
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(0)
  ok = True
  # 1. inference chain
  g = torch.Generator().manual_seed(4)
  I, H, S, T = 96, 1024, 2500, 8
  w = {k: v.to(dev) for k, v in dict(
      w_ih=torch.randn(3 * H, I, generator=g).mul_(0.1), w_hh=torch.randn(3 * H, H, generator=g).mul_(0.05),
      b_ih=torch.zeros(3 * H), b_hh=torch.zeros(3 * H)).items()}
  x = torch.randn(S, T, I, generator=g).to(dev)
  lens = np.random.RandomState(1).randint(1, T + 1, size=S)
  lens[:1500] = T
  req = dict(weights=w, pool_mode=ops.POOL_MAX, lens=lens, I=I, H=H, device=dev, x_ptrs=ops.padded_row_ptrs(x))
  with ops.tuned(tiny_max_seqs=0, mid_max_seqs=0):
    with ops.tuned(chain_min_steps=0):
      ref, _ = ops.gru_pool_fwd(**req)
      ref = ref.clone()
    t0 = time.perf_counter()
    out, _ = ops.gru_pool_fwd(**req)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
  same = torch.equal(out, ref) and ops.async_status() == 0
  print('step chain under the mask: bit-identical %s, status %d, %.1f ms' % (same, ops.async_status(), ms))
  ok = ok and same
  # 2. training step with a few-sequence tail (the resident kernels)
  torch.manual_seed(2)
  layer = layers.Seq2Seq(64, 1024).to(dev)
  S2, T2 = 40, 30
  lens2 = np.full(S2, 6)
  lens2[:8] = T2                      # 8 sequences run 24 steps past the others: a resident tail
  x2 = torch.randn(S2, T2, 64, device=dev)

  def grads():
    layer.zero_grad()
    xt = x2.clone().requires_grad_(True)
    layer(xt, torch.from_numpy(lens2)).sum().backward()
    torch.cuda.synchronize()
    return [xt.grad.clone()] + [p.grad.clone() for p in layer.parameters()]

  with ops.tuned(fwd_tail_min_steps=0, bwd_tail_min_steps=0):
    want = grads()                    # per-step launches: the reference
  ops.tune('resident_timeout_ms', 300)
  timed_out = False
  try:
    got = grads()
    status = ops.async_status()
    if status != 0:
      timed_out = True
    else:
      same2 = all(torch.allclose(a, b, rtol=1e-5, atol=1e-6) for a, b in zip(want, got))
      print('resident tails fitted under the mask: gradients equal %s' % same2)
      ok = ok and same2
  except RuntimeError as e:
    timed_out = 'grid barrier' in str(e) or 'TIMEOUT' in str(e).upper()
    print('training step raised:', str(e)[:120])
  if timed_out:
    assert ops.async_status(clear=True) == -5
    fell_back = ops.tune('multi_step_off') == 1
    again = grads()
    same3 = all(torch.equal(a, b) for a, b in zip(want, again)) and ops.async_status() == 0
    print('resident tails did not fit: CMHSE_ERR_TIMEOUT surfaced, fallback to per-step launches %s, gradients after the '
          'acknowledgement equal the reference %s' % (fell_back, same3))
    ok = ok and fell_back and same3
  sys.exit(0 if ok else 1)


if __name__ == '__main__':
  main()
