"""Timing-only experiment: where does one mid-size GRU step (gru_step_mid_kernel) spend its time,
and how long is the gap between two dependent step launches?  Builds a separate -DTILE_TRACE_BUILD
library (never loaded by the product path) and stamps s_memrealtime (10 ns) per workgroup:
  0 entry   1 operand addresses ready   2 MFMA loop done   3 LDS combine + barrier done
  4 gate math done (epilogue loads returned)   5 stores drained

  python tools/mid_trace.py [S] [T] [I] [H]
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  argv = [a for a in sys.argv[1:] if not a.startswith('--')]
  S = int(argv[0]) if len(argv) > 0 else 152
  T = int(argv[1]) if len(argv) > 1 else 12
  I = int(argv[2]) if len(argv) > 2 else 500
  H = int(argv[3]) if len(argv) > 3 else 1024
  csrc = os.path.join(ROOT, 'cmhse_amd', 'csrc')
  lib_path = os.path.join(ROOT, 'cmhse_amd', 'libcmhse_trace.so')
  cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
         '-DTILE_TRACE_BUILD', '-o', lib_path] + [os.path.join(csrc, f) for f in ('gru.hip', 'sim.hip', 'bwd.hip')]
  srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)]
  stale = not os.path.exists(lib_path) or \
      os.path.getmtime(lib_path) < max(os.path.getmtime(p) for p in srcs)
  if '--build-only' in sys.argv or stale:
    subprocess.check_call(cmd)
  if '--build-only' in sys.argv:
    return
  import torch
  from cmhse_amd import _lib, ops
  lib = ctypes.CDLL(lib_path)
  for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
  lib.cmhse_debug_set_trace.restype = ctypes.c_int
  lib.cmhse_debug_set_trace.argtypes = [ctypes.c_void_p]
  _lib._lib = lib
  dev = torch.device('cuda', 0)
  x = torch.randn(S, T, I, device=dev)
  lens = np.full(S, T, dtype=np.int64)
  g = torch.Generator(device='cpu').manual_seed(0)
  w = dict(w_ih=torch.randn(3 * H, I, generator=g).mul_(0.05).to(dev),
           w_hh=torch.randn(3 * H, H, generator=g).mul_(0.05).to(dev),
           b_ih=torch.zeros(3 * H, device=dev), b_hh=torch.zeros(3 * H, device=dev))
  bm = 16 if S <= 16 else 32
  m_blocks = (S + bm - 1) // bm
  bu = ops.tune('mid_units')
  if bu not in (4, 8, 16):   # mid_units() of gru.hip: narrowest unit tile whose grid fits 256 CUs
    bu = next((b for b in (4, 8) if ((H + b - 1) // b) * m_blocks <= 256), 16)
  n_wg = m_blocks * ((H + bu - 1) // bu)
  trace = torch.zeros(T * n_wg * 8, dtype=torch.int64, device=dev)
  ptrs = ops.padded_row_ptrs(x)
  for it in range(3):
    if it == 2:
      assert lib.cmhse_debug_set_trace(trace.data_ptr()) == 0
    ops.gru_pool_fwd(w, ops.POOL_LAST, lens, I, H, dev, x_ptrs=ptrs)
    torch.cuda.synchronize()
  lib.cmhse_debug_set_trace(None)
  tr = trace.cpu().numpy().reshape(T, n_wg, 8).astype(np.float64) * 0.01   # us
  print('S=%d T=%d I=%d H=%d: %d workgroups (%d hidden units each) per step' % (S, T, I, H, n_wg, bu))
  # marks 0,1,2,3,5 (the epilogue operands are prefetched before the loop: no mark between the
  # gate math and the drained stores)
  names = ['entry -> addresses', 'MFMA loop (K = H)', 'LDS combine + barrier',
           'epilogue gates + stores drained']
  slots = [0, 1, 2, 3, 5]
  for t in range(1, T):      # step 0 has no h phase
    a = tr[t]
    start, end = a[:, 0].min(), a[:, 5].max()
    prev_end = tr[t - 1][:, 5].max()
    line = 'step %2d: gap after previous step %5.2f us | kernel span %6.2f us | ' % (
        t, start - prev_end, end - start)
    line += 'last wg entry +%.2f | ' % (a[:, 0].max() - start)
    d = [a[:, slots[i + 1]] - a[:, slots[i]] for i in range(4)]
    line += '  '.join('%s %.2f' % (n.split()[0], x.mean()) for n, x in zip(names, d))
    print(line)
  a = tr[T // 2]
  d = [a[:, slots[i + 1]] - a[:, slots[i]] for i in range(4)]
  print('\nmid step, per-workgroup phases [us]:')
  for n, x in zip(names, d):
    print('  %-32s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f' %
          ((n, x.mean()) + tuple(np.percentile(x, [10, 50, 90])) + (x.max(),)))
  tot = a[:, 5] - a[:, 0]
  print('  %-32s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f' %
        (('whole workgroup', tot.mean()) + tuple(np.percentile(tot, [10, 50, 90])) + (tot.max(),)))
  print('  step = first entry -> last drain: %.2f us' % (a[:, 5].max() - a[:, 0].min()))


if __name__ == '__main__':
  main()
