"""Timing-only experiment: per-workgroup phase timestamps of gru_step_kernel on the MI355X.

Builds a SEPARATE library with -DTILE_TRACE_BUILD (cmhse_amd/libcmhse_trace.so, never loaded by the
product path), runs two time steps of a level-1-sized batch, and summarises where a tile's wall
time goes (s_memrealtime, 10 ns ticks) and how the workgroups co-resident on one CU overlap.

  python tools/tile_trace.py [S] [I] [H] > gpurun_out/tile_trace.txt
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sim_trace(lib, dev, N, D=1024):
  """--sim [N]: the counting pass of the ranking kernel (sim_kernel<Rank>, 128 x 128 tiles, K = D) on
  synthetic.correlated_embeddings(N, D): where a tile's time goes, the clock it runs at, how many
  workgroups share a CU — the explanation of roofline_sim.frac."""
  import torch
  from cmhse_amd import ops, synthetic
  lib.cmhse_debug_set_sim_trace.restype = ctypes.c_int
  lib.cmhse_debug_set_sim_trace.argtypes = [ctypes.c_void_p]
  a, b = synthetic.correlated_embeddings(N, D, 3.0, seed=0)
  ad, bd = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
  n_tiles = (N + 127) // 128
  n_wg = 8 * ((n_tiles * n_tiles * 2 + 7) // 8) + 64       # the grouped deal pads the grid
  trace = torch.zeros(n_wg * 8, dtype=torch.int64, device=dev)
  for it in range(4):
    if it == 3:
      assert lib.cmhse_debug_set_sim_trace(trace.data_ptr()) == 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    ops.sim_rank(ad, bd)
    ev1.record()
    torch.cuda.synchronize()
  lib.cmhse_debug_set_sim_trace(None)
  tr = trace.cpu().numpy().reshape(n_wg, 8)
  tr = tr[tr[:, 4] != 0]
  t = tr[:, [0, 1, 3, 4]].astype(np.float64) * 0.01
  t -= t[:, 0].min()
  clk = (tr[:, 7] - tr[:, 5]) / ((tr[:, 3] - tr[:, 1]) * 10.0)
  hw, xcc = tr[:, 6] & 0xffffffff, (tr[:, 6] >> 32) & 0xf
  cu_key = xcc * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 12) & 0x1) * 16 + ((hw >> 8) & 0xf)
  keys = np.unique(cu_key)
  span = t[:, 3].max()
  flop_tile = 2.0 * 128 * 128 * D
  mfma_us = flop_tile / (157.3e12 / 256) * 1e6
  print('sim_kernel<Rank>  N=%d D=%d: %d tiles that ran, %d distinct CUs, call (diag + counting pass + finalize) %.1f us by events'
        % (N, D, len(tr), len(keys), ev0.elapsed_time(ev1) * 1e3))
  print('counting-pass span %.1f us; algorithmic %.1f GFLOP -> %.1f TFLOP/s over the span' %
        (span, 2.0 * N * N * D / 1e9, 2.0 * N * N * D / span / 1e6))
  for nme, d in [('first instr -> K loop', t[:, 1] - t[:, 0]), ('K loop (D = %d)' % D, t[:, 2] - t[:, 1]),
                 ('epilogue (count / arg-max / atomics)', t[:, 3] - t[:, 2]), ('whole tile', t[:, 3] - t[:, 0])]:
    print('%-38s mean %7.2f us   p10 %7.2f   p50 %7.2f   p90 %7.2f' % (nme, d.mean(), *np.percentile(d, [10, 50, 90])))
  print('MFMA-only time of one 128 x 128 x %d tile on one CU at peak: %.2f us' % (D, mfma_us))
  print('in-kernel shader clock over the K loops: median %.3f GHz (p10 %.3f, p90 %.3f)'
        % (np.median(clk), *np.percentile(clk, [10, 90])))
  per_cu = np.array([np.sum(cu_key == k) for k in keys])
  print('tiles per CU: min %d mean %.2f max %d (a perfectly even deal: %.2f)' %
        (per_cu.min(), per_cu.mean(), per_cu.max(), len(tr) / 256.0))
  grid = np.linspace(0, span, 2000)
  resident = np.zeros_like(grid)
  inloop = np.zeros_like(grid)
  for a_, b_, c_, d_ in t:
    resident += (grid >= a_) & (grid < d_)
    inloop += (grid >= b_) & (grid < c_)
  print('workgroups resident per CU over the span: mean %.2f; inside their K loops: mean %.2f' %
        (resident.mean() / len(keys), inloop.mean() / len(keys)))
  for frac in (0.1, 0.3, 0.5, 0.7, 0.9, 0.97):
    i = int(frac * (len(grid) - 1))
    print('  at %3.0f %% of the span: resident/CU %.2f  in-loop/CU %.2f' % (100 * frac, resident[i] / len(keys), inloop[i] / len(keys)))
  last_end = np.array([t[cu_key == k, 3].max() for k in keys])
  print('per-CU last end: min %.1f mean %.1f max %.1f us' % (last_end.min(), last_end.mean(), last_end.max()))
  loop = t[:, 2] - t[:, 1]
  print('share of a tile\'s time outside its K loop: %.1f %%; K loop at the MFMA rate of the measured clock would take %.1f us'
        % (100.0 * (1.0 - loop.sum() / (t[:, 3] - t[:, 0]).sum()), mfma_us * 2.4 / np.median(clk)))


def main():
  argv = [a for a in sys.argv[1:] if not a.startswith('--')]
  S = int(argv[0]) if len(argv) > 0 else 22419
  I = int(argv[1]) if len(argv) > 1 else 500
  H = int(argv[2]) if len(argv) > 2 else 1024
  csrc = os.path.join(ROOT, 'cmhse_amd', 'csrc')
  lib_path = os.path.join(ROOT, 'cmhse_amd', os.environ.get('TRACE_LIB', 'libcmhse_trace.so'))
  cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
         '-DTILE_TRACE_BUILD'] + os.environ.get('TRACE_FLAGS', '').split() + ['-o', lib_path] + [os.path.join(csrc, f) for f in
                                              ('gru.hip', 'sim.hip', 'bwd.hip')]
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
  _lib._lib = lib   # route ops through the trace build for this process only
  dev = torch.device('cuda', 0)
  if '--sim' in sys.argv:
    return sim_trace(lib, dev, int(argv[0]) if argv else 4917)
  T = 2
  ops.set_math_mode(os.environ.get('TRACE_MATH', 'fp32'))     # bf16x3: the fast mode's tiled step kernel
  # per-step launches by default: the stamps are indexed by workgroup, and the last step's overwrite the first's
  ops.tune('chain_min_steps', int(os.environ.get('TRACE_CHAIN_MIN_STEPS', '0')))
  x = torch.randn(S, T, I, device=dev)
  scale = float(os.environ.get('TRACE_DATA_SCALE', '1'))   # 0: all-zero operands (clock ceiling check)
  x *= scale
  lens = np.full(S, T, dtype=np.int64)
  g = torch.Generator(device='cpu').manual_seed(0)
  w = dict(w_ih=torch.randn(3 * H, I, generator=g).mul_(0.05).to(dev),
           w_hh=torch.randn(3 * H, H, generator=g).mul_(0.05).to(dev),
           b_ih=torch.zeros(3 * H, device=dev), b_hh=torch.zeros(3 * H, device=dev),
           w_lin=torch.randn(H, H, generator=g).mul_(0.05).to(dev), b_lin=torch.zeros(H, device=dev),
           w_att=torch.randn(1, H, generator=g).mul_(0.05).to(dev))
  pool = getattr(ops, os.environ.get('TRACE_POOL', 'POOL_ATTN'))
  if scale != 1.0:
    w = {k: v * scale for k, v in w.items()}
  n_wg = ((S + 63) // 64) * ((H + 63) // 64)
  trace = torch.zeros((2 * n_wg + 64) * 8, dtype=torch.int64, device=dev)
  ptrs = ops.padded_row_ptrs(x)
  reps = int(os.environ.get('TRACE_REPS', '3'))
  for it in range(reps):
    if it == reps - 1:
      assert lib.cmhse_debug_set_trace(trace.data_ptr()) == 0
    ops.gru_pool_fwd(w, pool, lens, I, H, dev, x_ptrs=ptrs)
    torch.cuda.synchronize()
  lib.cmhse_debug_set_trace(None)
  tr = trace.cpu().numpy().reshape(-1, 8)
  tr = tr[tr[:, 4] != 0]          # 128-row tiles launch half as many workgroups
  n_wg = len(tr)
  # unset: the launcher picks 128-row tiles for launches of >= 2048 64-row workgroups (gru_msub_for)
  thr = ops.tune('tall_tile_min_wgs')
  rows_per_tile = 128 if (thr > 0 and n_wg >= thr) else 64
  t = tr[:, :5].astype(np.float64) * 0.01   # us (100 MHz)
  t0 = t[:, 0].min()
  t -= t0
  hw, xcc = tr[:, 6] & 0xffffffff, (tr[:, 6] >> 32) & 0xf
  clk = (tr[:, 7] - tr[:, 5]) / ((tr[:, 3] - tr[:, 1]) * 10.0)   # shader cycles per ns = GHz
  cu = (hw >> 8) & 0xf
  sh = (hw >> 12) & 0x1
  se = (hw >> 13) & 0x7
  cu_key = xcc * 1000 + se * 100 + sh * 16 + cu
  names = ['first instr -> loop', 'first instr -> mark0', 'loop (x+h)', 'epilogue + drain']
  print('S=%d I=%d H=%d  workgroups=%d  distinct CUs=%d' % (S, I, H, n_wg, len(np.unique(cu_key))))
  span = t[:, 4].max()
  print('launch span %.1f us' % span)
  dur = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 0], t[:, 3] - t[:, 1], t[:, 4] - t[:, 3]], axis=1)
  for i, nme in enumerate(names):
    print('%-24s mean %8.2f us   p10 %8.2f   p50 %8.2f   p90 %8.2f' %
          (nme, dur[:, i].mean(), *np.percentile(dur[:, i], [10, 50, 90])))
  tot = t[:, 4] - t[:, 0]
  print('%-24s mean %8.2f us   p10 %8.2f   p50 %8.2f   p90 %8.2f' %
        ('whole tile', tot.mean(), *np.percentile(tot, [10, 50, 90])))
  flop_tile = 2.0 * rows_per_tile * 192 * (I + H)
  print('MFMA-only time of one tile on one CU at peak: %.2f us' % (flop_tile / (157.3e12 / 256) * 1e6))
  # concurrency per CU over time: how many workgroups are inside their K loops
  keys = np.unique(cu_key)
  grid = np.linspace(0, span, 4000)
  inloop = np.zeros_like(grid)
  resident = np.zeros_like(grid)
  for k in keys:
    sel = cu_key == k
    for a, b, c, d in zip(t[sel, 0], t[sel, 1], t[sel, 3], t[sel, 4]):
      inloop += (grid >= b) & (grid < c)
      resident += (grid >= a) & (grid < d)
  print('avg workgroups resident per CU %.2f, inside the K loops %.2f' %
        (resident.mean() / len(keys), inloop.mean() / len(keys)))
  for frac in (0.1, 0.3, 0.5, 0.7, 0.9, 0.97):
    i = int(frac * (len(grid) - 1))
    print('  at %3.0f %% of the launch: resident/CU %.2f  in-loop/CU %.2f' %
          (100 * frac, resident[i] / len(keys), inloop[i] / len(keys)))
  print('in-kernel shader clock over the K loops: median %.3f GHz (p10 %.3f, p90 %.3f); first 768 '
        'workgroups %.3f GHz, last 768 %.3f GHz' % (np.median(clk), *np.percentile(clk, [10, 90]),
                                                   np.median(clk[:768]), np.median(clk[-768:])))
  # loop duration against the number of workgroups inside their loops on the same CU
  conc, loopd = [], []
  for k in keys:
    sel = np.where(cu_key == k)[0]
    b, c = t[sel, 1], t[sel, 3]
    for i in range(len(sel)):
      ov = np.clip(np.minimum(c, c[i]) - np.maximum(b, b[i]), 0, None).sum()
      conc.append(ov / (c[i] - b[i]))
      loopd.append(c[i] - b[i])
  conc, loopd = np.array(conc), np.array(loopd)
  mfma_us = flop_tile / (157.3e12 / 256) * 1e6
  for lo, hi in ((1.0, 1.5), (1.5, 2.0), (2.0, 2.4), (2.4, 2.7), (2.7, 2.9), (2.9, 3.01)):
    m = (conc >= lo) & (conc < hi)
    if m.sum():
      print('  in-loop concurrency %.1f-%.1f: %5d tiles, loop %.1f us, implied MFMA busy %.0f %%' %
            (lo, hi, m.sum(), loopd[m].mean(), 100 * conc[m].mean() * mfma_us / loopd[m].mean()))
  # one CU's timeline
  k = keys[len(keys) // 2]
  sel = np.where(cu_key == k)[0]
  sel = sel[np.argsort(t[sel, 0])]
  print('timeline of CU %d (%d workgroups): first instr, loop start, mark0, loop end, drained [us]' % (k, len(sel)))
  for i in sel[:24]:
    print('   wg %6d  %8.1f %8.1f %8.1f %8.1f %8.1f' % ((i,) + tuple(t[i])))
  per_cu = np.array([np.sum(cu_key == k) for k in keys])
  print('workgroups per CU: min %d mean %.1f max %d' % (per_cu.min(), per_cu.mean(), per_cu.max()))
  last_end = np.array([t[cu_key == k, 4].max() for k in keys])
  print('per-CU last end: min %.1f mean %.1f max %.1f us (idle tail = span - end)' %
        (last_end.min(), last_end.mean(), last_end.max()))


if __name__ == '__main__':
  main()
