#!/usr/bin/env python3
"""Timeline of ONE validation pass from a rocprofv3 --kernel-trace CSV of bench.py.

Usage: python tools/trace_timeline.py <kernel_trace.csv> [pass_index_from_end=0]

Splits the trace into passes at the sim_kernel<1> (counting pass) launches — each pass ends with
two of them (i2t, t2i) — and prints for the chosen pass: the phases (runs of one kernel name) with
wall span / summed kernel time / launches, the idle gaps, and for the small-batch step kernels a
table of launch duration against grid size.
"""
import csv
import sys
from collections import OrderedDict


def short(name):
  for key in ['gru_step_tiny_kernel', 'gru_step_mid_kernel', 'gru_step_chain_kernel', 'gru_step_kernel', 'attn_energy_kernel',
              'attn_pool_kernel', 'split_rows_kernel', 'split_bf16x3_kernel', 'sim_kernel<1', 'sim_kernel<0', 'sim_kernel<2', 'l2norm_rows',
              'contrastive', 'xproj_kernel', 'top1_finalize', 'copyBuffer', 'fillBuffer']:
    if key in name:
      return key
  return 'other:' + name[:40]


def main():
  argv = [a for a in sys.argv[1:] if not a.startswith('--')]
  path = argv[0]
  back = int(argv[1]) if len(argv) > 1 else 0
  rows = []
  for r in csv.DictReader(open(path)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']),
                 int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Stream_Id'])))
  rows.sort()
  ends = [i for i, r in enumerate(rows) if r[2] == 'sim_kernel<1']
  pass_ends = ends[1::2]
  hi = pass_ends[len(pass_ends) - 1 - back]
  lo = pass_ends[len(pass_ends) - 2 - back] + 1 if len(pass_ends) - 2 - back >= 0 else 0
  # skip the trailing finalize of the previous pass
  while rows[lo][2] in ('top1_finalize',):
    lo += 1
  sel = rows[lo:hi + 1]
  t0 = sel[0][0]
  print('pass: %d launches, wall %.2f ms' % (len(sel), (sel[-1][1] - t0) / 1e6))
  # phases = maximal runs of the same kernel name (in start order)
  phases = []
  for s, e, n, g, st in sel:
    if phases and phases[-1]['name'] == n:
      p = phases[-1]
      p['end'] = max(p['end'], e)
      p['busy'] += e - s
      p['n'] += 1
    else:
      phases.append(dict(name=n, start=s, end=e, busy=e - s, n=1))
  print('%-24s %9s %9s %9s %6s' % ('phase', 'start ms', 'span ms', 'busy ms', 'n'))
  merged = []
  for p in phases:
    if p['end'] - p['start'] < 0.3e6 and merged and merged[-1]['name'] == 'misc':
      m = merged[-1]
      m['end'] = max(m['end'], p['end']); m['busy'] += p['busy']; m['n'] += p['n']
    elif p['end'] - p['start'] < 0.3e6:
      merged.append(dict(name='misc', start=p['start'], end=p['end'], busy=p['busy'], n=p['n']))
    else:
      merged.append(p)
  for p in merged:
    print('%-24s %9.2f %9.2f %9.2f %6d' % (p['name'], (p['start'] - t0) / 1e6,
                                            (p['end'] - p['start']) / 1e6, p['busy'] / 1e6, p['n']))
  if '--streams' in sys.argv:
    # the same pass stream by stream: runs of one kernel name on one stream (what runs beside what)
    print('\nper stream: runs of one kernel (start ms, end ms, busy ms, launches)')
    for st_id in sorted(set(r[4] for r in sel)):
      runs = []
      for s, e, n, g, st in sel:
        if st != st_id:
          continue
        if runs and runs[-1][0] == n and s - runs[-1][2] < 0.2e6:
          runs[-1][2] = max(runs[-1][2], e); runs[-1][3] += e - s; runs[-1][4] += 1
        else:
          runs.append([n, s, e, e - s, 1])
      print('  stream %d' % st_id)
      for n, s, e, b, c in runs:
        if e - s >= 0.05e6:
          print('    %-24s %9.2f %9.2f %9.2f %6d' % (n, (s - t0) / 1e6, (e - t0) / 1e6, b / 1e6, c))
  tot = OrderedDict()
  for s, e, n, g, st in sel:
    a = tot.setdefault(n, [0, 0])
    a[0] += e - s
    a[1] += 1
  print('\nper kernel: busy ms / launches')
  for n, (b, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print('  %-24s %9.2f %6d' % (n, b / 1e6, c))
  # union of busy intervals -> idle time
  cur_e, idle = sel[0][0], 0
  for s, e, n, g, st in sel:
    if s > cur_e:
      idle += s - cur_e
    cur_e = max(cur_e, e)
  print('idle (no kernel running): %.2f ms' % (idle / 1e6))
  for key in ('gru_step_tiny_kernel', 'gru_step_mid_kernel'):
    small = [(g, e - s) for s, e, n, g, st in sel if n == key]
    if not small:
      continue
    print('\n%s: duration vs grid (workgroups)' % key)
    buckets = [(0, 128), (128, 256), (256, 512), (512, 1024), (1024, 2048), (2048, 4096),
               (4096, 8192), (8192, 1 << 30)]
    for a, b in buckets:
      d = [x[1] for x in small if a <= x[0] < b]
      if d:
        print('  grid %5d..%-6s n=%4d  avg %7.1f us  min %7.1f  max %7.1f  total %7.2f ms'
              % (a, b if b < 1 << 30 else 'inf', len(d), sum(d) / len(d) / 1e3, min(d) / 1e3,
                 max(d) / 1e3, sum(d) / 1e6))


if __name__ == '__main__':
  main()
