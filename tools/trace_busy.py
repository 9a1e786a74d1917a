#!/usr/bin/env python3
"""Kernel-time sum against wall for the timed region of a rocprofv3 --kernel-trace CSV: the region
after the LAST idle gap of >= 200 ms (tools/train_profile.py sleeps there).

  python tools/trace_busy.py <kernel_trace.csv> [steps] [--timeline]

--timeline adds, for the LAST step of the region, each stream's launches collapsed into runs of one
kernel: start and end (ms from the step's first launch), launches, busy ms.
"""
import csv
import sys
from collections import defaultdict


def main():
  rows = []
  for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                 int(r['Stream_Id'])))
  argv = [a for a in sys.argv[1:] if not a.startswith('--')]
  steps = int(argv[1]) if len(argv) > 1 else 1
  rows.sort()
  cut, cur_e = 0, rows[0][1]
  for i, (s, e, n, st) in enumerate(rows):
    if s - cur_e >= 200e6:
      cut = i
    cur_e = max(cur_e, e)
  sel = rows[cut:]
  t0, t1 = sel[0][0], max(r[1] for r in sel)
  busy_sum = sum(e - s for s, e, n, st in sel)
  union, cur_e, gaps = 0, t0, []
  for s, e, n, st in sel:
    if s > cur_e:
      gaps.append(s - cur_e)
      cur_s = s
    else:
      cur_s = cur_e
    if e > cur_s:
      union += e - cur_s
    cur_e = max(cur_e, e)
  wall = t1 - t0
  print('timed region: %d launches, wall %.3f ms (%.3f ms / step over %d steps)'
        % (len(sel), wall / 1e6, wall / 1e6 / steps, steps))
  print('kernel time sum %.3f ms (%.3f / step); union over streams %.3f ms (%.1f %% of wall); '
        'idle %.3f ms in %d gaps (%.1f us avg)'
        % (busy_sum / 1e6, busy_sum / 1e6 / steps, union / 1e6, 100.0 * union / wall,
           (wall - union) / 1e6, len(gaps), (sum(gaps) / max(1, len(gaps))) / 1e3))
  per = defaultdict(lambda: [0, 0])
  for s, e, n, st in sel:
    key = n.split('(')[0][-60:]
    per[key][0] += e - s
    per[key][1] += 1
  print('\n| kernel | launches / step | ms / step | avg us |')
  print('|---|---|---|---|')
  for k, (b, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:28]:
    print('| `%s` | %.1f | %.3f | %.1f |' % (k, c / steps, b / 1e6 / steps, b / c / 1e3))
  streams = defaultdict(int)
  for s, e, n, st in sel:
    streams[st] += e - s
  print('\nper stream busy ms / step: ' + ', '.join('%d: %.3f' % (k, v / 1e6 / steps)
                                                      for k, v in sorted(streams.items())))
  # Host lead: how long after the optimizer update of step k (the last multi_tensor / fused-Adam
  # launch of a step) the first projection / chain kernel of step k + 1 starts.
  is_adam = lambda n: 'multi_tensor_apply' in n or 'FusedAdam' in n or 'fused_adam' in n
  is_chain = lambda n: ('xproj_kernel' in n or 'gru_step' in n or 'gru_chain' in n or
                        'gru_fwd_tail' in n or 'pull_steps' in n)
  leads, last_adam_end, armed = [], None, False
  for s_, e_, n_, _ in sel:
    if is_adam(n_):
      last_adam_end = e_ if last_adam_end is None or not armed else max(last_adam_end, e_)
      armed = True
    elif armed and is_chain(n_):
      leads.append((s_ - last_adam_end) / 1e3)
      armed = False
  if leads:
    leads_sorted = sorted(leads)
    print('\nfirst projection / chain kernel of a step after the previous step\'s Adam: median %.1f us, '
          'max %.1f us over %d step boundaries (%s)'
          % (leads_sorted[len(leads) // 2], leads_sorted[-1], len(leads),
             ', '.join('%.0f' % v for v in leads)))
  if '--timeline' in sys.argv:
    # last step = launches after the region's last gap of >= 0.25 of the mean step... simpler:
    # the final 1/steps of the wall
    t_lo = t1 - wall // steps
    last = [r for r in sel if r[0] >= t_lo]
    base = last[0][0]
    by_stream = defaultdict(list)
    for r in last:
      by_stream[r[3]].append(r)
    for st, rs in sorted(by_stream.items()):
      print('\nstream %d (last step):' % st)
      run = None
      for s_, e_, n_, _ in rs:
        key = n_.split('(')[0][-48:]
        if run and run[0] == key and s_ - run[2] < 50e3:
          run[2], run[3], run[4] = e_, run[3] + 1, run[4] + (e_ - s_)
        else:
          if run:
            print('  %7.3f - %7.3f  %4d x %-48s busy %.3f' % ((run[1] - base) / 1e6, (run[2] - base) / 1e6, run[3], run[0], run[4] / 1e6))
          run = [key, s_, e_, 1, e_ - s_]
      if run:
        print('  %7.3f - %7.3f  %4d x %-48s busy %.3f' % ((run[1] - base) / 1e6, (run[2] - base) / 1e6, run[3], run[0], run[4] / 1e6))


if __name__ == '__main__':
  main()
