#!/usr/bin/env python3
"""One host-fed validation pass from a rocprofv3 kernel trace of tools/api_path_profile.py --host 1:
the pull kernels (PCIe), the step-chain launches and the attention launches of the LAST device pass
on one time axis.   python tools/host_fed_trace.py <trace dir>"""
import csv
import glob
import sys


def main():
  rows = []
  for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
      rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Stream_Id'])))
  rows.sort()
  # passes end with two sim_kernel<1 launches; take the last complete pass
  ends = [i for i, r in enumerate(rows) if 'sim_kernel<1' in r[2]]
  hi = ends[-1]
  lo = ends[-3] + 1 if len(ends) >= 3 else 0
  sel = rows[lo:hi + 1]
  t0 = sel[0][0]
  keys = ['pull_steps_kernel', 'gru_step_chain_kernel', 'gru_step_kernel', 'attn_energy_kernel', 'attn_pool_kernel', 'gru_step_mid_kernel',
          'xproj_kernel', 'push_bytes_kernel', 'sim_kernel']
  cur = None
  for s, e, n, st in sel:
    k = next((x for x in keys if x in n), None)
    if k is None:
      continue
    if k == 'gru_step_mid_kernel':
      if cur and cur[0] == k and cur[3] == st:
        cur[2] = e
        cur[4] += 1
        continue
    if cur:
      print('%9.2f -> %9.2f ms  %8.2f ms  stream %3d  %-24s x%d' % ((cur[1] - t0) / 1e6, (cur[2] - t0) / 1e6, (cur[2] - cur[1]) / 1e6, cur[3], cur[0], cur[4]))
    cur = [k, s, e, st, 1]
  if cur:
    print('%9.2f -> %9.2f ms  %8.2f ms  stream %3d  %-24s x%d' % ((cur[1] - t0) / 1e6, (cur[2] - t0) / 1e6, (cur[2] - cur[1]) / 1e6, cur[3], cur[0], cur[4]))


if __name__ == '__main__':
  main()
