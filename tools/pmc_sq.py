#!/usr/bin/env python3
"""Per-kernel SQ counter summary from one rocprofv3 --kernel-trace --pmc pass: how busy the matrix
pipes were and what clock the chip held.

  MFMA busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x elapsed cycles)
  elapsed cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 reports the sum over the 8 XCDs)
  clock       = elapsed cycles / kernel duration (MI355X_MICROARCH.md 'DVFS give-back')
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY are quad-cycles summed over waves."""
import collections
import csv
import glob
import os
import sys


def main():
  root = sys.argv[1]
  cc = glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)
  kt = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)
  if not cc:
    raise SystemExit('no counter_collection.csv under ' + root)
  per = collections.defaultdict(lambda: collections.defaultdict(float))
  calls = collections.defaultdict(set)
  for r in csv.DictReader(open(cc[0])):
    k = r['Kernel_Name'].split('(')[0]
    if 'cmhse' not in k:
      continue
    per[k][r['Counter_Name']] += float(r['Counter_Value'])
    calls[k].add(r.get('Dispatch_Id'))
  dur = collections.defaultdict(float)
  if kt:
    for r in csv.DictReader(open(kt[0])):
      dur[r['Kernel_Name'].split('(')[0]] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
  names = sorted({c for v in per.values() for c in v})
  print('| kernel | launches | total ms | ' + ' | '.join(names) + ' | MFMA busy | clock GHz |')
  print('|---|---|---|' + '---|' * (len(names) + 2))
  for k, v in sorted(per.items(), key=lambda kv: -dur.get(kv[0], 0.0)):
    cyc = v.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    busy = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (1024.0 * cyc) if cyc > 0 else float('nan')
    clk = cyc / dur[k] if dur.get(k, 0.0) > 0 else float('nan')
    print('| `%s` | %d | %.2f | %s | %.3f | %.2f |' % (
        k[-56:], len(calls[k]), dur.get(k, 0.0) / 1e6,
        ' | '.join('%.4g' % v.get(c, 0.0) for c in names), busy, clk))


if __name__ == '__main__':
  main()
