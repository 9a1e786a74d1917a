#!/usr/bin/env python3
"""Reduce two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected SEPARATELY as
MI355X_MICROARCH.md's HBM section prescribes) into per-kernel fabric traffic per launch.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/pmc_FETCH_SIZE -- python3 bench.py ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/pmc_WRITE_SIZE -- python3 bench.py ...
  python tools/pmc_traffic.py out/pmc_FETCH_SIZE out/pmc_WRITE_SIZE > profiles/rNN_pmc_hbm_traffic.json

Units / corrections: the counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of
wide (16 B per lane) reads at 64 bytes, so it is doubled (every global read of these kernels is a
dwordx4 or a dword-per-lane row walk; the doubling is an upper bound for the latter)."""
import collections
import csv
import glob
import json
import os
import sys


def load(directory, counter):
  files = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
  if not files:
    raise SystemExit('no counter_collection.csv under ' + directory)
  per_kernel = collections.defaultdict(list)
  for r in csv.DictReader(open(files[0])):
    if r['Counter_Name'] == counter:
      per_kernel[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
  return per_kernel


def main():
  fetch = load(sys.argv[1], 'FETCH_SIZE')
  write = load(sys.argv[2], 'WRITE_SIZE')
  out = {}
  for k in fetch:
    if 'cmhse' not in k:
      continue
    n = len(fetch[k])
    f, w = sum(fetch[k]), sum(write.get(k, [0.0]))
    out[k] = {'launches': n, 'FETCH_SIZE_KiB_sum': f, 'WRITE_SIZE_KiB_sum': w,
              'hbm_read_bytes_per_launch_corrected': 2.0 * f * 1024.0 / n,
              'hbm_write_bytes_per_launch': w * 1024.0 / max(1, len(write.get(k, [0.0])))}
  json.dump(out, sys.stdout, indent=1)
  print()


if __name__ == '__main__':
  main()
