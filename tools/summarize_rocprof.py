#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short markdown table
(kernel names truncated) for profiles/.  Usage: summarize_rocprof.py <kernel_stats.csv> [title]"""
import csv
import sys


def main():
  path = sys.argv[1]
  title = sys.argv[2] if len(sys.argv) > 2 else path
  rows = list(csv.DictReader(open(path)))
  total = sum(float(r['TotalDurationNs']) for r in rows)
  print('# %s\n' % title)
  print('source: `%s` (rocprofv3 --kernel-trace --stats), total kernel time %.3f ms\n'
        % (path, total / 1e6))
  print('| kernel | calls | total ms | avg us | min us | max us | % |')
  print('|---|---|---|---|---|---|---|')
  for r in rows:
    name = r['Name']
    if len(name) > 70:
      name = name[:67] + '...'
    print('| `%s` | %s | %.3f | %.2f | %.2f | %.2f | %.2f |' % (
        name, r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3,
        float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, float(r['Percentage'])))


if __name__ == '__main__':
  main()
