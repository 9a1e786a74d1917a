#!/usr/bin/env python3
"""Tail of one reference-API validation pass from a rocprofv3 trace directory (--kernel-trace
--memory-copy-trace of tools/api_path_profile.py): every kernel and copy from the end of level 1 to
the end of the pass, with start / duration / bytes, so that one can see what the device-to-host
staging overlaps with.   python tools/api_tail_trace.py <dir> [events=70]"""
import csv
import glob
import sys


def main():
  d = sys.argv[1]
  n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
  ev = []
  for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
      ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:60], ''))
  for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
      ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r.get('Direction', ''),
                 r.get('Size', r.get('Bytes', ''))))
  ev.sort()
  tail = ev[-n:]
  t0 = tail[0][0]
  for s, e, name, size in tail:
    print('%10.3f ms  %9.1f us  %-64s %s' % ((s - t0) / 1e6, (e - s) / 1e3, name, size))


if __name__ == '__main__':
  main()
