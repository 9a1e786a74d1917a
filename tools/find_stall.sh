#!/bin/bash
# which HIP API call blocks the host inside a validation pass?  rocprofv3 --hip-trace of tools/pass_jitter.py,
# then every API call longer than 5 ms with the calls around it.
OUT=${1:-stall}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
D=$R/gpurun_out/$OUT; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --output-format csv -d $D/t -- python3 $R/tools/pass_jitter.py --passes 10 > $D/jitter.txt 2>&1
cat $D/jitter.txt | grep "pass "
F=$(ls $D/t/*/*hip_api_trace.csv | head -1)
python3 - "$F" > $D/long_calls.txt <<'PY'
import collections, csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Function']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t0 = rows[0][0]
print('%d API calls' % len(rows))
# pass boundaries: the host-side sync of a pass (hipEventSynchronize / hipMemcpy* > 150 ms)
cuts = [i for i, (s, e, f) in enumerate(rows) if e - s > 150e6]
for a, b in zip(cuts[:-1], cuts[1:]):
  seg = rows[a + 1:b]
  if not seg:
    continue
  lo, hi = rows[a][1], rows[b][0]
  per = collections.defaultdict(lambda: [0, 0, 0])
  api = 0
  for s, e, f in seg:
    per[f][0] += 1; per[f][1] += e - s; per[f][2] = max(per[f][2], e - s)
    api += e - s
  gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2], seg[i + 1][2]) for i in range(len(seg) - 1))[-3:]
  print('host busy window %.2f ms (from %.1f): %d calls, %.2f ms inside the API, %.2f outside' % ((hi - lo) / 1e6, (lo - t0) / 1e6, len(seg), api / 1e6, (hi - lo - api) / 1e6))
  for f, (n, tot, mx) in sorted(per.items(), key=lambda kv: -kv[1][1])[:5]:
    print('    %-32s n %5d  total %8.2f ms  avg %7.1f us  max %8.1f us' % (f, n, tot / 1e6, tot / n / 1e3, mx / 1e3))
  print('    largest gaps between calls (us): ' + '; '.join('%.0f after %s before %s' % g for g in gaps))
PY
cat $D/long_calls.txt | tail -80
rm -rf $D/t
