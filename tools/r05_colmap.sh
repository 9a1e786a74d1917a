#!/bin/bash
# A/B + PMC: half the A-row fabric traffic through the ticket -> tile map alone (no kernel change)
out=gpurun_out/r05f
mkdir -p $out
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
python tools/ab_pass.py --modes "tune.chain_col_map=0;tune.chain_col_map=1" --rounds 4 --passes 3 2>&1 | grep -v amdgpu | tee $out/ab.txt
python tools/ab_pass.py --modes "tune.chain_col_map=0;tune.chain_col_map=1" --rounds 4 --passes 4 --n_videos 615 2>&1 | grep -v amdgpu | tee -a $out/ab.txt
cat > /tmp/colmap_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['R']); sys.path.insert(0, os.path.join(os.environ['R'], 'tools'))
import torch, bench
from cmhse_amd import ops, synthetic
from cmhse_amd.evaluation import encode_data_device
from cmhse_amd.model import VSE
ops.tune('chain_col_map', int(sys.argv[1]))
wl = dict(bench.WORKLOADS['anet_icep_val']); opt = bench.make_opt(wl, 'attention', 1024)
torch.cuda.set_device(0); torch.manual_seed(1); model = VSE(opt)
spec = synthetic.anet_like_spec(wl['n_videos'], seed=0, dataset='anet')
batches = bench.build_loader(spec, wl, torch.device('cuda', 0), 0, (spec.n_videos + 31) // 32)
for _ in range(3):
  encode_data_device(opt, model, batches, logging=lambda *a, **k: None)
  torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp R=$R
for m in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$out/pmc_${m}_$c -- python3 /tmp/colmap_run.py $m > /dev/null 2>&1
  done
  (cd $R && python tools/pmc_traffic.py $out/pmc_${m}_FETCH_SIZE $out/pmc_${m}_WRITE_SIZE > $out/traffic_map$m.json)
  rm -rf $R/$out/pmc_${m}_FETCH_SIZE $R/$out/pmc_${m}_WRITE_SIZE
done
cd $R
python - <<'PY'
import json
for m in (0, 1):
  d = json.load(open('gpurun_out/r05f/traffic_map%d.json' % m))
  for k, v in d.items():
    if 'chain' in k:
      print('col_map', m, k[:60], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
