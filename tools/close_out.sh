#!/bin/bash
# round close-out: full GPU test suite, the default bench line, profiles (tools/collect_profiles.sh); copy the results to profiles/rNN_*
R=${1:-r06_final}
N=${2:-6}
out=gpurun_out/$R
mkdir -p $out
python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee $out/summary.txt
grep -h "passed\|failed" $out/pytest_gpu.log | tail -2 | tee -a $out/summary.txt
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" | tee -a $out/summary.txt
bash tools/collect_profiles.sh $R $N > $out/collect.log 2>&1
# a rank's share of the split (8 / 4 / 2 ranks) under the whole split's plan, and the pass through the reference API
# (every rank of the deal and the whole split in one process: slowest rank, implied efficiency; both deals at 8 ranks)
for w in 8 4 2; do python tools/rank_share.py --world $w --all_ranks 1 --steps 8 --warmup 2 2>/dev/null | tail -1; done > $out/rank_share.jsonl
python tools/rank_share.py --world 8 --all_ranks 1 --deal lpt --steps 8 --warmup 2 2>/dev/null | tail -1 >> $out/rank_share.jsonl
# the 8-rank path at full size on THIS box's GPU(s) (gloo when there are fewer GPUs than ranks): a functional run; its CRC must be the single process's
Q="--fast_steps 0 --power_steps 0 --train_steps 0 --host_steps 0 --cpu_batches 0 --rank_check 0 --cached_steps 0 --api_steps 0"
timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 $Q > $out/bench_w8.json 2> $out/bench_w8.err
python tools/api_path_profile.py --passes 8 2>&1 | grep -v amdgpu.ids > $out/api_path.txt
python tools/api_path_profile.py --passes 8 --host 1 2>&1 | grep -v amdgpu.ids >> $out/api_path.txt
python - $out <<'PY' | tee -a $out/summary.txt
import json, sys
d = json.loads([l for l in open(sys.argv[1] + '/bench.json') if l.startswith('{')][0])
print('ms', round(d['ms_per_step'], 2), 'value', round(d['value'] / 1e6, 2), 'M pairs/s  frac', round(d['roofline']['frac'], 4), 'crc', d['ranks_crc32'])
print('leg_seconds', d.get('leg_seconds'))
print('cpu_baseline', {k: d['cpu_baseline'].get(k) for k in ('value', 'cores', 'kind')} if 'cpu_baseline' in d else None)
print('noise', {k: d['rank_noise_floor'].get(k) for k in ('videos', 'max_abs_embedding_diff', 'random_init', 'scorer_only', 'correlated')} if 'rank_noise_floor' in d else None)
print('train', {k: round(v['ms_per_step'], 2) for k, v in d.get('train_steps', {}).items() if isinstance(v, dict) and 'ms_per_step' in v})
print('pcie', d.get('pcie_inclusive', {}).get('ms_per_step'))
print('dropin_validate', {k: (round(v['ms_per_step'], 2), v.get('vs_device_pass') or v.get('vs_pcie_inclusive')) for k, v in d.get('dropin_validate', {}).items() if isinstance(v, dict)})
print('fast_mode', {k: d.get('fast_mode', {}).get(k) for k in ('ms_per_step', 'max_abs_embedding_diff_vs_fp32', 'rank_rows_moved_on_correlated_embeddings')})
try:
  w8 = json.loads([l for l in open(sys.argv[1] + '/bench_w8.json') if l.startswith('{')][0])
  print('8 ranks:', w8['config']['backend'], 'crc', w8['ranks_crc32'], 'equal to the single process:', w8['ranks_crc32'] == d['ranks_crc32'])
except Exception as e:
  print('8 ranks: no line', e)
PY
