#!/bin/bash
# A/B after the bf16x3 tile loop moved to v_mfma_f32_32x32x8_bf16_1k pairs (profiles/r05_bf16_mfma_bystander.txt):
#   libcmhse_hip.so    the product build
#   libcmhse_16k.so    -DCMHSE_BF3_MFMA_32X32X16 (the single gfx950 instruction, as before)
#   libcmhse_vec.so    the vectorisers left on for device code (v_pk_fma_f32 in six kernels, as before)
# built here (CPU side) with:  python tools/r05_mfma_fix_ab.sh build
if [ "$1" = build ]; then
  python - <<'PY'
import sys; sys.path.insert(0, '.')
from cmhse_amd import build
print(build.build())
print(build.build(extra_flags=['-DCMHSE_BF3_MFMA_32X32X16'], out='cmhse_amd/libcmhse_16k.so'))
print(build.build(device_flags=[], out='cmhse_amd/libcmhse_vec.so'))
PY
  exit 0
fi
O=gpurun_out/r05m; mkdir -p $O
for lib in hip 16k vec; do
  CMHSE_HIP_LIB=$PWD/cmhse_amd/libcmhse_$lib.so python bench.py --steps 10 --warmup 3 --cpu_batches 0 --fast_steps 6 > $O/bench_$lib.json 2> $O/bench_$lib.err
  python - "$O/bench_$lib.json" "$lib" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fm = d.get('fast_mode', {})
print(sys.argv[2], 'exact ms/pass %.2f' % d['ms_per_step'], 'crc', d.get('ranks_crc32'), 'bf16x3:', json.dumps(fm)[:400])
PY
done
for lib in hip 16k; do
  for n in steps chain; do
    CMHSE_HIP_LIB=$PWD/cmhse_amd/libcmhse_$lib.so python tools/pkfma_canary.py --modes bf16x3 --reps 60 --neighbour $n 2>&1 | grep "neighbour\|library" | sed "s/^/$lib $n: /"
  done
done
