mkdir -p gpurun_out/r04k
Q="--cpu_batches 0 --host_steps 0 --cached_steps 0 --rank_check 0 --train_steps 0 --pin_shapes 1"
python bench.py --steps 2 --warmup 1 $Q > gpurun_out/r04k/w1_pinned.json 2>/dev/null
timeout 1200 python bench.py --gpus 8 --steps 1 --warmup 1 $Q > gpurun_out/r04k/w8_pinned.json 2> gpurun_out/r04k/w8.err
python - <<PY
import json
a=json.loads(open('gpurun_out/r04k/w1_pinned.json').read().strip().splitlines()[-1])
b=json.loads(open('gpurun_out/r04k/w8_pinned.json').read().strip().splitlines()[-1])
print('crc', a['ranks_crc32'], b['ranks_crc32'], a['ms_per_step'], b['ms_per_step'])
PY
export GPU_MAX_HW_QUEUES=8
python tools/host_lead.py --config icep --steps 30 --rounds 2 --arms resident,upload,pull:copy,pull:s3,prefetch:before:copy,prefetch:before:s3 2>&1 | tail -8 | tee gpurun_out/r04k/host_lead_icep_hwq8.txt
python tools/host_lead.py --config didemo_recon --steps 30 --rounds 2 --arms resident,pull:copy,prefetch:before:copy 2>&1 | tail -5 | tee gpurun_out/r04k/host_lead_didemo_hwq8.txt
