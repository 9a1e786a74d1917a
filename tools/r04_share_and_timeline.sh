mkdir -p gpurun_out/r04j
Q="--cpu_batches 0 --host_steps 0 --cached_steps 0 --rank_check 0 --train_steps 0"
for n in 615 1230 2460; do python bench.py --n_videos $n --steps 12 --warmup 3 $Q > gpurun_out/r04j/share_$n.json 2>/dev/null; done
timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 $Q > gpurun_out/r04j/w8_shared_gpu.json 2> gpurun_out/r04j/w8.err
python bench.py --steps 3 --warmup 2 $Q > gpurun_out/r04j/w1_ref.json 2>/dev/null
for c in icep c3d icep_recon didemo_recon; do python tools/train_profile.py --config $c --steps 12 --feed prefetch --timeline 2 2>&1 | grep -v "^Eit" | tail -24; done > gpurun_out/r04j/steady_timeline_prefetch.txt
for c in icep c3d icep_recon didemo_recon; do python tools/train_profile.py --config $c --steps 12 --feed resident --timeline 2 2>&1 | grep -v "^Eit" | tail -24; done > gpurun_out/r04j/steady_timeline_resident.txt
FEED=prefetch bash tools/profile_train.sh r04j/prof_prefetch icep c3d icep_recon didemo_recon
tail -2 gpurun_out/r04j/w8.err; ls gpurun_out/r04j
