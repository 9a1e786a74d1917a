# round 4: traces and timelines of the training step fed the default way (pinned host batches straight
# into train_emb), a soak, the full GPU suite and the driver's bench command
mkdir -p gpurun_out/r04u
python tools/soak_train.py --config icep_recon --epochs 2 --steps 150 2>&1 | tail -4 > gpurun_out/r04u/soak.txt
python tools/soak_train.py --config didemo_recon --epochs 1 --steps 150 2>&1 | tail -3 >> gpurun_out/r04u/soak.txt
cat gpurun_out/r04u/soak.txt
for c in icep c3d icep_recon didemo_recon; do python tools/train_profile.py --config $c --steps 12 --feed auto --timeline 2 2>&1 | grep -v "^Eit" | tail -24; done > gpurun_out/r04u/steady_timeline_auto.txt
FEED=auto bash tools/profile_train.sh r04u/prof_auto icep c3d icep_recon didemo_recon
grep -h "first projection" gpurun_out/r04u/prof_auto/train_step_*.md
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r04u/bench.json 2> gpurun_out/r04u/bench.err; tail -4 gpurun_out/r04u/bench.err
