#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/$1 (default r03_final):
#   rocprofv3 kernel stats + trace of the default bench command, PMC fabric traffic (two separate
#   passes), SQ MFMA-busy counters, training-step traces and timelines, the training-schedule A/B,
#   stand-alone rates of the weight-gradient products and of the BPTT step, the small-batch step
#   sweep, in-kernel clock / phase stamps.  Run from the repo root via gpurun.
OUT=${1:-r03_final}
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
D=$R/gpurun_out/$OUT
mkdir -p $D
B="bench.py --steps 2 --warmup 1 --cpu_batches 0 --fast_steps 0 --train_steps 0 --host_steps 0 --rank_check 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/$B > $D/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/pmc_FETCH_SIZE -- python3 $R/$B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/pmc_WRITE_SIZE -- python3 $R/$B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $D/pmc_SQ -- python3 $R/$B > $D/pmc_SQ.log 2>&1
for c in c3d icep icep_recon; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/train_$c -- python3 $R/tools/train_profile.py --config $c --steps 10 > $D/train_$c.txt 2>/dev/null
done
for cfg in "9600 2048" "9600 500" "3400 300" "1216 2048"; do set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/wg_$1_$2 -- python3 $R/tools/bench_wgrad.py --S $1 --I $2 --reps 3 2>/dev/null | tail -1 >> $D/wgrad_rate.txt
  grep -h "gemm_tn_rows\|tn_rows_reduce" $D/wg_$1_$2/*/*kernel_stats.csv | cut -c1-120 >> $D/wgrad_rate.txt
  rm -rf $D/wg_$1_$2
done
for arm in "tune.bwd_split_min_seqs=0,side_streams=0" "tune.bwd_split_min_seqs=33,side_streams=0"; do
  echo "== $arm" >> $D/bptt_step.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/bp -- python3 $R/tools/bench_bptt.py --arms "$arm" 2>/dev/null | tail -1 >> $D/bptt_step.txt
  grep -h "bwd_rec_part\|bwd_gates\|gru_bwd_step" $D/bp/*/*kernel_stats.csv | cut -c1-120 >> $D/bptt_step.txt
  rm -rf $D/bp
done
cd $R
python tools/pmc_traffic.py $D/pmc_FETCH_SIZE $D/pmc_WRITE_SIZE > $D/pmc_hbm_traffic.json
python tools/pmc_sq.py $D/pmc_SQ > $D/pmc_sq_summary.md 2>&1
rm -rf $D/pmc_FETCH_SIZE $D/pmc_WRITE_SIZE $D/pmc_SQ
python tools/summarize_rocprof.py $D/stats/*/*kernel_stats.csv "round 3, final kernels: rocprofv3 --kernel-trace --stats -- python3 $B (anet_icep_val, exact fp32)" > $D/kernel_stats.md
python tools/trace_timeline.py $D/stats/*/*kernel_trace.csv > $D/pass_timeline.txt
rm -rf $D/stats
for c in c3d icep icep_recon; do
  python tools/trace_busy.py $D/train_$c/*/*kernel_trace.csv 10 --timeline > $D/train_step_$c.md
  python tools/train_profile.py --config $c --timeline 1 >> $D/train_$c.txt 2>/dev/null
  rm -rf $D/train_$c
done
python bench.py --steps 20 --warmup 5 > $D/bench_default.json 2> $D/bench_default.err
python bench.py --workload anet_c3d_val --host_steps 0 --cpu_batches 0 --train_steps 0 > $D/bench_c3d.json 2>/dev/null
python bench.py --workload didemo_icep_val --host_steps 0 --cpu_batches 0 --fast_steps 0 --train_steps 0 > $D/bench_didemo.json 2>/dev/null
python bench.py --gpus 2 --n_videos 1230 --steps 3 --warmup 1 --fast_steps 0 --train_steps 0 --host_steps 0 --cpu_batches 0 > $D/bench_w2_shared_gpu.json 2>/dev/null
for c in c3d icep icep_recon; do
  python tools/ab_train.py --config $c --rounds 5 --steps 10 --modes "schedule=towers,side_streams=0,tune.bwd_split_min_seqs=0;schedule=interleaved,side_streams=0,tune.bwd_split_min_seqs=0;schedule=interleaved,side_streams=1,tune.bwd_split_min_seqs=0;schedule=interleaved,side_streams=1" 2>&1 | grep -v amdgpu >> $D/train_ab.txt
  # the late-round-3 ladder: node per level + operator-by-operator losses + per-step tail + 1024-row chunks, then one change at a time
  python tools/ab_train.py --config $c --rounds 5 --steps 10 --modes "schedule=levels,fused_losses=0,tune.bwd_tail_min_steps=0,tune.fwd_tail_min_steps=0,tune.bwd_chunk_rows=1024;schedule=levels,tune.bwd_tail_min_steps=0,tune.fwd_tail_min_steps=0,tune.bwd_chunk_rows=1024;tune.bwd_tail_min_steps=0,tune.fwd_tail_min_steps=0,tune.bwd_chunk_rows=1024;tune.fwd_tail_min_steps=0,tune.bwd_chunk_rows=1024;tune.bwd_chunk_rows=1024;label=default" 2>&1 | grep -v amdgpu >> $D/train_ab_late.txt
done
python tools/step_sweep.py --sizes 1,8,16,32,64,152,320,512,1024 --dims 500,300,1024 --arms "tune.mid_max_seqs=0;tune.mid_units=16;tune.mid_units=0" > $D/step_sweep.txt 2>&1
python tools/tile_trace.py 22419 2048 1024 > $D/tile_trace.txt 2>&1
ls -la $D
