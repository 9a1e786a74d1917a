#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/$1 (default r02_final):
#   rocprofv3 kernel stats + trace of the default bench command, PMC fabric traffic (two separate
#   passes), SQ MFMA-busy counters, training-step traces, the rasterisation A/B with FETCH_SIZE, the
#   upload-pipeline sweep, in-kernel clock / phase stamps.  Run from the repo root via gpurun.
OUT=${1:-r02_final}
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
CMHSE_GRU_RASTER=4 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/pmc_FETCH_R4 -- python3 $R/$B > /dev/null 2>&1
for c in c3d icep_recon; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/train_$c -- python3 $R/tools/train_profile.py --config $c --steps 10 > $D/train_$c.txt 2>/dev/null
done
cd $R
python tools/pmc_traffic.py $D/pmc_FETCH_SIZE $D/pmc_WRITE_SIZE > $D/pmc_hbm_traffic.json
python tools/pmc_traffic.py $D/pmc_FETCH_R4 $D/pmc_WRITE_SIZE > $D/pmc_hbm_traffic_raster4.json
python tools/pmc_sq.py $D/pmc_SQ > $D/pmc_sq_summary.md 2>&1
rm -rf $D/pmc_FETCH_SIZE $D/pmc_WRITE_SIZE $D/pmc_FETCH_R4 $D/pmc_SQ
python tools/summarize_rocprof.py $D/stats/*/*kernel_stats.csv "round 2, final kernels: rocprofv3 --kernel-trace --stats -- python3 $B (anet_icep_val, exact fp32)" > $D/kernel_stats.md
python tools/trace_timeline.py $D/stats/*/*kernel_trace.csv > $D/pass_timeline.txt
for c in c3d icep_recon; do
  python tools/trace_busy.py $D/train_$c/*/*kernel_trace.csv 10 --timeline > $D/train_step_$c.md
  python tools/train_profile.py --config $c --timeline 1 >> $D/train_$c.txt 2>/dev/null
  rm -rf $D/train_$c
done
rm -f $D/stats/*/*kernel_trace.csv
python bench.py > $D/bench_default.json 2> $D/bench_default.err
python bench.py --workload anet_c3d_val --host_steps 0 --cpu_batches 0 > $D/bench_c3d.json 2>/dev/null
python bench.py --workload didemo_icep_val --host_steps 0 --cpu_batches 0 --fast_steps 0 --train_steps 0 > $D/bench_didemo.json 2>/dev/null
python tools/ab_pass.py --modes "CMHSE_GRU_RASTER=0;CMHSE_GRU_RASTER=4;CMHSE_GRU_RASTER=8;CMHSE_GRU_RASTER=16" --rounds 3 > $D/raster_ab_icep.txt 2>&1
python tools/ab_pass.py --modes "CMHSE_GRU_MSUB=1;CMHSE_GRU_MSUB=2;CMHSE_X=1" --rounds 3 > $D/tile_height_ab.txt 2>&1
for m in 1 2 1 2; do CMHSE_GRU_MSUB=$m bash tools/power_probe.sh "tile rows $((64*m))" python tools/ab_pass.py --modes "CMHSE_X=1" --rounds 6 --passes 3; done > $D/power_probe.txt 2>&1
python tools/ab_host.py --rounds 2 --modes "PIPE=0;PIPE=1;PIPE=1,CMHSE_PULL_GRID=16,CMHSE_PULL_THREADS=256;PIPE=1,CMHSE_PULL_GRID=128" > $D/upload_pipeline.txt 2>&1
bash tools/mid_shape_sweep.sh > $D/step_sweep.txt 2>&1
python tools/ab_train.py --config c3d --modes "CMHSE_BWD_MID_MAX_SEQS=0,CMHSE_MID_UNITS=16,CMHSE_MID_WAVES=4,CMHSE_BATCHED_LOSSES=0,CMHSE_FUSED_ADAM=0;CMHSE_BATCHED_LOSSES=0;CMHSE_X=1" > $D/train_ab.txt 2>&1
python tools/ab_train.py --config icep_recon --modes "CMHSE_BWD_MID_MAX_SEQS=0,CMHSE_MID_UNITS=16,CMHSE_MID_WAVES=4,CMHSE_BATCHED_LOSSES=0;CMHSE_X=1" >> $D/train_ab.txt 2>&1
timeout 60 tools/microbench/weights_reread.bin > $D/weights_reread.txt 2>&1
python tools/mid_trace.py 152 12 > $D/mid_trace_S152.txt 2>&1
python tools/mid_trace.py 8 12 > $D/mid_trace_S8.txt 2>&1
python tools/mid_trace.py 32 12 > $D/mid_trace_S32.txt 2>&1
python tools/tile_trace.py 22419 2048 1024 > $D/tile_trace.txt 2>&1
ls -la $D
