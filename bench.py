#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: video<->paragraph pairs encoded + scored per second.

One "step" = one full validation pass of the hot path over a synthetic ActivityNet-val-shaped
split: encode every clip / sentence / whole-video stream / paragraph with the
hierarchical GRU encoders (level 1 + level 2, L2-normalise, per-loader-batch ContrastiveLoss like
evaluation.py:129), then score all N x N video-paragraph pairs in both directions (i2t + t2i:
ranks, top-1, Recall@K).  pairs = N^2; inputs are resident in HBM before the timed region.

Workloads (config.workload):
  anet_icep_val  the metric's configuration (BASELINE configs[4], north_star "ActivityNet-ICEP val"):
                 img_dim=2048, embed=1024, attention pooling, loader batch 32, over the
                 val_1-shaped split N=4917, sumC~17.5k  [default]
  anet_c3d_val   configs[1] model (img_dim=500) over the same split
  didemo_icep_val configs[3] (img_dim=2048, all-80-frame clips, vocab 7205, N=1004)
  plumbing       configs[0]: 64 videos x 4 clips x 10 frames, batch 16

N>1, one rank per GPU, backend nccl = RCCL: the SAME split is sharded over ranks (strong scaling):
each rank encodes its share, embeddings are all-gathered, each rank scores its row stripe
(cmhse_amd/parallel_eval.py).  Launched either by torch.distributed.run (RANK / WORLD_SIZE /
MASTER_* in the environment) or bare — `python bench.py --gpus N` with no WORLD_SIZE set starts
its own N ranks as fresh child processes (the parent never touches the GPU, nothing is re-exec'd)
and exits with the worst child's return code.  On a box with fewer GPUs than ranks the children
share the GPUs over a gloo group (a functional run of the N-rank path, flagged in the JSON).

The JSON line also carries
  roofline      for the dominant kernel (the LDS-tiled GRU step: gru_step_chain_kernel, one launch per
                chain of time steps, and gru_step_kernel): algorithmic FLOPs of the timed launches
                (2*3H*I + 2*3H*H + 14H per sequence-step) / their HIP-event time,
                against the exact-fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md);
  roofline_sim  the similarity GEMM (sim_kernel<Rank>): 2*nrows*M*D FLOP per direction over the
                HIP-event time of that launch alone, against the same peak (north_star: "MFMA
                utilisation on the similarity GEMM");
  rank_check    in-run correctness signal: i2t/t2i ranks of the HIP path on
                synthetic.correlated_embeddings(N, 1024, 3.0) against an fp64 NumPy recomputation
                of a 256-row sample (mismatches must be 0);
  pcie_inclusive the same pass with the loader batches in pinned HOST memory, uploaded inside the
                timed pass (the reference's loader hands over host tensors, model.py:225-227);
  cpu_baseline  the torch-CPU restatement of the path (oracle/cmhse_torch_cpu.py: nn.GRU over
                pack_padded_sequence, kind "torch-cpu") timed on this host's cores on a bounded
                sample of the same workload, extrapolated to the full split (encode ~ N,
                scoring ~ N^2) — a reported baseline, never the thing measured as `value`;
                cpu_baseline_numpy (--cpu_numpy 1): the NumPy oracle ("port") the same way;
  rank_noise_floor  the HIP path against that oracle on the SAME sample, end to end: largest embedding
                difference, rank rows on which the two runs differ (random-init encoders: every score
                within ~1e-3 of every other, so this is the floor any fp32 evaluation order has), and the
                rows the same deviation moves on separable (correlated) embeddings — the ruler
                every optional math mode is held to;
  cached_schedule_pass  the pass for a resident, unchanged split with the level-1 schedules kept
                between passes (opt-in; the headline pass rebuilds them like the reference).
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

sys.path.insert(0, os.path.join(REPO, 'tools'))

from cmhse_amd import ops, parallel_eval, synthetic  # noqa: E402
from cmhse_amd.evaluation import encode_data_device  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402
# workloads / synthetic loader batches (tools/bench_common.py) and the supplementary legs
# (tools/bench_legs.py); re-exported here because the tools and the full-size tests import them from bench
from bench_common import (BF16_MFMA_PEAK_TFLOPS, FP32_MFMA_PEAK_TFLOPS, WORKLOADS, build_loader,  # noqa: E402,F401
                          device_batch, gru_flops_per_step, make_opt)
from bench_legs import (TRAIN_CONFIGS, cpu_baseline, dropin_validate_bench, fast_mode_bench, power_probe,  # noqa: E402,F401
                        isa_audit, measured_clock_ghz, profile_source,
                        measured_step_latency_us, measured_traffic, rank_check, rank_noise_floor,
                        train_bench, train_step_work)

def costs_sum(costs, idx):
  return float(sum(costs[i][0] for i in idx))


def rank_environments(n, n_gpus, port, base):
  """The environment of each of the `n` rank processes launch_ranks starts on a box with `n_gpus`
  GPUs: the torch.distributed rendezvous variables (127.0.0.1: the container's hostname may not
  resolve), one GPU per rank (LOCAL_RANK), dmabuf IPC for RCCL (HSA_ENABLE_IPC_MODE_LEGACY=0), and —
  with fewer GPUs than ranks — the gloo backend, ranks sharing GPUs round-robin (RCCL needs one device
  per rank): a functional run of the N-rank path, reported as such ("backend" in the JSON)."""
  env = dict(base)
  env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(n),
             HSA_ENABLE_IPC_MODE_LEGACY=base.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
  if n_gpus < n:
    env['CMHSE_BENCH_BACKEND'] = 'gloo'
  else:
    env.pop('CMHSE_BENCH_BACKEND', None)
  return [dict(env, RANK=str(r), LOCAL_RANK=str(r % n_gpus)) for r in range(n)]


def launch_ranks(n):
  """`python bench.py --gpus N` from a bare shell: start N rank processes of this same command
  (fresh children; this parent has not initialised the GPU — torch.cuda.device_count() does not —
  and nothing is exec'd over it), wait for them, return the worst return code.  Rank 0's stdout
  is this process's stdout, so exactly one JSON line comes out."""
  import socket
  import subprocess
  n_gpus = torch.cuda.device_count()
  if n_gpus < 1:
    sys.stderr.write('bench.py: no GPU visible\n')
    return 2
  with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
  if n_gpus < n:
    sys.stderr.write('bench.py: %d GPU(s) for %d ranks: sharing GPUs over a gloo group\n'
                     % (n_gpus, n))
  procs = []
  for r, e in enumerate(rank_environments(n, n_gpus, port, os.environ)):
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                  stdout=None if r == 0 else subprocess.DEVNULL))
  worst, live = 0, set(range(n))
  while live:
    for r in sorted(live):
      rc = procs[r].poll()
      if rc is None:
        continue
      live.discard(r)
      if rc != 0:
        worst = worst or rc
        for q in live:          # a dead rank leaves the others waiting in a collective
          procs[q].terminate()
    time.sleep(0.05)
  return worst


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=3)
  ap.add_argument('--warmup', type=int, default=1)
  ap.add_argument('--workload', default='anet_icep_val', choices=sorted(WORKLOADS))
  ap.add_argument('--rnn_type', default='attention', choices=['attention', 'maxout', 'seq2seq'])
  ap.add_argument('--embed', type=int, default=1024)
  ap.add_argument('--n_videos', type=int, default=0, help='override the split size (debug)')
  ap.add_argument('--fast_steps', type=int, default=5,
                  help='also time this many passes in the opt-in bf16x3 math mode (0 = skip).  Reported as the '
                       'fenced `fast_mode` object: the mode is outside the bit-identical-ranks contract and never `value`')
  ap.add_argument('--power_steps', type=int, default=4,
                  help='extra passes per math mode beside which socket power and shader clock are sampled in-process '
                       '(the `power` object; 0 = off)')
  ap.add_argument('--train_steps', type=int, default=10,
                  help='also time this many VSE.train_emb steps per BASELINE training config (0 = skip)')
  ap.add_argument('--train_configs', default='anet_c3d_tau0,anet_icep_tau0,anet_icep_recon,didemo_icep_recon',
                  help='comma-separated subset of %s' % ', '.join(sorted(TRAIN_CONFIGS)))
  ap.add_argument('--host_steps', type=int, default=10,
                  help='also time this many passes with the loader batches in pinned HOST memory '
                       '(PCIe-inclusive rate; never the headline value)')
  ap.add_argument('--api_steps', type=int, default=10,
                  help='also time this many passes through the reference API (train.py:223-236 via cmhse_amd/dropin: '
                       'encode_data -> NumPy -> i2t -> t2i), resident and host-fed; 0 = skip')
  ap.add_argument('--rank_check', type=int, default=1,
                  help='check HIP ranks on correlated embeddings against fp64 NumPy (0 = skip)')
  ap.add_argument('--cpu_batches', type=int, default=16,
                  help='loader batches in the sample of the CPU baseline (default 16 x 32 = the 512-video subset '
                       'BASELINE.md section 3 names; 0 = skip)')
  ap.add_argument('--cpu_numpy', type=int, default=0,
                  help='1: also time the NumPy oracle ("port", rounds 1-3\'s baseline) on the same sample')
  ap.add_argument('--pin_shapes', type=int, default=0,
                  help='1: every GRU step on the LDS-tiled kernel whatever its active count (tiny_max_seqs = '
                       'mid_max_seqs = 0).  An A/B switch only since round 5: the DEFAULT run is partition-'
                       'independent bit for bit (every rank picks its kernels from the whole split\'s step plan, '
                       'parallel_eval.global_step_plan), so ranks_crc32 is the same for any --gpus without it')
  ap.add_argument('--cached_steps', type=int, default=5,
                  help='also time this many passes that reuse the level-1 schedules of a resident '
                       'split (evaluation.encode_group(plan=)); 0 = skip')
  args = ap.parse_args()

  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    sys.exit(launch_ranks(args.gpus))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
  # CMHSE_BENCH_BACKEND=gloo: the N-rank path (deal, gathers, merge, per-rank report) with ranks
  # sharing GPUs on a box that has fewer of them than ranks (set by launch_ranks)
  backend = os.environ.get('CMHSE_BENCH_BACKEND', 'nccl')
  # The ISA audit of the loaded library disassembles it in child processes (llvm-objdump, ~3 s): done
  # HERE, before this process touches the GPU — a fork of a process that holds a GPU context, and three
  # idle seconds in the middle of the run, cost the legs behind it 10 % (profiles/r06_fast_mode_ab.txt).
  lib_info = isa_audit() if rank == 0 else None
  if backend == 'gloo':
    local_rank = local_rank % max(1, torch.cuda.device_count())
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)
  if world > 1:
    import torch.distributed as dist
    if backend == 'gloo':
      dist.init_process_group('gloo')
    else:
      dist.init_process_group('nccl', device_id=device)

  if args.pin_shapes:
    ops.tune('tiny_max_seqs', 0)
    ops.tune('mid_max_seqs', 0)
  wl = dict(WORKLOADS[args.workload])
  if args.n_videos:
    wl['n_videos'] = args.n_videos
  opt = make_opt(wl, args.rnn_type, args.embed)
  torch.manual_seed(1)
  model = VSE(opt)       # same seed on every rank -> replicated weights
  if wl['dataset'] == 'uniform':
    spec = synthetic.uniform_spec(wl['n_videos'], clips=4, frames=10, words=12)
  else:
    spec = synthetic.anet_like_spec(wl['n_videos'], seed=0, dataset=wl['dataset'])
  # batches are dealt to ranks by work (GRU FLOPs of their frame / word steps), longest paragraph
  # first: every rank derives the same assignment from the split's sizes alone
  costs = [parallel_eval.batch_cost(lc, lv, lw, lp, wl['img_dim'], 300, args.embed)
           for lc, lv, lw, lp in synthetic.batch_lengths(spec, wl['batch'])]
  _lens = synthetic.batch_lengths(spec, wl['batch'])
  tail_from = parallel_eval.tail_horizon(np.concatenate([np.concatenate([lw, lp]) for _, _, lw, lp in _lens]))
  assignment = parallel_eval.assign_batches(costs, world, tail_from=tail_from)
  batches = build_loader(spec, wl, device, assignment[rank])
  N = spec.n_videos
  quiet = lambda *a, **k: None

  from cmhse_amd.evaluation import report_from_ranks
  phase_ms, phase_sum = {}, {}
  last = {}

  debug = os.environ.get('CMHSE_BENCH_DEBUG', '0') == '1'   # host-side marks of every pass on stderr

  host_queue_ms = []     # per pass: host ms until the encoders / the ranking were queued

  def step(plan=None, src=None):
    """One validation pass, scored: the embeddings are encoded, both directions ranked, the
    per-batch meters replayed, and the ranks brought to the host and turned into the Recall@K /
    median-rank report (evaluation.py:173-184) — all inside the timed region.  A pass does all the
    work the reference's pass does, the sort / step counts / pointer tables of its packed schedules
    included (layers.py:94-97); only the `cached_schedule_pass` leg passes a `plan` that keeps them."""
    if world == 1:
      h0 = time.perf_counter()
      cat, _, _, finish_log = encode_data_device(opt, model, batches if src is None else src,
                                                 logging=quiet, defer_logging=True, plan=plan)
      h1 = time.perf_counter()
      r_i, t_i = ops.sim_rank(cat['vid_emb'], cat['para_emb'])
      r_t, t_t = ops.sim_rank(cat['para_emb'], cat['vid_emb'])
      packed = torch.stack([r_i, t_i, r_t, t_t])
      h2 = time.perf_counter()
      host_queue_ms.append(((h1 - h0) * 1e3, (h2 - h1) * 1e3))
      finish_log()     # the per-batch 'Letest' meters (evaluation.py:129), after the ranking is queued
      host = packed.cpu().numpy().astype(np.float64)     # ONE device-to-host copy for both directions
      last['rep_i'], last['rep_t'] = report_from_ranks(host[0]), report_from_ranks(host[2])
      if debug:
        sys.stderr.write('pass host ms: encode queued %.1f, ranking queued %.1f, scored %.1f\n'
                         % ((h1 - h0) * 1e3, (h2 - h1) * 1e3, (time.perf_counter() - h2) * 1e3))
      return host[0], host[2]
    res = parallel_eval.validate_sharded(opt, model, batches, device=device, dim=args.embed,
                                         assignment=assignment, timings=phase_ms, plan=plan)
    last['rep_i'], last['rep_t'] = res[0], res[1]
    for k, v in phase_ms.items():
      phase_sum[k] = phase_sum.get(k, 0.0) + v
    return res[2], res[3]

  def sync():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  # warm-up passes run under the same timers as the timed ones: their HIP events come from (and go
  # back to) the library's free list, so the timed region creates none
  for _ in range(args.warmup):
    with ops.StepTimers() as wt, ops.SimTimers() as wst:
      step()
      sync()
    wt.collect()
    wst.collect()
  sync()
  phase_sum.clear()      # the per-rank phase times below are those of the timed passes only
  if os.environ.get('CMHSE_BENCH_GC_FREEZE', '0') == '1':
    # (opt-in since late round 4: encode_data_device itself keeps Python's cyclic collector out of
    # the stretch in which a pass's launches are queued — what this used to be for — so the timed
    # region runs as a caller's would; `pass_ms.python_gc_ms` reports what the collector took)
    import gc
    gc.collect()
    gc.freeze()
  t0 = time.perf_counter()
  with ops.StepTimers() as timers, ops.SimTimers() as sim_timers:
    pass_marks = [t0]
    del host_queue_ms[:]
    gc_spans = []
    def _gc_cb(phase, info, _t=[0.0]):
      if phase == 'start':
        _t[0] = time.perf_counter()
      else:
        gc_spans.append(((time.perf_counter() - _t[0]) * 1e3, info.get('generation')))
    import gc as _gc
    _gc.callbacks.append(_gc_cb)
    for _ in range(args.steps):
      ranks_i, ranks_t = step()
      pass_marks.append(time.perf_counter())   # (a pass ends with its ranks on the host)
    sync()
  elapsed = time.perf_counter() - t0
  _gc.callbacks.remove(_gc_cb)
  timed_host_queue = list(host_queue_ms)
  pass_list = [(b - a) * 1e3 for a, b in zip(pass_marks[:-1], pass_marks[1:])]
  pass_ms = sorted((b - a) * 1e3 for a, b in zip(pass_marks[:-1], pass_marks[1:]))
  my_elapsed = elapsed
  if world > 1:
    t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if backend == 'gloo' else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  # Roofline of the dominant kernel (the LDS-tiled GRU step: gru_step_chain_kernel — the tiled steps
  # of a call in one launch — and gru_step_kernel): every one of its launches in the timed region
  # sits between its own HIP event pair (cmhse_timer_tiled), priced with its algorithmic FLOPs.  `all_step_kernels` adds the small-batch step kernel (the ragged
  # tails), from the event spans around each encoder call's whole step sequence.
  spans = timers.collect()
  sims = sim_timers.collect()
  sim_ms = sum(x[0] for x in sims)
  sim_flops = sum(2.0 * x[1] * x[2] * x[3] for x in sims)
  sim_achieved = sim_flops / (sim_ms * 1e-3) / 1e12 if sim_ms > 0 else 0.0
  per_rank = [{'ms_per_step': my_elapsed / args.steps * 1e3}]
  if world > 1:
    # per-rank phase times of the TIMED passes: device time between HIP events on the rank's
    # stream (parallel_eval._Phases) — the passes are not synchronised for them
    mine = [my_elapsed / args.steps * 1e3] + [phase_sum.get(k, 0.0) / max(1, args.steps)
                                               for k in ('encode_ms', 'exchange_ms', 'score_ms')] + \
        [float(phase_ms.get('videos', 0)), costs_sum(costs, assignment[rank])]
    t = torch.tensor(mine, dtype=torch.float64, device='cpu' if backend == 'gloo' else device)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t)
    per_rank = [{'ms_per_step': float(x[0]), 'encode_ms': float(x[1]), 'exchange_ms': float(x[2]),
                 'score_ms': float(x[3]), 'videos': int(x[4]), 'gru_tflop': float(x[5]) / 1e12}
                for x in allt]
  flops_all = sum(sum_T * gru_flops_per_step(I, H)
                  for (_, _, metas, _) in spans for (_, sum_T, I, H, _, _) in metas)
  ms_all = sum(s[0] for s in spans)
  launches_all = sum(s[1] for s in spans)
  ms = sum(s[3][0] for s in spans)
  flops = sum(s[3][1] for s in spans)
  alg_bytes = sum(s[3][2] for s in spans)
  launches = sum(s[3][3] for s in spans)
  achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
  ms_per_step = elapsed / args.steps * 1e3
  pairs = float(N) * float(N)

  clk = measured_clock_ghz()
  traffic = measured_traffic(('gru_step_kernel<', 'gru_step_chain_kernel<'))
  if rank == 0:
    r1 = 100.0 * float((np.asarray(ranks_i) < 1).mean())
    out = {
        'metric': 'video-paragraph pairs encoded+scored per second (full hot path: hierarchical '
                  'GRU encode + i2t + t2i)',
        'value': pairs * args.steps / elapsed, 'unit': 'pairs/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': args.workload, 'n_videos': N, 'n_clips': len(spec.frames_per_clip),
                   'loader_batch': wl['batch'], 'img_dim': wl['img_dim'], 'embed': args.embed,
                   'rnn_type': args.rnn_type,
                   'step': 'encode_data + i2t + t2i over the split, ranks on the host and the '
                           'Recall@K / median-rank report computed (evaluation.py:173-184)',
                   'sharding': 'loader batches dealt to ranks by GRU work + the per-rank tail of its longest paragraph '
                               '(parallel_eval.assign_batches), one all-gather of the embeddings, row-stripe scoring',
                   'schedule_cache': 'none: every timed pass rebuilds its packed schedules (sequence sort, '
                                     'step counts, pointer tables) as the reference does (layers.py:94-97); '
                                     '`cached_schedule_pass` times the opt-in that keeps them for a resident split',
                   'pin_shapes': bool(args.pin_shapes),
                   'backend': ('single process' if world == 1 else
                               ('RCCL (nccl), one rank per GPU' if backend != 'gloo' else
                                'gloo, %d ranks sharing %d GPU(s): functional run of the N-rank '
                                'path, not a scaling measurement' % (world, torch.cuda.device_count())))},
        'videos_per_s': N * args.steps / elapsed, 'r1_i2t_random_init': r1,
        # spread of the timed passes on rank 0 (`ms_per_step` is their mean, whatever the spread)
        'pass_ms': {'min': pass_ms[0], 'median': pass_ms[len(pass_ms) // 2], 'max': pass_ms[-1],
                    'slowest_pass': int(np.argmax(pass_list)),
                    'host_ms_until_encoders_queued': ([round(x[0], 2) for x in timed_host_queue] or None),
                    'python_gc_ms': round(sum(x[0] for x in gc_spans), 2),
                    'python_gc_full_collections': sum(1 for x in gc_spans if x[1] == 2)},
        'report_i2t_random_init': {k: float(v) for k, v in last['rep_i'].items()},
        'report_t2i_random_init': {k: float(v) for k, v in last['rep_t'].items()},
        # identity of the integer ranks of both directions (partition-independent by construction:
        # the same value for every n_gpus)
        'ranks_crc32': zlib.crc32(np.asarray(ranks_i, dtype=np.int64).tobytes() +
                                  np.asarray(ranks_t, dtype=np.int64).tobytes()),
        'per_rank': per_rank,
        'roofline': {'kernel': 'gru_step_chain_kernel / gru_step_kernel (the LDS-tiled GRU step: one launch per '
                               'chain of time steps, or per step where a chain cannot be used)',
                     'bound': 'mfma', 'achieved': achieved,
                     'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': achieved / FP32_MFMA_PEAK_TFLOPS,
                     'traffic': traffic,
                     'traffic_source': profile_source('r*_pmc_hbm_traffic.json'),
                     'traffic_unit': 'fabric bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + '
                                     'WRITE_SIZE, newest profiles/r*_pmc_hbm_traffic.json)',
                     'algorithmic_bytes_per_launch': (alg_bytes / launches) if launches else None,
                     # north_star: achieved GB/s of the GRU = PMC traffic / live launch duration
                     # (against ~8000 GB/s HBM; counts Infinity-Cache hits too — the kernel is
                     # MFMA-bound, this is context)
                     'traffic_gbps': (traffic / (ms * 1e-3 / launches) / 1e9)
                                     if (traffic and launches and ms > 0) else None,
                     'launches': launches,
                     'avg_launch_us': (ms * 1e3 / launches) if launches else None,
                     'flops_per_launch': (flops / launches) if launches else None,
                     'kernel_time_share': (ms * 1e-3) / elapsed if elapsed > 0 else None,
                     # context: `peak` assumes 2.4 GHz; an instrumented build of this kernel read
                     # this in-kernel clock under load (profiles/r01_tile_trace.txt)
                     'in_kernel_clock_ghz': clk,
                     'in_kernel_clock_source': profile_source('r*_tile_trace.txt'),
                     'all_step_kernels': {
                         'achieved': flops_all / (ms_all * 1e-3) / 1e12 if ms_all > 0 else None,
                         'launches': launches_all,
                         'avg_launch_us': (ms_all * 1e3 / launches_all) if launches_all else None,
                         'kernel_time_share': (ms_all * 1e-3) / elapsed if elapsed > 0 else None}},
    }
    out['roofline_sim'] = {
        'kernel': 'sim_kernel<Rank>', 'bound': 'mfma', 'achieved': sim_achieved,
        'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
        'frac': sim_achieved / FP32_MFMA_PEAK_TFLOPS, 'launches': len(sims),
        'avg_launch_us': (sim_ms * 1e3 / len(sims)) if sims else None,
        'flops_per_launch': (sim_flops / len(sims)) if sims else None,
        'algorithmic_bytes_per_launch': (sum(4.0 * (x[1] + x[2]) * x[3] + 8.0 * x[1] for x in sims)
                                         / len(sims)) if sims else None,
        'traffic': measured_traffic('sim_kernel<1'),
        'traffic_source': profile_source('r*_pmc_hbm_traffic.json'),
        'note': '2*nrows*M*D FLOP per direction / HIP-event time of the counting pass alone '
                '(cmhse_sim_rank_ex timer); rank 0\'s stripe when n_gpus > 1'}
    out['library'] = lib_info          # cmhse_version() + the ISA audit of the library this run loaded
    leg_seconds = out['leg_seconds'] = {}

    def leg(name, fn):
      """A supplementary leg must never cost the headline line: a failure is reported in its place."""
      t_leg = time.perf_counter()
      try:
        out[name] = fn()
      except Exception as e:      # noqa: BLE001  (reported, not swallowed)
        out[name] = {'error': '%s: %s' % (type(e).__name__, e)}
        sys.stderr.write('bench.py: leg %s failed: %r\n' % (name, e))
      leg_seconds[name] = round(time.perf_counter() - t_leg, 2)

    if args.rank_check:
      leg('rank_check', lambda: rank_check(N, args.embed))
    if world == 1 and args.cached_steps > 0:
      def cached():
        """The same pass for a split that stays resident and unchanged between passes: the level-1
        schedules are built once (the first pass below) and reused — an opt-in of this package
        (encode_group(plan=)), not what the reference does, hence not the headline."""
        plan = {}
        step(plan)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.cached_steps):
          step(plan)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.cached_steps
        return {'steps': args.cached_steps, 'ms_per_step': dt * 1e3, 'value': pairs / dt, 'unit': 'pairs/s'}
      leg('cached_schedule_pass', cached)
    if world == 1 and args.fast_steps > 0:
      leg('fast_mode', lambda: fast_mode_bench(opt, model, batches, N, args.fast_steps))
    if world == 1 and args.power_steps > 0:
      def power_leg():
        def fast():
          ops.set_math_mode('bf16x3')
          try:
            step()
          finally:
            ops.set_math_mode('fp32')
        return power_probe({'exact_fp32': step, 'bf16x3': fast}, args.power_steps, device)
      leg('power', power_leg)
    if world == 1 and args.train_steps > 0:
      # the BASELINE training configurations (configs[1..3]) as driver-timed train_emb steps —
      # timed while the GPU is still at its working clocks (right behind the validation passes:
      # after the tens of idle seconds of the CPU baseline the first hundred milliseconds of GPU
      # work run at idle clocks, and a training leg is only a few hundred milliseconds long)
      out['train_steps'] = {}
      for name in args.train_configs.split(','):
        if not name:
          continue
        t_leg = time.perf_counter()
        try:
          out['train_steps'][name] = train_bench(name, args.embed, args.rnn_type, args.train_steps, device)
        except Exception as e:    # noqa: BLE001
          out['train_steps'][name] = {'error': '%s: %s' % (type(e).__name__, e)}
          sys.stderr.write('bench.py: train leg %s failed: %r\n' % (name, e))
        leg_seconds['train_steps.' + name] = round(time.perf_counter() - t_leg, 2)
      if 'anet_icep_tau0' in out['train_steps']:      # round 2's key, same configuration
        out['train_step'] = out['train_steps']['anet_icep_tau0']
    host_batches = [None]
    if world == 1 and args.host_steps > 0:
      def pcie_leg():
        # the reference's loader contract hands over host tensors (activity_net/data.py:114-150):
        # same pass, inputs pulled from pinned host memory inside the timed region
        def to_pinned(t):      # straight into page-locked memory (no pageable stop-over: 14.7 GB)
          return torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t)
        host = [tuple(to_pinned(t) if isinstance(t, torch.Tensor) and t.is_cuda else t for t in b)
                for b in batches]
        torch.cuda.synchronize()
        host_batches[0] = host

        step(src=host)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.host_steps):
          step(src=host)        # the headline pass itself (report on the host included), fed from the host
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / args.host_steps
        nbytes = sum(t.numel() * t.element_size() for b in host for t in b[:4])
        return {'steps': args.host_steps, 'ms_per_step': dt * 1e3, 'value': pairs / dt,
                'unit': 'pairs/s', 'host_bytes_per_pass': nbytes,
                'note': 'loader batches in pinned host memory, pulled chunk by chunk under the '
                        'step pipeline inside the timed pass; not the headline value'}
      leg('pcie_inclusive', pcie_leg)
    if world == 1 and args.api_steps > 0:
      # the pass through the reference's own API: NumPy out of encode_data, NumPy
      # into i2t / t2i; `vs_device_pass` / `vs_pcie_inclusive` = its time over the corresponding leg above
      def api_leg():
        crc = lambda ri, rt: zlib.crc32(np.asarray(ri, dtype=np.int64).tobytes() +
                                        np.asarray(rt, dtype=np.int64).tobytes())
        # the device-resident pass once more, right next to the API passes (the headline above was timed
        # a minute earlier in this process)
        step()
        torch.cuda.synchronize()
        t_adj = time.perf_counter()
        for _ in range(5):
          step()
        torch.cuda.synchronize()
        adjacent_ms = (time.perf_counter() - t_adj) / 5 * 1e3
        res = dropin_validate_bench(opt, model, {'resident': batches, 'host_fed': host_batches[0]},
                                    args.api_steps, crc)
        res['resident']['vs_device_pass'] = res['resident']['ms_per_step'] / ms_per_step
        res['resident']['device_pass_adjacent_ms'] = adjacent_ms
        res['resident']['vs_device_pass_adjacent'] = res['resident']['ms_per_step'] / adjacent_ms
        res['resident']['ranks_equal_headline'] = res['resident']['ranks_crc32'] == out['ranks_crc32']
        if 'host_fed' in res and isinstance(out.get('pcie_inclusive'), dict) and 'ms_per_step' in out['pcie_inclusive']:
          res['host_fed']['vs_pcie_inclusive'] = res['host_fed']['ms_per_step'] / out['pcie_inclusive']['ms_per_step']
          res['host_fed']['ranks_equal_headline'] = res['host_fed']['ranks_crc32'] == out['ranks_crc32']
        return res
      leg('dropin_validate', api_leg)
      host_batches[0] = None
    if world == 1 and args.cpu_batches > 0:
      # north_star's baseline: "the reference PyTorch CPU path" -> the torch-CPU restatement; the NumPy
      # port (rounds 1-3's cpu_baseline) beside it on the same sample
      leg('cpu_baseline', lambda: cpu_baseline('torch-cpu', wl, opt, model, spec, args.cpu_batches, N))
      if args.cpu_numpy:
        leg('cpu_baseline_numpy', lambda: cpu_baseline('port', wl, opt, model, spec, args.cpu_batches, N))
      # the exact path's own distance from the oracle on the same sample, rank by rank
      leg('rank_noise_floor', lambda: rank_noise_floor(wl, opt, model, spec, args.cpu_batches, N))
    pw = out.get('power') if isinstance(out.get('power'), dict) else None
    if pw and pw.get('available') and pw.get('exact_fp32', {}).get('sclk_mhz_mean'):
      # context only (`frac` stays priced against the nominal peak): the matrix peak at the shader clock the
      # box held under this pass — boxes differ by 5 % in what their power limit leaves
      sclk = pw['exact_fp32']['sclk_mhz_mean']
      rf = out['roofline']
      rf['sclk_mhz_under_load'] = sclk
      rf['peak_at_that_clock'] = rf['peak'] * sclk / 2400.0
      rf['frac_at_that_clock'] = rf['achieved'] / rf['peak_at_that_clock']
      rf['sclk_source'] = 'the `power` leg of this run (amdsmi, in-process); nominal 2400 MHz'
    print(json.dumps(out))
    sys.stdout.flush()
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
