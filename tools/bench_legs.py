"""The supplementary legs of bench.py: the CPU baselines, the in-run rank check, the BASELINE training
configurations as timed VSE.train_emb steps (resident and PCIe-inclusive), the opt-in bf16x3 pass, and
the readers of committed profile artefacts (PMC traffic, in-kernel clock, step latency).  None of
them produces the headline `value`; bench.py guards each so that a failure costs only that leg."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
  sys.path.insert(0, REPO)

from cmhse_amd import ops, synthetic  # noqa: E402
from cmhse_amd.evaluation import encode_data_device  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402
from bench_common import (BF16_MFMA_PEAK_TFLOPS, FP32_MFMA_PEAK_TFLOPS, device_batch,  # noqa: E402,F401
                          make_opt)


def _host_cpu_model():
  try:
    for line in open('/proc/cpuinfo'):
      if line.startswith('model name'):
        return line.split(':', 1)[1].strip()
  except OSError:
    pass
  return ''


_CPU_SAMPLE = {}


def cpu_sample(wl, spec, n_batches):
  """The first `n_batches` loader batches of the split as host 12-tuples (generated once, shared by
  the two CPU legs)."""
  key = (wl['img_dim'], wl['vocab'], wl['batch'], n_batches)
  if key not in _CPU_SAMPLE:
    sub = synthetic.SplitSpec(spec.num_clips, spec.frames_per_clip, spec.frames_per_video,
                              spec.words_per_sent)
    nv = min(spec.n_videos, n_batches * wl['batch'])
    sub.num_clips = sub.num_clips[:nv]
    nc = sum(sub.num_clips)
    sub.frames_per_clip = sub.frames_per_clip[:nc]
    sub.words_per_sent = sub.words_per_sent[:nc]
    sub.frames_per_video = sub.frames_per_video[:nv]
    _CPU_SAMPLE[key] = synthetic.make_batches(sub, wl['batch'], wl['img_dim'], wl['vocab'], seed=0,
                                              feat=wl['feat'])
  return _CPU_SAMPLE[key]


def cpu_baseline(kind, wl, opt, model, spec, n_sample_batches, n_full, repeats=3):
  """The CPU restatements of the path on this host's cores over the first `n_sample_batches`
  loader batches (a 512-video subset, one warm-up pass, median of >= 3 timed
  passes, scaled encode ~ N and scoring ~ N^2).
    kind 'torch-cpu'  oracle/cmhse_torch_cpu.py: nn.GRU over pack_padded_sequence + the pooling on
                      CPU tensors, one loader batch at a time — the stand-in for "the reference
                      PyTorch CPU path" north_star names (layers.py:75-79,93-119);
    kind 'port'       oracle/cmhse_oracle.py: the NumPy oracle (OpenBLAS)."""
  sys.path.insert(0, os.path.join(REPO, 'oracle'))
  nv = min(spec.n_videos, n_sample_batches * wl['batch'])
  batches = cpu_sample(wl, spec, n_sample_batches)
  sds = [{k: v.detach().cpu().numpy() for k, v in sd.items()} for sd in model.state_dict(opt)]
  ncpu = os.cpu_count() or 1
  counts = sorted({min(ncpu, c) for c in (16, 32)})
  if kind == 'torch-cpu':
    import cmhse_torch_cpu as impl
    cpu_model = impl.Model(opt.rnn_type, sds)
    data = batches
    encode = lambda bs: impl.encode_data(opt.rnn_type, sds, bs, margin=opt.margin, model=cpu_model)
    old_threads = torch.get_num_threads()

    class limit(object):
      def __init__(self, n):
        self.n = n

      def __enter__(self):
        torch.set_num_threads(self.n)

      def __exit__(self, *a):
        torch.set_num_threads(old_threads)
    what = 'torch CPU restatement (oracle/cmhse_torch_cpu.py: nn.GRU + pack_padded_sequence, torch.set_num_threads'
  else:
    import cmhse_oracle as impl
    from threadpoolctl import threadpool_limits
    data = [tuple(x.numpy() if hasattr(x, 'numpy') else x for x in b) for b in batches]
    encode = lambda bs: impl.encode_data(opt.rnn_type, sds, bs, margin=opt.margin)
    limit = lambda n: threadpool_limits(limits=n)
    what = 'NumPy oracle (oracle/cmhse_oracle.py, OpenBLAS threads'
  # thread count: the per-step GEMMs are small, so all host cores oversubscribe; calibrate on one
  # loader batch over two settings and keep the faster (that count is what `cores` reports)
  with limit(counts[0]):
    encode(data[:1])          # cold start (thread pools, first-touch) outside the calibration
  best_n, best_t = counts[0], None
  for n_thr in counts:
    with limit(n_thr):
      t0 = time.time()
      encode(data[:2])
      dt = time.time() - t0
    if best_t is None or dt < best_t:
      best_n, best_t = n_thr, dt
  enc_s, score_s = [], []
  with limit(best_n):
    # warm-up: the calibration above ran the encoders; the scoring path once (a full warm-up pass
    # would double this leg's share of the default run for nothing)
    res = encode(data[:1])
    impl.i2t(res[0], res[1])
    for rep in range(repeats):
      t0 = time.time()
      res = encode(data)
      t1 = time.time()
      impl.i2t(res[0], res[1])
      impl.t2i(res[0], res[1])
      t2 = time.time()
      enc_s.append(t1 - t0)
      score_s.append(t2 - t1)
  t_enc, t_score = float(np.median(enc_s)), float(np.median(score_s))
  if kind == 'torch-cpu':       # kept for rank_noise_floor(): the oracle's embeddings of the sample
    _CPU_RESULT[(wl['img_dim'], wl['vocab'], wl['batch'], n_sample_batches, opt.rnn_type)] = \
        [np.asarray(r, dtype=np.float32) for r in res[:6]]
  scale = n_full / float(nv)
  t_full = t_enc * scale + t_score * scale * scale
  return {
      'value': n_full * n_full / t_full, 'unit': 'pairs/s', 'cores': best_n,
      # "port" = a restatement of the reference's algorithm (oracle/), not the reference's own files (which
      # cannot travel to the GPU box); `implementation` says which of the two restatements
      'kind': 'port', 'implementation': ('oracle/cmhse_torch_cpu.py (torch CPU ops: nn.GRU over pack_padded_sequence)'
                                         if kind == 'torch-cpu' else 'oracle/cmhse_oracle.py (NumPy / OpenBLAS)'),
      'host_cpu': _host_cpu_model(), 'host_logical_cpus': ncpu,
      'videos_per_s': nv / t_enc, 'passes': repeats,
      'sample': ('%s = %d, the faster of {%s} on two loader batches of this %d-CPU host) on the first '
                 '%d videos (%d loader batches) of the same split: a 2-batch warm-up, median of %d timed '
                 'passes: encode %.2f s, i2t+t2i %.3f s; extrapolated to N=%d with encode ~ N and '
                 'scoring ~ N^2' % (what, best_n, ','.join(str(c) for c in counts), ncpu, nv,
                                    len(batches), repeats, t_enc, t_score, n_full)),
  }


_CPU_RESULT = {}


def rank_diff(r_a, r_b):
  """(rows whose integer rank differs, largest |difference|) of two rank vectors."""
  d = np.abs(np.asarray(r_a, dtype=np.int64) - np.asarray(r_b, dtype=np.int64))
  return int((d != 0).sum()), int(d.max()) if d.size else 0


def rank_noise_floor(wl, opt, model, spec, n_sample_batches, n_full):
  """The EXACT path's own distance from the oracle, on the ruler every math mode is held to
  (VERDICT r04 item 3): the HIP path and the torch-CPU oracle encode the same videos (the CPU
  baseline's sample) and rank them end to end, each with its own scorer —
    max_abs_embedding_diff     over the six embedding matrices (bar: 1e-4);
    random_init                rank rows (both directions) on which the two END-TO-END runs differ, and
                               the largest rank difference: random-init encoders on random inputs put
                               every score within ~1e-3 of every other, so ANY two fp32 evaluation
                               orders disagree on some rows — this is the noise floor, not an error;
    correlated                 the same deviation (HIP - oracle embeddings of the sample, tiled over
                               N rows) applied to SURVEY S5's separable embeddings (R@1 ~ 33 %) and
                               re-ranked by the HIP scorer: rows that move against the unperturbed
                               ranks — what a trained model's ranks would feel of it;
    scorer_only                HIP scorer against an fp64 NumPy scorer on the SAME (oracle) embeddings."""
  sys.path.insert(0, os.path.join(REPO, 'oracle'))
  import cmhse_oracle as oracle
  key = (wl['img_dim'], wl['vocab'], wl['batch'], n_sample_batches, opt.rnn_type)
  batches = cpu_sample(wl, spec, n_sample_batches)
  if key in _CPU_RESULT:
    want = _CPU_RESULT[key]
  else:                               # the CPU leg did not run: encode the sample here
    import cmhse_torch_cpu as impl
    sds = [{k: v.detach().cpu().numpy() for k, v in sd.items()} for sd in model.state_dict(opt)]
    want = [np.asarray(r, dtype=np.float32) for r in
            impl.encode_data(opt.rnn_type, sds, batches, margin=opt.margin)[:6]]
  cat, _, _ = encode_data_device(opt, model, batches, logging=lambda *a, **k: None)
  names = ['vid_emb', 'para_emb', 'clip_emb', 'cap_emb', 'vid_ctx', 'para_ctx']
  got = [cat[k].cpu().numpy() for k in names]
  emb_diff = {k: float(np.abs(g - w).max()) for k, g, w in zip(names, got, want)}
  nv = got[0].shape[0]
  # end to end: each side's own embeddings through its own scorer
  r_i, _ = ops.sim_rank(cat['vid_emb'], cat['para_emb'])
  r_t, _ = ops.sim_rank(cat['para_emb'], cat['vid_emb'])
  _, _, o_i = oracle.i2t(want[0], want[1])
  _, _, o_t = oracle.t2i(want[0], want[1])
  e2e = [rank_diff(r_i.cpu().numpy(), o_i), rank_diff(r_t.cpu().numpy(), o_t)]
  # the scorer alone: HIP ranks of the ORACLE's embeddings against fp64 ranks of the same
  wv, wp = torch.from_numpy(want[0]).cuda(), torch.from_numpy(want[1]).cuda()
  s_i, _ = ops.sim_rank(wv, wp)
  s_t, _ = ops.sim_rank(wp, wv)
  d64 = want[0].astype(np.float64) @ want[1].astype(np.float64).T
  dg = np.diag(d64)
  f_i = (d64 > dg[:, None]).sum(1)
  f_t = (d64.T > dg[:, None]).sum(1)
  sc = [rank_diff(s_i.cpu().numpy(), f_i), rank_diff(s_t.cpu().numpy(), f_t)]
  # the deviation on separable embeddings
  a, b = synthetic.correlated_embeddings(n_full, got[0].shape[1], 3.0, seed=0)
  ad, bd = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
  reps = (n_full + nv - 1) // nv
  dev_v = torch.from_numpy(np.tile(got[0] - want[0], (reps, 1))[:n_full]).cuda()
  dev_p = torch.from_numpy(np.tile(got[1] - want[1], (reps, 1))[:n_full]).cuda()
  base_i, _ = ops.sim_rank(ad, bd)
  base_t, _ = ops.sim_rank(bd, ad)
  pert_i, _ = ops.sim_rank(ops.l2norm_rows(ad + dev_v), ops.l2norm_rows(bd + dev_p))
  pert_t, _ = ops.sim_rank(ops.l2norm_rows(bd + dev_p), ops.l2norm_rows(ad + dev_v))
  corr = [rank_diff(pert_i.cpu().numpy(), base_i.cpu().numpy()), rank_diff(pert_t.cpu().numpy(), base_t.cpu().numpy())]
  return {
      'videos': nv, 'rank_rows': 2 * nv, 'oracle': 'oracle/cmhse_torch_cpu.py embeddings, oracle/cmhse_oracle.py i2t / t2i',
      'max_abs_embedding_diff': max(emb_diff.values()), 'embedding_diff_by_matrix': emb_diff,
      'random_init': {'rank_rows_differing_from_hip': e2e[0][0] + e2e[1][0],
                      'max_abs_rank_diff': max(e2e[0][1], e2e[1][1])},
      'scorer_only': {'rank_rows_differing_from_fp64': sc[0][0] + sc[1][0],
                      'max_abs_rank_diff': max(sc[0][1], sc[1][1])},
      'correlated': {'rank_rows': 2 * n_full, 'rank_rows_moved': corr[0][0] + corr[1][0],
                     'max_abs_rank_diff': max(corr[0][1], corr[1][1]),
                     'note': 'correlated_embeddings(N, D, 3.0) + (HIP - oracle) deviation of the sample, '
                             're-normalised, HIP scorer; bf16x3 moved 6 of 9834 on this ruler (DESIGN section 9)'},
  }


def power_probe(passes, n_steps, device, interval_s=0.05):
  """Socket power and shader clock WHILE the pass loops: a thread samples torch.cuda.power_draw() /
  clock_rate() (amdsmi, in-process: no child process, see profiles/r06_fast_mode_ab.txt) beside
  `n_steps` extra passes of every mode in `passes` {name: callable}.  Runs behind the timed legs and is
  not part of any of them.  Context for roofline.frac: the pass runs at the package power limit, and the
  clock it holds there — not the loop's MFMA occupancy — is what separates it from the nominal peak
  (profiles/r06_fast_mode.txt item 4, r06_fp32_ring.txt)."""
  import threading
  out = {'source': 'amdsmi via torch.cuda.power_draw / clock_rate, sampled in-process every %d ms beside %d extra '
                   'passes per mode; not part of any timed leg' % (int(interval_s * 1e3), n_steps)}
  try:
    torch.cuda.power_draw(device)
    torch.cuda.clock_rate(device)
  except Exception as e:    # noqa: BLE001
    return {'available': False, 'error': '%s: %s' % (type(e).__name__, e)}
  try:
    import amdsmi
    h = torch.cuda._get_amdsmi_handler(device)
    cap = amdsmi.amdsmi_get_power_cap_info(h)
    out['power_cap_w'] = float(cap['power_cap']) / 1e6 if float(cap['power_cap']) > 1e4 else float(cap['power_cap'])
  except Exception:         # noqa: BLE001
    out['power_cap_w'] = None
  for name, fn in passes.items():
    fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sample():
      while not stop.is_set():
        try:
          samples.append((float(torch.cuda.power_draw(device)), float(torch.cuda.clock_rate(device))))
        except Exception:   # noqa: BLE001
          pass
        stop.wait(interval_s)
    th = threading.Thread(target=sample, daemon=True)
    t0 = time.perf_counter()
    th.start()
    for _ in range(n_steps):
      fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_steps
    stop.set()
    th.join()
    w = [x[0] for x in samples[1:]] or [x[0] for x in samples]   # (the first sample predates the load)
    c = [x[1] for x in samples[1:]] or [x[1] for x in samples]
    scale = 1e-3 if w and max(w) > 1e4 else 1.0                   # mW (the API's unit) or W
    out[name] = {'ms_per_step_while_sampled': dt * 1e3, 'samples': len(w),
                 'socket_w_mean': scale * sum(w) / len(w) if w else None,
                 'socket_w_max': scale * max(w) if w else None,
                 'sclk_mhz_mean': sum(c) / len(c) if c else None,
                 'sclk_mhz_min': min(c) if c else None}
  out['available'] = True
  return out


def fast_mode_bench(opt, model, batches, N, n_steps):
  """Supplementary: the same validation pass with CMHSE_MATH_BF16X3 (3-term bf16 hi/lo split on
  the bf16 matrix pipe, fp32 accumulate, for the large encoder GEMMs; ranking stays exact fp32),
  with its measured deviation from the exact-fp32 embeddings and ranks, and its own roofline: the
  tiled step kernel's algorithmic FLOPs over its HIP-event time against the bf16 MFMA peak / 3
  (three MFMAs per product).  Never the headline `value`."""
  quiet = lambda *a, **k: None

  def one_pass():
    cat, _, _ = encode_data_device(opt, model, batches, logging=quiet)
    r_i, _ = ops.sim_rank(cat['vid_emb'], cat['para_emb'])
    r_t, _ = ops.sim_rank(cat['para_emb'], cat['vid_emb'])
    return cat, r_i, r_t

  ref_cat, ref_i, ref_t = one_pass()          # exact fp32
  try:
    ops.set_math_mode('bf16x3')
    with ops.StepTimers() as wt:
      one_pass()
      torch.cuda.synchronize()
    wt.collect()
    t0 = time.perf_counter()
    with ops.StepTimers() as timers:
      for _ in range(n_steps):
        cat, r_i, r_t = one_pass()
      torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_steps
    spans = timers.collect()
  finally:
    ops.set_math_mode('fp32')
  ms = sum(s[3][0] for s in spans)
  flops = sum(s[3][1] for s in spans)
  launches = sum(s[3][3] for s in spans)
  achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
  peak = BF16_MFMA_PEAK_TFLOPS / 3.0
  diff = max(float((cat[k] - ref_cat[k]).abs().max()) for k in ['vid_emb', 'para_emb', 'clip_emb',
                                                                 'cap_emb', 'vid_ctx', 'para_ctx'])
  moved = int((r_i != ref_i).sum()) + int((r_t != ref_t).sum())
  # ranks of a trained model's embeddings: the same perturbation applied to SURVEY S5's correlated
  # embeddings (R@1 ~ 33 %): re-rank normalize(a + (bf16x3 - fp32 deviation of the video rows)),
  # which shows whether a 1e-6 deviation moves any rank where the scores are spread like a real
  # model's (with random-init encoders on random inputs all scores sit within ~1e-3 of each other)
  a, b = synthetic.correlated_embeddings(N, cat['vid_emb'].shape[1], 3.0, seed=0)
  ad, bd = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
  dev_v = cat['vid_emb'] - ref_cat['vid_emb']
  dev_p = cat['para_emb'] - ref_cat['para_emb']
  base_i, _ = ops.sim_rank(ad, bd)
  base_t, _ = ops.sim_rank(bd, ad)
  pert_i, _ = ops.sim_rank(ops.l2norm_rows(ad + dev_v), ops.l2norm_rows(bd + dev_p))
  pert_t, _ = ops.sim_rank(ops.l2norm_rows(bd + dev_p), ops.l2norm_rows(ad + dev_v))
  moved_corr = int((pert_i != base_i).sum()) + int((pert_t != base_t).sum())
  from cmhse_amd.evaluation import report_from_ranks
  keys = ('r1', 'r5', 'r10', 'medr')

  def same_report(a, b):
    ra, rb = report_from_ranks(a.cpu().numpy()), report_from_ranks(b.cpu().numpy())
    return {k: bool(ra[k] == rb[k]) for k in keys}
  return {'math': 'bf16x3: a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x8_bf16_1k pairs, fp32 '
                  'accumulate (encoder GEMMs of steps with > 1024 active sequences and the '
                  'attention projection; pre-split operands staged global -> LDS by global_load_lds_dwordx4 '
                  'into a 3-stage ring); ranking kernel exact fp32',
          'contract': 'OUTSIDE the bit-identical-ranks contract: an opt-in (ops.set_math_mode), never `value`',
          'steps': n_steps, 'ms_per_step': dt * 1e3, 'value': float(N) * N / dt,
          'unit': 'pairs/s', 'max_abs_embedding_diff_vs_fp32': diff,
          'rank_rows_moved_vs_fp32_random_init': moved, 'rank_rows_total': 2 * N,
          'rank_rows_moved_on_correlated_embeddings': moved_corr,
          # Recall@1 / @5 / @50 ('r10' upstream) / median rank of each direction: equal to the exact path's?
          'report_equal_exact_random_init': {'i2t': same_report(r_i, ref_i), 't2i': same_report(r_t, ref_t)},
          'report_equal_exact_correlated': {'i2t': same_report(pert_i, base_i), 't2i': same_report(pert_t, base_t)},
          'isa_audit': isa_audit(),
          'roofline': {'kernel': 'gru_step_kernel<bf16x3>', 'bound': 'mfma', 'achieved': achieved,
                       'peak': peak, 'unit': 'TFLOP/s (fp32-equivalent products)',
                       'frac': achieved / peak, 'launches': launches,
                       'avg_launch_us': (ms * 1e3 / launches) if launches else None,
                       'kernel_time_share': (ms * 1e-3) / (dt * n_steps) if dt > 0 else None,
                       'peak_of_the_instructions_used': peak / 2.0,
                       'frac_of_that': achieved / (peak / 2.0),
                       'note': 'peak = dense bf16 MFMA 2500 TFLOP/s / 3 MFMAs per product; the library restricts '
                               'itself to v_mfma_f32_32x32x8_bf16_1k (two per 16 k, half the rate of the '
                               'v_mfma_f32_32x32x16_bf16 the 2500 are quoted for: DESIGN section 4, the lost-update '
                               'finding), hence peak_of_the_instructions_used = peak / 2; the mode runs at the '
                               'package power limit (the `power` object)'}}


def rank_check(N, D, n_sample=256, seed=0):
  """In-run correctness signal: HIP ranks (both directions) on SURVEY §8d S5's scoring inputs
  (normalize(z + 3 eps) pairs: R@1 ~ 33 %, a non-trivial rank distribution) against an fp64
  recomputation of `n_sample` rows, rank_i = #{j : d_ij > d_ii} (evaluation.py:164-171)."""
  a, b = synthetic.correlated_embeddings(N, D, 3.0, seed=seed)
  ad, bd = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
  rows = np.random.RandomState(seed).choice(N, min(n_sample, N), replace=False)
  a64, b64 = a.astype(np.float64), b.astype(np.float64)
  mism, r1 = 0, None
  for q, g, q64, g64 in [(ad, bd, a64, b64), (bd, ad, b64, a64)]:
    rank, top1 = ops.sim_rank(q, g)
    rank, top1 = rank.cpu().numpy(), top1.cpu().numpy()
    d = q64[rows] @ g64.T
    dii = d[np.arange(len(rows)), rows]
    want = (d > dii[:, None]).sum(1)
    mism += int((rank[rows] != want).sum()) + int((top1[rows] != d.argmax(1)).sum())
    if r1 is None:
      r1 = 100.0 * float((rank < 1).mean())
  return {'rows': int(len(rows)) * 2, 'mismatches': mism, 'r1_i2t': r1,
          'inputs': 'synthetic.correlated_embeddings(%d, %d, 3.0), both directions, ranks and '
                    'top-1 vs fp64 NumPy' % (N, D)}


TRAIN_CONFIGS = {
    # BASELINE configs[1]: README "HSE tau=0 on ActivityNet with C3D": --low_level_loss --norm
    'anet_c3d_tau0': dict(img_dim=500, feat='normal', vocab=13058, dataset='anet',
                          flags=dict(low_level_loss=True, norm=True),
                          baseline='configs[1]: ActivityNet C3D (img_dim=500, embed=1024) HSE tau=0, batch 32'),
    # the metric's model (ICEP dims) with configs[1]'s losses: round 2's train_step leg, kept for continuity
    'anet_icep_tau0': dict(img_dim=2048, feat='relu', vocab=13058, dataset='anet',
                           flags=dict(low_level_loss=True, norm=True),
                           baseline='ICEP dims with configs[1] losses (no reconstruction)'),
    # BASELINE configs[2]
    'anet_icep_recon': dict(img_dim=2048, feat='relu', vocab=13058, dataset='anet',
                            flags=dict(low_level_loss=True, norm=True, reconstruct_loss=True,
                                       weight_recon=5e-4),
                            baseline='configs[2]: ActivityNet ICEP HSE tau=5e-4 --low_level_loss '
                                     '--reconstruct_loss --norm, batch 32'),
    # BASELINE configs[3]: DiDeMo ICEP, tau = 5e-4 (same flags as configs[2]; all-80-frame clips,
    # short sentences, vocab 7205)
    'didemo_icep_recon': dict(img_dim=2048, feat='relu', vocab=7205, dataset='didemo',
                              flags=dict(low_level_loss=True, norm=True, reconstruct_loss=True,
                                         weight_recon=5e-4),
                              baseline='configs[3]: DiDeMo ICEP HSE tau=5e-4, batch 32'),
}


def train_step_work(batch, img_dim, H, flags, attention, word_dim=300):
  """Algorithmic FLOPs of one VSE.train_emb step on `batch` (model.py:309-369) and the number of
  DEPENDENT GRU steps on its longest tower (each a kernel launch that cannot start before the
  previous one has finished: the latency floor of a small-batch step).

  Per packed (sequence, step) row of an encoder with input width I:
    forward   2*3H*(I+H) + 14H            (+ 2H^2 + 6H for the attention projection)
    backward  2*3H*H   dh_{t-1} = dGh . W_hh               (the BPTT chain)
              2*3H*I   dW_ih += dGx^T x,   2*3H*H  dW_hh += dGh^T h_{t-1}
              2*3H*I   dx = dGx . W_ih     (only where the input needs a gradient: level 2, the
                                            word-embedding table, the decoders)
              4H^2     attention: dW_lin += du^T h and dpool += du . W_lin
  A decoder's input is constant over its steps (model.py:261-265): its input projection and dx are
  counted once per SEQUENCE."""
  lc, lw, lv, lp = (np.asarray(batch[i]) for i in (4, 5, 6, 7))
  n_clips, B = len(lc), len(lv)
  att = (2.0 * H * H + 6.0 * H) if attention else 0.0
  att_b = 4.0 * H * H if attention else 0.0

  def enc(rows, I, need_dx, seqs=None):
    if seqs is None:    # ordinary input
      fwd = rows * (6.0 * H * (I + H) + 14.0 * H + att)
      bwd = rows * (6.0 * H * H + 6.0 * H * I + 6.0 * H * H + (6.0 * H * I if need_dx else 0.0) + att_b)
    else:               # time-constant input: projection / dW_ih / dx once per sequence
      fwd = rows * (6.0 * H * H + 14.0 * H) + seqs * 6.0 * H * I
      bwd = rows * (12.0 * H * H) + seqs * 12.0 * H * I
    return fwd, bwd
  parts = [enc(float(lc.sum() + lv.sum()), img_dim, False),        # clip_enc on clips + whole videos
           enc(float(lw.sum() + lp.sum()), word_dim, True),        # txt_enc (+ d embedding table)
           enc(float(n_clips), H, True), enc(float(n_clips), H, True)]   # level 2, h0 = context
  vis_chain = int(max(lc.max(), lv.max())) + int(max(batch[8]))
  txt_chain = int(max(lw.max(), lp.max())) + int(max(batch[9]))
  if flags.get('reconstruct_loss'):
    attention_saved, att, att_b = att, 0.0, 0.0                      # decoders pool nothing
    parts += [enc(float(n_clips), H, True, seqs=B), enc(float(n_clips), H, True, seqs=B)]
    att = attention_saved
    vis_chain += int(max(batch[8]))
    txt_chain += int(max(batch[9]))
  fwd = sum(p[0] for p in parts)
  bwd = sum(p[1] for p in parts)
  return fwd, bwd, 2 * max(vis_chain, txt_chain)      # forward + backward launches of that tower


def train_bench(name, embed, rnn_type, n_steps, device):
  """One BASELINE training configuration as driver-timed VSE.train_emb steps (model.py:309-369:
  forward, 4-7 contrastive (+2 reconstruction) losses, backward, Adam) on batch-32 loader batches
  of an ActivityNet- / DiDeMo-shaped split — forward and backward on the HIP path.  Priced with
  the step's algorithmic FLOPs against the exact-fp32 MFMA peak and with its dependent-step count."""
  from cmhse_amd.evaluation import LogCollector
  cfg = TRAIN_CONFIGS[name]
  wl = dict(batch=32, img_dim=cfg['img_dim'], vocab=cfg['vocab'], feat=cfg['feat'])
  opt = make_opt(wl, rnn_type, embed)
  for k, v in cfg['flags'].items():
    setattr(opt, k, v)
  torch.manual_seed(1)
  model = VSE(opt)
  model.logger = LogCollector()
  model.train_start(opt)
  # the first loader batches of the same val-shaped split the validation pass runs on (round 2's
  # train_step leg used exactly these, so the figures compare across rounds)
  spec = synthetic.anet_like_spec(1004 if cfg['dataset'] == 'didemo' else 4917, seed=0,
                                  dataset=cfg['dataset'])
  gen = torch.Generator(device=device)
  batches, clip_pos = [], 0
  n_batches = max(1, min(n_steps, 10))
  for bi, b0 in enumerate(range(0, n_batches * wl['batch'], wl['batch'])):
    gen.manual_seed(bi)       # (build_loader's seeding: the same batches as the validation split's)
    b1 = min(spec.n_videos, b0 + wl['batch'])
    batches.append(device_batch(spec, b0, b1, clip_pos, wl['img_dim'], wl['vocab'], wl['feat'], gen,
                                device))
    clip_pos += sum(spec.num_clips[b0:b1])
  # steady state: every batch shape of the timed steps has been seen once (the caching allocator
  # and the event pools grow on first sight of a shape — tens of ms that belong to start-up)
  # (two warm-up rounds: building the model left the GPU idle for a second, and a round of ten
  # steps is too short to bring it back to its working clocks)
  use = [batches[i % len(batches)] for i in range(n_steps)]

  def timed(loader_of):
    """ms per step over the LAST `n_steps` steps of ONE loop `for b in loader: train_emb(opt, *b)`
    (what train.py:185-193 runs) over three rounds of the batches: the first two rounds are the
    warm-up, and a loader that looks ahead (DevicePrefetcher) is in its steady state when the clock
    starts — an epoch is hundreds of steps, not ten."""
    it = iter(loader_of(3))
    for _ in range(2 * n_steps):
      model.train_emb(opt, *next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in it:
      model.train_emb(opt, *b)
    str(model.logger)          # a reader: the last step's loss values have reached the host meters
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_steps
  dt = timed(lambda r: use * r)
  # PCIe-inclusive twins: the SAME steps fed from pinned host memory, as the reference's loader
  # hands batches over (activity_net/data.py:157-162 pin_memory=True; model.py:225-227 uploads inside
  # the step) — the padded 12-tuples of its own collate_fn, the larger of the two host forms
  from cmhse_amd import collate
  host = [tuple(t.cpu().pin_memory() if isinstance(t, torch.Tensor) and t.is_cuda else t for t in b)
          for b in batches]
  host_use = [host[i % len(host)] for i in range(n_steps)]
  host_bytes = float(np.mean([sum(t.numel() * t.element_size() for t in b[:4]) for b in host_use]))
  dt_auto = timed(lambda r: host_use * r)
  dt_prefetch = timed(lambda r: collate.DevicePrefetcher(host_use * r, prepare=model.prepare_batch))
  work = [train_step_work(b, wl['img_dim'], embed, cfg['flags'], rnn_type == 'attention') for b in use]
  fwd = float(np.mean([w[0] for w in work]))
  bwd = float(np.mean([w[1] for w in work]))
  chain = float(np.mean([w[2] for w in work]))
  mfma_floor_ms = (fwd + bwd) / (FP32_MFMA_PEAK_TFLOPS * 1e12) * 1e3
  step_us = measured_step_latency_us()       # (forward, backward) us per dependent launch
  chain_floor_ms = chain * 0.5 * (step_us[0] + step_us[1]) * 1e-3 if step_us else None
  floor = max(mfma_floor_ms, chain_floor_ms or 0.0)
  losses = {k: float(m.val) for k, m in model.logger.meters.items() if k.startswith('Le')}
  del model
  torch.cuda.empty_cache()
  return {'config': cfg['baseline'],
          'flags': ' '.join('--%s' % k if v is True else '--%s %g' % (k, v)
                            for k, v in sorted(cfg['flags'].items())),
          'rnn_type': rnn_type, 'batch': wl['batch'], 'img_dim': wl['img_dim'], 'embed': embed,
          'steps': n_steps, 'ms_per_step': dt * 1e3, 'videos_per_s': wl['batch'] / dt,
          'pcie_inclusive': {
              'ms_per_step': dt_auto * 1e3, 'vs_resident': dt_auto / dt,
              'host_bytes_per_step': host_bytes,
              'feed': 'pinned host 12-tuples (collate_fn form: what DataLoader(pin_memory=True) hands over, '
                      'activity_net/data.py:157-162) straight into train_emb — train.py unchanged: while the host '
                      'runs ahead of the GPU the batch is copied into one of two persistent device slots on the '
                      "copy stream and is resident before its step starts (model.HOST_FEED 'auto')"},
          'pcie_inclusive_prefetcher': {
              'ms_per_step': dt_prefetch * 1e3, 'vs_resident': dt_prefetch / dt,
              'host_bytes_per_step': host_bytes,
              'feed': 'the same batches through collate.DevicePrefetcher(loader, prepare=model.prepare_batch): '
                      'the look-ahead form for a loop that synchronises with the GPU every step'},
          'tflop_per_step': (fwd + bwd) / 1e12, 'tflop_forward': fwd / 1e12,
          'tflop_backward': bwd / 1e12, 'achieved_tflops': (fwd + bwd) / dt / 1e12,
          'dependent_steps': chain,
          'mfma_floor_ms': mfma_floor_ms, 'chain_floor_ms': chain_floor_ms,
          'chain_step_us': step_us,
          'chain_step_us_source': profile_source('r*_step_latency.json'),
          'frac_of_floor': (floor / (dt * 1e3)) if floor > 0 else None,
          # the LOOSER yardstick (it assumes chains and products cannot overlap at all); the bar is
          # frac_of_floor, against max() of the two floors
          'frac_of_sum_of_floors': ((mfma_floor_ms + (chain_floor_ms or 0.0)) / (dt * 1e3)),
          'bound': 'max(FLOPs / 157.3 TFLOP/s fp32 MFMA, dependent steps x the measured latency '
                   'of one small-batch step launch on an idle chip)',
          'last_losses': losses}


def profile_source(pattern):
  """Provenance of a number that bench.py READS from a committed artefact instead of measuring it in
  the run: {"file", "sha256_12", "commit"} of the newest profiles/<pattern>;
  `commit` from profiles/SOURCES.json (tools/profile_sources.py writes it in the build container:
  the GPU box has no .git).  None if no such file."""
  import glob
  import hashlib
  paths = sorted(glob.glob(os.path.join(REPO, 'profiles', pattern)))
  if not paths:
    return None
  rel = os.path.relpath(paths[-1], REPO)
  commits = {}
  try:
    commits = json.load(open(os.path.join(REPO, 'profiles', 'SOURCES.json')))
  except (OSError, ValueError):
    pass
  return {'file': rel, 'sha256_12': hashlib.sha256(open(paths[-1], 'rb').read()).hexdigest()[:12],
          'commit': commits.get(rel), 'measured_in_this_run': False}


def isa_audit():
  """build.audit_isa() of the library this process loaded ({} = clean) plus cmhse_version(): the
  device code holds neither v_pk_fma_f32 nor a double-rate matrix instruction.
  Needs llvm-objdump (present in this image); reported as unavailable otherwise."""
  from cmhse_amd import _lib, build
  if _ISA_AUDIT:
    return dict(_ISA_AUDIT[0])
  out = {'library': os.path.relpath(_lib.LIB_PATH, REPO),
         'version': _lib.load().cmhse_version().decode()}
  try:
    out['forbidden_instructions_found'] = build.audit_isa(_lib.LIB_PATH)
    out['clean'] = not out['forbidden_instructions_found']
    out['checked_for'] = list(build.FORBIDDEN_ISA)
  except Exception as e:      # noqa: BLE001
    out['error'] = '%s: %s' % (type(e).__name__, e)
  _ISA_AUDIT.append(dict(out))
  return out


_ISA_AUDIT = []


def measured_step_latency_us():
  """Latency of ONE dependent small-batch GRU step launch, forward and BPTT (the mid-size step
  kernels at a handful of sequences on an idle chip, launch gap included), from the committed
  sweep of the newest round (profiles/r*_step_latency.json: {"forward_us": x, "backward_us": y}).
  None if absent."""
  import glob
  paths = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_step_latency.json')))
  if not paths:
    return None
  try:
    d = json.load(open(paths[-1]))
    return float(d['forward_us']), float(d['backward_us'])
  except (ValueError, KeyError, OSError):
    return None


def measured_traffic(kernel='gru_step'):
  """Fabric bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
  (profiles/r*_pmc_hbm_traffic.json, newest round: separate --pmc FETCH_SIZE / WRITE_SIZE runs, KiB units,
  FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md §HBM).  None if absent."""
  import glob
  paths = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_pmc_hbm_traffic.json')))
  if not paths:
    return None
  d = json.load(open(paths[-1]))
  tot, n = 0.0, 0
  names = (kernel,) if isinstance(kernel, str) else tuple(kernel)
  for k, v in d.items():
    if any(nm in k for nm in names):
      tot += v['launches'] * (v['hbm_read_bytes_per_launch_corrected'] +
                              v['hbm_write_bytes_per_launch'])
      n += v['launches']
  return tot / n if n else None


def measured_clock_ghz():
  """In-kernel shader clock of the tiled GRU step under load (median over workgroups), from the
  committed tools/tile_trace.py run (profiles/r*_tile_trace.txt, newest round).  None if absent."""
  import glob
  paths = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_tile_trace.txt')))
  if not paths:
    return None
  for line in open(paths[-1]):
    if line.startswith('in-kernel shader clock') and 'median' in line:
      try:
        return float(line.split('median')[1].split('GHz')[0])
      except ValueError:
        return None
  return None



def dropin_validate_bench(opt, model, loaders, n_steps, crc_of):
  """The pass as train.py runs it (train.py:223-236): `from evaluation import encode_data, i2t, t2i`
  through the drop-in shims (cmhse_amd/dropin first on sys.path), encode_data -> six NumPy matrices
  -> i2t(vid, para) -> t2i(vid, para) -> currscore, for every loader in `loaders` ({name: list of
  12-tuples}: the split resident in HBM, and in pinned host memory as the reference's DataLoader
  hands it over).  Per loader: ms per pass with the package's defaults (page-locked staging under the
  level-2 encoders, the ranking queued by encode_data and served to i2t / t2i after a content check
  of the arrays), the same with that reuse switched off ('recompute': i2t / t2i upload the arrays
  and rank again), and the CRC of the integer ranks, which must equal the headline pass's."""
  import importlib
  import logging as pylog
  from cmhse_amd import evaluation as core
  dropin = os.path.join(REPO, 'cmhse_amd', 'dropin')
  sys.path.insert(0, dropin)
  try:
    ev = importlib.import_module('evaluation')
  finally:
    sys.path.remove(dropin)
  log = pylog.getLogger('cmhse_bench_validate')
  log.addHandler(pylog.NullHandler())
  log.propagate = False
  log_step = getattr(opt, 'log_step', 10)

  def validate(val_loader):      # train.py:223-236, verbatim in structure
    vid_seq_embs, para_seq_embs, clip_embs, cap_embs, _, _, num_clips, cur_vid_total = ev.encode_data(
        opt, model, val_loader, log_step, log.info, contextual_model=True)
    vid_seq_rep, top1_v2p, rank_vid_v2p = ev.i2t(vid_seq_embs, para_seq_embs, measure=opt.measure)
    para_seq_rep, top1_p2v, rank_para_p2v = ev.t2i(vid_seq_embs, para_seq_embs, measure=opt.measure)
    return vid_seq_rep['sum'] + para_seq_rep['sum'], rank_vid_v2p, rank_para_p2v

  def timed(val_loader, steps):
    validate(val_loader)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
      score, ri, rt = validate(val_loader)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, score, crc_of(ri, rt)

  out = {}
  for name, loader in loaders.items():
    if loader is None:
      continue
    hits0 = dict(core.CACHE_STATS)
    ms, score, crc = timed(loader, n_steps)
    hits = core.CACHE_STATS['hits'] - hits0['hits']
    core.SPECULATE_RANKS[0] = False
    try:
      ms_re, score_re, crc_re = timed(loader, max(2, n_steps // 2))
    finally:
      core.SPECULATE_RANKS[0] = True
    out[name] = {'steps': n_steps, 'ms_per_step': ms, 'currscore': float(score), 'ranks_crc32': crc,
                 'served_from_encode_data': hits, 'calls': 2 * (n_steps + 1),
                 'recompute': {'ms_per_step': ms_re, 'ranks_crc32': crc_re,
                               'note': 'SPECULATE_RANKS off: i2t / t2i upload their NumPy arguments '
                                       'and rank again'}}
  out['api'] = ('cmhse_amd/dropin: evaluation.encode_data -> NumPy 8-tuple -> evaluation.i2t -> evaluation.t2i '
                '(the body of train.py:223-236)')
  out['note'] = ('every timed pass encodes the split and ranks both directions ONCE, inside encode_data; i2t / t2i of the '
                 'same pass report that ranking after comparing the arrays they are handed with the device copies '
                 '(nothing is carried from one pass to the next: the next encode_data replaces the entry); `recompute` '
                 'is the same pass with i2t / t2i uploading their arguments and ranking again')
  return out
