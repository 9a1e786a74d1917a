"""A/B arms for the tools: an arm is "k=v,k=v" with keys
  tune.<name>          a cmhse_tune crossover (tiny_max_seqs, mid_max_seqs, mid_units, mid_waves,
                       tall_tile_min_wgs, bwd_mid_max_seqs)
  schedule             model.TRAIN_SCHEDULE  (interleaved | levels | towers | grouped | serial)
  side_streams         ops.SIDE_STREAMS      (0 | 1)
  batched_losses       model.BATCHED_LOSSES  (0 | 1)
  fused_losses         model.FUSED_LOSSES    (0 | 1)
  early_pool, group_towers, two_streams, pipeline_upload, upload_chunk   evaluation.* flags
  math                 ops.set_math_mode     (fp32 | bf16x3)
Everything an arm does not mention is put back to its default."""
from cmhse_amd import evaluation, model as model_mod, ops

TUNE_DEFAULTS = dict(tiny_max_seqs=1024, mid_max_seqs=1024, mid_units=0, mid_waves=0,
                     tall_tile_min_wgs=2048, bwd_mid_max_seqs=512, bwd_split_min_seqs=33, bwd_tail_min_steps=4, bwd_chunk_rows=2048,
                     fwd_tail_min_steps=4, mid_tall_min_seqs=129, xproj_chunk_rows=1536, tn_rows_bm=0, chain_min_steps=2, chain_tall_min_wgs=256, early_xproj=1)


def parse(spec):
  return [dict(kv.split('=') for kv in a.split(',') if kv) for a in spec.split(';')]


def apply(arm):
  for k, v in TUNE_DEFAULTS.items():
    ops.tune(k, int(arm.get('tune.' + k, v)))
  model_mod.TRAIN_SCHEDULE[0] = arm.get('schedule', 'interleaved')
  model_mod.BATCHED_LOSSES[0] = arm.get('batched_losses', '1') == '1'
  model_mod.FUSED_LOSSES[0] = arm.get('fused_losses', '1') == '1'
  ops.SIDE_STREAMS[0] = arm.get('side_streams', '1') == '1'
  evaluation.EARLY_POOL[0] = arm.get('early_pool', '1') == '1'
  evaluation.GROUP_TOWERS[0] = arm.get('group_towers', '1') == '1'
  evaluation.TWO_STREAMS[0] = arm.get('two_streams', '0') == '1'
  evaluation.PIPELINE_UPLOAD[0] = arm.get('pipeline_upload', '1') == '1'
  evaluation.UPLOAD_CHUNK[0] = int(arm.get('upload_chunk', '8'))
  ops.set_math_mode(arm.get('math', 'fp32'))
  unknown = [k for k in arm if not (k.startswith('tune.') and k[5:] in TUNE_DEFAULTS) and k not in
             ('schedule', 'batched_losses', 'fused_losses', 'side_streams', 'early_pool', 'group_towers', 'two_streams',
              'pipeline_upload', 'upload_chunk', 'math', 'label')]
  if unknown:
    raise SystemExit('unknown arm keys: %s' % unknown)


def label(arm):
  return ','.join('%s=%s' % kv for kv in arm.items()) or '(default)'
