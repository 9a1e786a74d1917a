#!/usr/bin/env python3
"""Mutation check of the parity tests: do they notice a kernel that drops a reference quirk?

  python tools/mutation_check.py build [name ...]   (build container: hipcc, no GPU needed)
      copies cmhse_amd/csrc to build/mutants/<name>/, applies ONE source edit per mutant, and builds
      build/mutants/libcmhse_<name>.so (same ABI; the product sources and library are untouched).
  python tools/mutation_check.py run          (GPU box)
      runs the selected GPU tests once per mutant with CMHSE_HIP_LIB pointing at it and reports
      which tests FAILED — a mutant that no test kills is a hole in the suite.  Also runs the same
      selection on the product library (must pass).

Mutants:
  attn_eps_fwd   gru_attention.hpp: the masked softmax's `+ 0.0001f` (layers.py:158-162) -> `+ 0.0f`
  attn_eps_bwd   bwd_pool_kernels.hpp: the same epsilon in the attention backward
  norm_by_n, gru_bhn_outside_r(_tiled), rank_counts_ties, l2norm_eps   (round 6, second batch) see MUTANTS
"""
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, 'build', 'mutants')

MUTANTS = {
    'attn_eps_fwd': ('gru_attention.hpp', 's_den = d + 0.0001f;', 's_den = d + 0.0f;'),
    'attn_eps_bwd': ('bwd_pool_kernels.hpp', 's_den = d + 0.0001f;', 's_den = d + 0.0f;'),
    # loss.py:114-115: `norm` divides by n * m, not by n
    'norm_by_n': ('sim.hip', 'loss = loss / static_cast<float>(static_cast<int64_t>(p.n) * p.n);',
                  'loss = loss / static_cast<float>(p.n);'),
    # nn.GRU: n = tanh(W_in x + b_in + r * (W_hn h + b_hn)) — b_hn INSIDE the reset gate's product
    # (the mid-size step kernel, the one the golden-sized batches run on)
    'gru_bhn_outside_r': ('gru_small_batch.hpp', 'const float ng = tanhf_(e_gx[q][2] + e_b[q][2] + rg * ghn);',
                          'const float ng = tanhf_(e_gx[q][2] + e_b[q][2] + rg * hn_ + e_b[q][3]);'),
    # ... and the LDS-tiled step (batches of more than 1024 sequences)
    'gru_bhn_outside_r_tiled': ('gru_step_tile.hpp', 'const float ng = tanhf_(acc[ms][2][r] + b_in + rg * ghn);',
                                'const float ng = tanhf_(acc[ms][2][r] + b_in + rg * acc[ms][3][r] + b_hn);'),
    # evaluation.py:92-95 / 131-134 via the documented tie rule: strict '>' for the rank
    'rank_counts_ties': ('sim.hip', 'valid && j != gi && v > dii', 'valid && j != gi && v >= dii'),
    # loss.py:100-103: the diagonal of both cost matrices is masked to zero
    'loss_diag_not_masked': ('sim.hip', 'if (j == i) c = 0.f;', ''),
    # loss.py:106-111: max_violation keeps the hardest negative per row / column only
    'max_violation_sums': ('sim.hip', 'total += p.max_violation ? static_cast<double>(mx) : sum;', 'total += sum;'),
    # F.normalize's eps (model.py l2norm of the encoder outputs): a zero row stays zero, x / max(|x|, 1e-12)
    'l2norm_eps': ('gru_rows.hpp', 's_inv = 1.0f / fmaxf(sqrtf(t), 1e-12f);', 's_inv = 1.0f / sqrtf(t);'),
}
# the tests that must kill them (and pass on the product library)
SELECT = ['tests/test_quirks_tight.py', 'tests/test_gpu_golden.py', 'tests/test_gpu_gru.py', 'tests/test_gpu_scoring.py', '-k',
          'epsilon or test_layers_vs_golden or test_layer_backward_vs_golden or test_model_vs_golden or norm or '
          'test_gru_pool_vs_oracle or tie or zero or l2norm or test_contrastive_vs_oracle']


def build(only=()):
  from cmhse_amd import build as b
  os.makedirs(OUT, exist_ok=True)
  for name, (fname, old, new) in MUTANTS.items():
    if only and name not in only:
      continue
    root = os.path.join(OUT, name)
    shutil.rmtree(root, ignore_errors=True)
    src = os.path.join(root, 'cmhse_amd', 'csrc')          # (the sources include ../../include/cmhse_hip.h)
    shutil.copytree(b.CSRC, src)
    shutil.copytree(os.path.join(REPO, 'include'), os.path.join(root, 'include'))
    text = open(os.path.join(src, fname)).read()
    assert text.count(old) == 1, (name, text.count(old))
    open(os.path.join(src, fname), 'w').write(text.replace(old, new))
    target = os.path.join(OUT, 'libcmhse_%s.so' % name)
    cmd = [b._hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared']
    cmd += list(b.DEVICE_FLAGS) + ['-o', target]
    cmd += [os.path.join(src, s) for s in b.SOURCES]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
      raise SystemExit('hipcc failed for %s:\n%s' % (name, res.stdout[-3000:]))
    shutil.rmtree(root)
    print('built', target)


def run():
  results = {}
  for name in [None] + sorted(MUTANTS):
    env = dict(os.environ)
    if name:
      env['CMHSE_HIP_LIB'] = os.path.join(OUT, 'libcmhse_%s.so' % name)
    res = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
                          '--tb=line'] + SELECT, cwd=REPO, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    failed = [l for l in res.stdout.splitlines() if l.startswith('FAILED')]
    errors = [l for l in res.stdout.splitlines() if l.startswith('ERROR')]
    tail = res.stdout.strip().splitlines()[-1] if res.stdout.strip() else ''
    if errors:      # a mutant that does not even load (stale build, ABI drift) has killed nothing
      print('=== %s: %d collection / setup ERRORS — rebuild the mutants (python tools/mutation_check.py build)' % (name, len(errors)))
      failed = []
    results[name or 'product'] = (res.returncode, failed, tail)
    print('=== %s: rc %d  %s' % (name or 'product library', res.returncode, tail))
    for l in failed:
      print('   ', l[:240])
    for l in res.stdout.splitlines():
      if 'Max absolute difference' in l or 'Mismatched elements' in l:
        print('      ', l.strip())
  ok = results['product'][0] == 0 and all(results[m][1] for m in MUTANTS)
  print('mutation check:', 'every mutant killed, product green' if ok else 'HOLE: see above')
  return 0 if ok else 1


if __name__ == '__main__':
  sys.exit(build(sys.argv[2:]) if sys.argv[1:2] == ['build'] else run())
