"""Builds libcmhse_hip.so (the C-ABI HIP library, include/cmhse_hip.h) in-tree for gfx950.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
repo snapshot.  `python -m cmhse_amd.build` or `cmhse_amd.build.build()`.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libcmhse_hip.so')
SOURCES = ['gru.hip', 'sim.hip', 'bwd.hip']
HEADERS = [os.path.join(CSRC, 'nt_core.hpp'), os.path.join(CSRC, 'tn_core.hpp'),
           os.path.join(CSRC, 'tn_rows.hpp'), os.path.join(CSRC, 'step_loss.hpp'),
           os.path.join(CSRC, 'gru_ws.hpp'), os.path.join(CSRC, 'grid_sync.hpp'),
           os.path.join(os.path.dirname(HERE), 'include', 'cmhse_hip.h')]


def _hipcc():
  for cand in [shutil.which('hipcc'), '/opt/rocm/bin/hipcc']:
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError('hipcc not found: cannot build libcmhse_hip.so')


def is_stale():
  if not os.path.exists(LIB):
    return True
  t = os.path.getmtime(LIB)
  deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
  return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), out=None):
  """Build the product library (default) or, with `out` / `extra_flags`, another build of the same
  sources at another path (tools/: instrumented builds; load it through CMHSE_HIP_LIB)."""
  global LIB
  if out is None and not extra_flags and not force and not is_stale():
    return LIB
  target = out or LIB
  cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
         '-o', target + '.tmp'] + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES]
  if verbose:
    cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
  res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
  if res.returncode != 0:
    raise RuntimeError('hipcc failed:\n' + res.stdout)
  if verbose:
    print(res.stdout)
  os.replace(target + '.tmp', target)
  return target


if __name__ == '__main__':
  print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
