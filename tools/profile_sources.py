#!/usr/bin/env python3
"""profiles/SOURCES.json: {file under profiles/: short hash of the commit that last touched it}.
bench.py reads a few context values from committed profile artefacts (PMC traffic, in-kernel clock,
step latency) and names their provenance in its JSON line; the GPU box has no .git, so the commits
are looked up here, in the build container, and committed as data.  Run after committing profiles/."""
import json
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
  out = {}
  for name in sorted(os.listdir(os.path.join(REPO, 'profiles'))):
    rel = os.path.join('profiles', name)
    if name == 'SOURCES.json':
      continue
    h = subprocess.run(['git', 'log', '-1', '--format=%h', '--', rel], cwd=REPO, stdout=subprocess.PIPE,
                       text=True).stdout.strip()
    if h:
      out[rel] = h
  json.dump(out, open(os.path.join(REPO, 'profiles', 'SOURCES.json'), 'w'), indent=0, sort_keys=True)
  print('%d files' % len(out))


if __name__ == '__main__':
  main()
