#!/bin/bash
# A/B on one GPU box: bench.py with the in-tree library and with each library variant given as
# argument (build/*.so), alternating, three rounds.  Prints kernel milliseconds per run.
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { timeout -k 10 100 python bench.py --cpu-spp 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['roofline']['kernel_ms'])"; }
for round in 1 2 3; do
  one tree || exit 1
  for lib in "$@"; do RT_HIP_LIBRARY=$PWD/$lib one $lib || exit 1; done
done
