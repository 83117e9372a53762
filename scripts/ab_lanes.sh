#!/bin/bash
# A/B on one GPU box over lanes per pixel: in-tree library against one variant library.
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { timeout -k 10 100 python bench.py --cpu-spp 0 --steps 5 --warmup 1 --lanes-per-pixel $2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'lanes', $2, d['roofline']['kernel_ms'])"; }
for l in 1 2 4 8; do
  one tree $l || exit 1
  RT_HIP_LIBRARY=$PWD/$1 one $1 $l || exit 1
done
