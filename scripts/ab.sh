#!/bin/bash
# A/B on one GPU box: bench.py with the in-tree library and with each library variant given as
# argument (build/*.so), alternating, three rounds.  Prints kernel milliseconds per run.
# CONFIG=<bench config> (default cornell); BENCH_FLAGS="..." extra bench.py flags for every run.
cd ${GRAFT_REPO_ROOT:-/root/repo}
CONFIG=${CONFIG:-cornell}
one() { timeout -k 10 200 python bench.py --config $CONFIG --cpu-spp 0 ${BENCH_FLAGS} $2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$CONFIG', '$1', '$2', d['roofline']['kernel_ms'], d['value'])"; }
for round in 1 2 3; do
  one tree || exit 1
  if [ -n "$TREE_ALT_FLAG" ]; then one tree "$TREE_ALT_FLAG" || exit 1; fi
  for lib in "$@"; do RT_HIP_LIBRARY=$PWD/$lib one $lib || exit 1; done
done
