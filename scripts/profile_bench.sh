#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace statistics of the bench command itself, then the
# HBM-traffic counters (one pass each) of the same command, then a host-inclusive timing.
set -u
TAG=${1:-bench_prof}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 10 --warmup 2 --cpu-spp 0"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1 || { echo stats failed; tail -5 $OUT/stats.log; exit 1; }
grep '^{' $OUT/stats.log > $OUT/bench_under_rocprof.json
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || { echo fetch failed; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || { echo write failed; tail -5 $OUT/pmc_write.log; exit 1; }
cd $REPO
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
timeout -k 10 120 python3 - <<'PY'
import time, numpy as np
from raytracer_amd import binding as B
from raytracer_amd.host_scene import build_scene
ctx = B.Context(0); ctx.upload_world(build_scene("cornell"))
p = B.default_params(1920, 1080, 256, lanes_per_pixel=4, flags=B.FLAG_COMPACT_TAILS)
ctx.render(p, want_radiance=False)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); pix, _, rays = ctx.render(p, want_radiance=False); ts.append(time.perf_counter() - t0)
print("host-inclusive rt_render (pixels to host memory): best %.3f ms, %.1f Mrays/s" % (min(ts) * 1e3, rays / min(ts) / 1e6))
PY
