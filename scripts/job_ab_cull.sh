# Runs on the GPU box: GPU test suite, then bench.py with and without the primary-ray cull on one box.
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 900 python -u -m pytest tests -x -q --timeout 600 --timeout-method thread -m gpu > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
for c in ${CONFIG_LIST:-cornell}; do
  for flag in "" "--no-cull" "" "--no-cull"; do
    timeout -k 10 200 python bench.py --config $c --cpu-spp 0 --steps 10 --warmup 2 $flag > gpurun_out/ab_cull.json 2> gpurun_out/ab_cull.err || { tail -20 gpurun_out/ab_cull.err; exit 1; }
    python - "$c" "$flag" <<PY
import json, sys
d = json.loads(open("gpurun_out/ab_cull.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:12s} {sys.argv[2]:10s} {d['value']:10.1f} Mrays/s {d['ms_per_step']:9.3f} ms kernel {d['roofline']['kernel_ms']} frac {d['roofline']['frac']:.4f} rays {d['config']['rays_per_step']}")
PY
  done
done
