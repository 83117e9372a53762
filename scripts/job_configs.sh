# Runs on the GPU box: bench.py for every configuration (no CPU baseline), one JSON line each into gpurun_out/configs_<tag>.jsonl
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-now}
OUT=gpurun_out/configs_$TAG.jsonl
: > $OUT
for c in ${CONFIG_LIST:-cornell spheres spheres4096 glass}; do
  timeout -k 10 200 python bench.py --config $c --cpu-spp 0 --steps ${STEPS:-5} --warmup 1 >> $OUT 2> gpurun_out/configs_$TAG.err || { tail -20 gpurun_out/configs_$TAG.err; exit 1; }
done
python - <<PY
import json
for line in open("$OUT"):
    d = json.loads(line)
    print(f"{d['config']['name']:12s} {d['value']:10.1f} Mrays/s  {d['ms_per_step']:9.3f} ms  frac {d['roofline']['frac']:.4f}  lanes {d['config']['lanes_per_pixel']} compact {d['config']['compact_tails']}")
PY
