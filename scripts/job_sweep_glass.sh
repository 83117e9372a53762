# Runs on the GPU box: one configuration (CFG, default glass) over lanes per pixel and hand-over bounds, compacted and single launch.
# The RT_* launch knobs exist only in the experiment build: make -C raytracer_amd/csrc knobs, then RT_HIP_LIBRARY=$PWD/build/librt_hip_knobs.so
export RT_HIP_LIBRARY=${RT_HIP_LIBRARY:-${GRAFT_REPO_ROOT:-/root/repo}/build/librt_hip_knobs.so}
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" CONFIG_LIST="${CFG:-glass}" STEPS=3 bash scripts/job_configs.sh sweep | tail -1; }
for lanes in 1 2 4 8; do
  echo "-- lanes $lanes"; timeout -k 10 200 python bench.py --config ${CFG:-glass} --cpu-spp 0 --steps 3 --warmup 1 --lanes-per-pixel $lanes | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['compact_tails'])"
done
for t in 16 32 48; do for s in 4 12 32; do
  echo "-- lanes 4 threshold $t max_samples $s"; RT_POOL_THRESHOLD=$t RT_POOL_MAX_SAMPLES=$s timeout -k 10 200 python bench.py --config ${CFG:-glass} --cpu-spp 0 --steps 3 --warmup 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
echo "-- no compact"; timeout -k 10 200 python bench.py --config ${CFG:-glass} --cpu-spp 0 --steps 3 --warmup 1 --no-compact | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
