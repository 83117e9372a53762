# Runs on the GPU box: 4096-sphere configuration at several lanes per pixel, single launch and with hand-over bounds that keep the compaction on.
# The RT_* launch knobs exist only in the experiment build: make -C raytracer_amd/csrc knobs, then RT_HIP_LIBRARY=$PWD/build/librt_hip_knobs.so
export RT_HIP_LIBRARY=${RT_HIP_LIBRARY:-${GRAFT_REPO_ROOT:-/root/repo}/build/librt_hip_knobs.so}
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { echo "-- $LANES $*"; env "$@" timeout -k 10 250 python bench.py --config spheres4096 --cpu-spp 0 --steps 2 --warmup 1 $LANES | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1; }
for l in 4 8 16 32; do
  LANES="--lanes-per-pixel $l --no-compact" one A=1
  LANES="--lanes-per-pixel $l" one RT_POOL_MAX_SAMPLES=$((64 / l / 4 - 1))
done
