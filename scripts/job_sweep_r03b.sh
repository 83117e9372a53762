# Runs on the GPU box: alternating A/B of a few launch-parameter candidates on the Cornell bench render (experiment build).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export RT_HIP_LIBRARY=$PWD/build/librt_hip_knobs.so
for round in 1 2 3; do
  for v in "A=0" "RT_POOL_THRESHOLD=48" "RT_BLOCK_THREADS=64" "RT_BLOCK_THREADS=128" "RT_BLOCK_THREADS=64 RT_POOL_THRESHOLD=48" "RT_BLOCK_THREADS=128 RT_POOL_THRESHOLD=48" "RT_BLOCK_THREADS=128 RT_POOL_THRESHOLD=40 RT_POOL_MAX_SAMPLES=10"; do
    echo -n "$v : "; env $v timeout -k 10 120 python scripts/sweep_compaction.py 4 | tail -1 || exit 1
  done
done
