# Runs on the GPU box: Cornell bench render over hand-over bounds, lanes per pixel and workgroup sizes (experiment build).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export RT_HIP_LIBRARY=$PWD/build/librt_hip_knobs.so
for S in 4 8 12 20; do for T in 16 32 48; do
  RT_POOL_MAX_SAMPLES=$S RT_POOL_THRESHOLD=$T timeout -k 10 120 python scripts/sweep_compaction.py 2,4,8 || exit 1
done; done
for bt in 64 128 512; do echo "block $bt"; RT_BLOCK_THREADS=$bt timeout -k 10 120 python scripts/sweep_compaction.py 4 || exit 1; done
for w in 8 16 32; do echo "stage2 waves $w"; RT_STAGE2_WAVES_PER_CU=$w timeout -k 10 120 python scripts/sweep_compaction.py 4 || exit 1; done
