# Runs on the GPU box: bench.py on all configurations at workgroup sizes 64 / 128 / 256 (RT_BLOCK_THREADS).
# The RT_* launch knobs exist only in the experiment build: make -C raytracer_amd/csrc knobs, then RT_HIP_LIBRARY=$PWD/build/librt_hip_knobs.so
export RT_HIP_LIBRARY=${RT_HIP_LIBRARY:-${GRAFT_REPO_ROOT:-/root/repo}/build/librt_hip_knobs.so}
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
for bt in 64 128 256; do
  echo "== RT_BLOCK_THREADS=$bt"
  RT_BLOCK_THREADS=$bt STEPS=3 bash scripts/job_configs.sh bt$bt
done
