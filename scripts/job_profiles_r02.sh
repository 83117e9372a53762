# Runs on the GPU box: the round-2 profile set.  Kernel-trace statistics and counters for every benchmark
# configuration (native driver, one render each), then the kernel statistics and HBM counters of bench.py itself.
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
RT_PROFILE_ARGS="--scene cornell --width 1920 --height 1080 --spp 256 --lanes-per-pixel 4 --compact" bash scripts/profile_gpu.sh r02_cornell > /dev/null
RT_PROFILE_ARGS="--scene glass_spheres --width 1920 --height 1080 --spp 512 --bounces 32 --lanes-per-pixel 8 --compact" bash scripts/profile_gpu.sh r02_glass > /dev/null
RT_PROFILE_ARGS="--scene random_spheres --count 4096 --width 1920 --height 1080 --spp 64 --lanes-per-pixel 4 --compact" bash scripts/profile_gpu.sh r02_spheres4096 > /dev/null
RT_PROFILE_ARGS="--scene spheres --width 1280 --height 720 --spp 512 --lanes-per-pixel 8 --compact" bash scripts/profile_gpu.sh r02_spheres > /dev/null
bash scripts/profile_bench.sh r02_bench
