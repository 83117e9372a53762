# Runs on the GPU box: the round-3 profile set.  Kernel-trace statistics and counters (one --pmc pass per group) for
# every benchmark configuration (native driver, one cold render each, the launch sequence bench.py times), the kernel
# statistics and HBM counters of the bench command itself, and the diagnostic build's breakdown (cull statistics).
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
RT_PROFILE_ARGS="--scene cornell --width 1920 --height 1080 --spp 256 --lanes-per-pixel 4 --compact" bash scripts/profile_gpu.sh r03_cornell > /dev/null
RT_PROFILE_ARGS="--scene glass_spheres --width 1920 --height 1080 --spp 512 --bounces 32 --lanes-per-pixel 8 --compact" bash scripts/profile_gpu.sh r03_glass > /dev/null
RT_PROFILE_ARGS="--scene random_spheres --count 4096 --width 1920 --height 1080 --spp 64 --lanes-per-pixel 4 --compact" bash scripts/profile_gpu.sh r03_spheres4096 > /dev/null
RT_PROFILE_ARGS="--scene spheres --width 1280 --height 720 --spp 512 --lanes-per-pixel 8 --compact" bash scripts/profile_gpu.sh r03_spheres > /dev/null
RT_PROFILE_ARGS="--scene cornell --width 1920 --height 1080 --spp 1024 --lanes-per-pixel 4 --compact" bash scripts/profile_gpu.sh r03_cornell1024 > /dev/null
bash scripts/profile_bench.sh r03_bench
RT_HIP_LIBRARY=$PWD/build/librt_hip_stamps.so timeout -k 10 300 python scripts/stamp_breakdown.py cornell spheres glass > gpurun_out/r03_stamp_breakdown.txt
RT_HIP_LIBRARY=$PWD/build/librt_hip_stamps.so timeout -k 10 300 python scripts/stamp_breakdown.py cornell --no-cull >> gpurun_out/r03_stamp_breakdown.txt
tail -5 gpurun_out/r03_stamp_breakdown.txt
