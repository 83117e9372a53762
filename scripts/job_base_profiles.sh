set -e
cd $GRAFT_REPO_ROOT
RT_PROFILE_ARGS="--scene glass_spheres --width 1920 --height 1080 --spp 512 --bounces 32 --lanes-per-pixel 4 --compact" bash scripts/profile_gpu.sh r02_base_glass
RT_PROFILE_ARGS="--scene random_spheres --count 4096 --width 1920 --height 1080 --spp 64 --lanes-per-pixel 4 --compact" bash scripts/profile_gpu.sh r02_base_s4096
timeout -k 10 200 python bench.py > gpurun_out/r02_base_bench.json 2> gpurun_out/r02_base_bench.err
cat gpurun_out/r02_base_bench.json
