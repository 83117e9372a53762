#!/bin/bash
# Runs on the GPU box: the other benchmark configurations through the native driver, one cold process
# per line, each printing the reference's report lines.  Output: gpurun_out/configs.log
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=raytracer_amd/bin/raytracer
OUT=gpurun_out/configs.log
: > $OUT
run() { echo "== $*" >> $OUT; timeout -k 10 300 $R "$@" --out /tmp/configs.bmp >> $OUT 2>&1 || { echo "FAILED: $*"; tail -3 $OUT; exit 1; }; }
run --scene spheres --width 1280 --height 720 --spp 512 --lanes-per-pixel 4 --compact
run --scene cornell --width 1920 --height 1080 --spp 256 --lanes-per-pixel 4 --compact
run --scene cornell --width 1920 --height 1080 --spp 1024 --lanes-per-pixel 4 --compact
run --scene cornell --width 3840 --height 2160 --spp 512 --lanes-per-pixel 4 --compact
run --scene random_spheres --count 4096 --width 1920 --height 1080 --spp 64 --lanes-per-pixel 4 --compact
run --scene glass_spheres --width 1920 --height 1080 --spp 512 --bounces 32 --lanes-per-pixel 4 --compact
grep "==\|Performance\|Total" $OUT
