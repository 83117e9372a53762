#!/bin/bash
# A/B sweep on the GPU box: block size variants (build/librt_hip_bt*.so) and lanes per pixel.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { timeout -k 10 200 python bench.py --steps 5 --warmup 1 --cpu-spp 0 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['config']['lanes_per_pixel'])"; }
for bt in 64 128 512; do echo "block $bt:"; RT_HIP_LIBRARY=$PWD/build/librt_hip_bt$bt.so run || exit 1; done
echo "block 256:"; run || exit 1
for l in 2 4 8; do echo "lanes $l:"; run --lanes-per-pixel $l || exit 1; done
