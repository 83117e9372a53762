cd ${GRAFT_REPO_ROOT:-/root/repo}
export RT_HIP_LIBRARY=$PWD/build/librt_hip_knobs.so
for l in 2 4; do for t in 16 32 48; do for m in 3 7 12 24; do
RT_POOL_THRESHOLD=$t RT_POOL_MAX_SAMPLES=$m timeout -k 10 200 python bench.py --spp 128 --lanes-per-pixel $l --cpu-spp 0 --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes $l thr $t max $m', d['roofline']['kernel_ms'])"
done; done; done
