# Runs on the GPU box: does the compaction pay at few samples per lane?  Cornell / glass / spheres at reduced spp, compacted vs single launch.
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { timeout -k 10 250 python bench.py --cpu-spp 0 --steps 5 --warmup 1 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['config']['compact_tails'])" || exit 1; }
for cfg in cornell glass spheres; do for spp in 32 64 128; do
  one --config $cfg --spp $spp --lanes-per-pixel 4
  one --config $cfg --spp $spp --lanes-per-pixel 4 --no-compact
done; done
for l in 2 4 8; do one --config spheres4096 --lanes-per-pixel $l --steps 2; done
