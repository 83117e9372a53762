# Runs on the one-GPU box: bench.py's multi-rank path with 2 and 4 ranks sharing GPU 0 over gloo (both shard modes).
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
for args in "--gpus 2" "--gpus 4" "--gpus 4 --shard rows" "--gpus 2 --config glass --spp 64"; do
  echo "== bench.py $args --rehearse-on-one-gpu"
  timeout -k 10 280 python bench.py $args --rehearse-on-one-gpu --steps 3 --warmup 1 > gpurun_out/rehearse.json 2> gpurun_out/rehearse.err || { tail -30 gpurun_out/rehearse.err; exit 1; }
  cat gpurun_out/rehearse.json
done
