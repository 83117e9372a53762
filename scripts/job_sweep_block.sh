# Runs on the GPU box: bench.py on all configurations at workgroup sizes 64 / 128 / 256 (RT_BLOCK_THREADS).
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
for bt in 64 128 256; do
  echo "== RT_BLOCK_THREADS=$bt"
  RT_BLOCK_THREADS=$bt STEPS=3 bash scripts/job_configs.sh bt$bt
done
