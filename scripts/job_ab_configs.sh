# usage: job_ab_configs.sh <lib or "tree"> ... : bench.py on the sphere configurations with each library
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  echo "== $lib"
  if [ "$lib" = tree ]; then CONFIG_LIST="${CONFIG_LIST:-spheres spheres4096 glass}" STEPS=3 bash scripts/job_configs.sh ab
  else RT_HIP_LIBRARY=$PWD/$lib CONFIG_LIST="${CONFIG_LIST:-spheres spheres4096 glass}" STEPS=3 bash scripts/job_configs.sh ab; fi
done
