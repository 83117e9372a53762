# Runs on the GPU box: the GPU test suite, then bench.py as the driver calls it (N = 1 through the launcher path too).
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 900 python -u -m pytest tests -x -q --timeout 600 --timeout-method thread -m gpu > gpurun_out/gputests.log 2>&1 || { tail -30 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -20 gpurun_out/bench_default.err; exit 1; }
cat gpurun_out/bench_default.json
