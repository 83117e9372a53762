# Runs on the one-GPU box: the multi-rank code path of bench.py through RCCL with a single rank (process group
# on the GPU, gather / all_gather, fixed-order sum, resolve, composition check), both shard modes.
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
for shard in samples rows; do
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 5 --warmup 1 --through-process-group --shard $shard --cpu-spp 0 $EXTRA | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$shard', d['value'], d['ms_per_step'], d['config']['workload'], d.get('composition_check'))"
done
