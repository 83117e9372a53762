# Runs on the GPU box: kernel stats + VALU counters for one driver render.  usage: job_prof_one.sh <tag> <driver args...>
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
BIN=$REPO/raytracer_amd/bin/raytracer
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BIN "$@" --out /tmp/p.bmp > $OUT/stats.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_valu -o pmc_valu -- $BIN "$@" --out /tmp/p.bmp > $OUT/pmc_valu.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc_wait -o pmc_wait -- $BIN "$@" --out /tmp/p.bmp > $OUT/pmc_wait.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -o pmc_grbm -- $BIN "$@" --out /tmp/p.bmp > $OUT/pmc_grbm.log 2>&1
grep -E "Total|Perf" $OUT/stats.log
