#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace statistics, then hardware counters,
# each counter group in its own short run, for one render by the native driver (default: Cornell
# 1920x1080 at 256 spp; RT_PROFILE_ARGS replaces the scene/size options).  Output: gpurun_out/<tag>/...
set -u
TAG=${1:-prof}
shift || true
EXTRA="$@"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
BIN=$REPO/raytracer_amd/bin/raytracer
ARGS="${RT_PROFILE_ARGS:---scene cornell --width 1920 --height 1080 --spp 256} --out /tmp/render_prof.bmp $EXTRA"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp

timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BIN $ARGS > $OUT/stats.log 2>&1 || { echo "stats run failed"; tail -5 $OUT/stats.log; exit 1; }

pass() {
    name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $BIN $ARGS > $OUT/$name.log 2>&1 || { echo "pmc pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
pass pmc_valu   SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS
pass pmc_wait   SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32
pass pmc_mix    SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FLOPS_FP32 SQ_LEVEL_WAVES SQ_CYCLES
pass pmc_fetch  FETCH_SIZE
pass pmc_write  WRITE_SIZE
pass pmc_grbm   GRBM_GUI_ACTIVE GRBM_COUNT
find $OUT -name '*.csv' | head -40
