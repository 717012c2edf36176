#!/bin/bash
# Collects the round's profiles on the GPU box: rocprofv3 kernel stats of bench.py, then PMC passes (one run each)
# over a few frames of the default pipeline.  Usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --headline-only --sync-gather > $OUT/bench_under_rocprof.json 2> $OUT/bench.err || exit 1
pass() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/prof_frame.py 6 5 > /dev/null 2>&1 || exit 1; }
pass pmc_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass pmc_sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT
pass pmc_cache TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
ls -R $OUT | head -40
